#!/bin/bash
# round-5 closing run, last part: fast profile refresh (kernel stats, HBM traffic, bench line) + full GPU suite on the final sources
cd "$(dirname "$0")/.."
REFRESH_FAST=1 bash scripts/refresh_profiles.sh r05 > gpurun_out/refresh_r05.log 2>&1
tail -2 gpurun_out/refresh_r05.log | cut -c1-300
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_gputests_final.log; cat gpurun_out/r05_gputests_final.log
