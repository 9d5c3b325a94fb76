#!/bin/bash
# round-5 closing run, second part: profile refresh + full GPU suite on the final sources
cd "$(dirname "$0")/.."
bash scripts/refresh_profiles.sh r05 > gpurun_out/refresh_r05.log 2>&1
tail -3 gpurun_out/refresh_r05.log | cut -c1-400
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_gputests_final.log; cat gpurun_out/r05_gputests_final.log
