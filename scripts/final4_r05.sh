#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_gputests_final.log; cat gpurun_out/r05_gputests_final.log
