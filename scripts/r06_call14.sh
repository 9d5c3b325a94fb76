#!/bin/bash
set -u
mkdir -p gpurun_out/cfg
python bench.py --steps 20 --warmup 3 > gpurun_out/cfg/r06_bench_extra.log 2>&1
tail -1 gpurun_out/cfg/r06_bench_extra.log > gpurun_out/cfg/r06_bench_extra.json
python -c "
import json; d=json.load(open('gpurun_out/cfg/r06_bench_extra.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['device_state']['sclk_mhz']['mean'], d['device_state']['socket_power_w']['mean'], d['roofline']['hbm']['stale'])"
python scripts/stability_run.py 160 gpurun_out/cfg/r06_stability_160steps.md > gpurun_out/cfg/r06_stability.log 2>&1
tail -12 gpurun_out/cfg/r06_stability_160steps.md | cut -c1-250
