#!/bin/bash
set -u
mkdir -p gpurun_out/refresh
REFRESH_FAST=1 bash scripts/refresh_profiles.sh r06 > gpurun_out/refresh_r06c.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/refresh/r06_bench_b1024.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['device_state']['sclk_mhz']['mean'], d['device_state']['socket_power_w']['mean'], d['roofline']['hbm']['stale'], d['roofline']['hbm']['bytes_per_step'])"
