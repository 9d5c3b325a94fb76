#!/bin/bash
set -u
mkdir -p gpurun_out/cfg
bash scripts/ab_generic.sh "" "--switch G1_LT=3" 3 > gpurun_out/cfg/r06_g1_lt_dgrad_ab.txt 2>&1
cat gpurun_out/cfg/r06_g1_lt_dgrad_ab.txt
