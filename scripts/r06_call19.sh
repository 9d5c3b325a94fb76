#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/cfg
python -m pytest tests/test_gpu_backbone_ops.py -k "bnact or chain or 1x1" tests/test_gpu_configs.py::test_config1_rn50_handclr_w_bf16_every_route_against_oracle tests/test_gpu_configs.py::test_bn_on_load_step_is_bit_identical_to_the_separate_bn_apply_pass tests/test_gpu_merged_launches.py::test_rn50_step_is_bit_identical_with_fewer_launches tests/test_gpu_fullsize.py -m gpu -x -q --timeout=900 2>&1 | tail -3
{
for i in 1 2 3; do
for v in "--lib scripts/abl/libg1old.so" ""; do
python bench.py --no-cpu-baseline --no-parity-probe --steps 8 --warmup 3 $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('[$v]'.ljust(40), round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn','misc')})"
done; done
} > gpurun_out/cfg/r06_g1_fwd_masks_ab.txt 2>&1
cat gpurun_out/cfg/r06_g1_fwd_masks_ab.txt
