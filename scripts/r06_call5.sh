#!/bin/bash
set -u
mkdir -p gpurun_out/cfg
export TMPDIR=/tmp
python -m pytest tests/test_gpu_merged_launches.py "tests/test_gpu_backbone_ops.py::test_stem_direct_conv" tests/test_gpu_configs.py::test_rn50_handclr_w_bf16_at_the_reference_128px_geometry tests/test_gpu_fp8.py tests/test_gpu_fullsize.py::test_fullsize_stem_forward_and_wgrad -m gpu -x -q --timeout=900 > gpurun_out/gputests_c.log 2>&1
tail -5 gpurun_out/gputests_c.log
{
  for cfg in "--image-size 128 --per-gpu-batch 3136" "--image-size 128 --per-gpu-batch 3136 --switch STEM_RING=0 --switch STEM_WG_RING=0" "--image-size 128 --per-gpu-batch 3136" "--image-size 128 --per-gpu-batch 3136 --switch STEM_RING=0 --switch STEM_WG_RING=0" "--image-size 128" "--image-size 128 --switch STEM_RING=0 --switch STEM_WG_RING=0"; do
    python bench.py --steps 8 --warmup 3 --no-cpu-baseline $cfg 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('| bench.py $cfg |', round(d['ms_per_step'], 2), 'ms/step |', round(d['value']), 'pairs/s |', d['dtype'], '|', round(d['roofline']['step_tflops_per_gpu']), 'TFLOP/s whole step |')"
  done
} > gpurun_out/cfg/r06_stem_128px_ab.md
cat gpurun_out/cfg/r06_stem_128px_ab.md
bash scripts/ab_generic.sh "" "--switch FUSE_S2=1" 2 > gpurun_out/cfg/r06_fuse_s2_ab.txt 2>&1
bash scripts/ab_generic.sh "" "--switch FUSE_S2=2" 2 >> gpurun_out/cfg/r06_fuse_s2_ab.txt 2>&1
cat gpurun_out/cfg/r06_fuse_s2_ab.txt
