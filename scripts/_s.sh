SIMHAND_STEM_RING_FL=1 timeout 600 python -m pytest tests/test_gpu_backbone_ops.py tests/test_gpu_fullsize.py -q -x -k 'stem' 2>&1 | tail -1
for i in 1 2 3; do for v in 0 1; do echo "FL=$v: $(SIMHAND_STEM_RING_FL=$v python scripts/stem_bench.py 2>/dev/null | grep -E '^fwd' | tr '\n' ' ')"; done; done
