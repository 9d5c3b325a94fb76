for i in 1 2; do for v in 0 1; do echo "NT=$v: $(SIMHAND_STEM_RING_NT=$v python scripts/stem_bench.py 2>/dev/null | grep -E '^fwd|bn\+relu\+pool fwd' | tr '\n' ' ')"; done; done
timeout 600 python -m pytest tests/test_gpu_backbone_ops.py -q -x -k 'stem' 2>&1 | tail -1
for v in 0 1 0 1; do echo "step NT=$v: $(SIMHAND_STEM_RING_NT=$v python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["kernel_ms_per_step"]; print(round(d["ms_per_step"],2), round(k["conv_fwd"],2), round(k["bn"],2))')"; done
