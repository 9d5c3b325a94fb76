for i in 1 2 3; do
for cfg in "" "--engine stem_two_pass=1" "--switch STEM_WG_RING=0"; do
python bench.py --no-cpu-baseline --steps 8 --warmup 3 $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$cfg'.ljust(28), round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn')})"
done; done
