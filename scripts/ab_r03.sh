# ON THE GPU BOX: round-3 tree (scripts/abl/r03, git archive of a86a630 built in place; not tracked) vs this tree, alternating on one box
for i in 1 2 3; do
python scripts/abl/r03/bench.py --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('round 3 (a86a630)'.ljust(20), round(d['ms_per_step'],2), round(d['value']), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn','misc')})"
python bench.py --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('round 4 (this tree)'.ljust(20), round(d['ms_per_step'],2), round(d['value']), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn','misc')})"
done
