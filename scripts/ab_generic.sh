# same-box A/B: bash scripts/ab_generic.sh "<flags A>" "<flags B>" [rounds]
A="$1"; B="$2"; R=${3:-3}
for i in $(seq 1 $R); do
for cfg in "$A" "$B"; do
python bench.py --no-cpu-baseline --steps 8 --warmup 3 $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('[$cfg]'.ljust(34), round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn','misc')})"
done; done
