#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['hbm']['stale'], d['device_state']['sclk_mhz']['mean'])"
