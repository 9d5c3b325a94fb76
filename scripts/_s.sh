timeout 500 python scripts/race_probe.py 400 1 4 2>&1 | tail -8
timeout 300 python scripts/race_probe.py 300 0 4 2>&1 | tail -5
