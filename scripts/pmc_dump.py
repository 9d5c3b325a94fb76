import sqlite3, sys, re, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
pat = sys.argv[2]
acc = collections.defaultdict(list)
for name, cn, cv in cur.execute("select name, counter_name, counter_value from pmc_events"):
    if pat in name: acc[cn].append(cv)
for k, v in sorted(acc.items()):
    print(f"{k:28s} n={len(v)} last={v[-1]:.4g}")
