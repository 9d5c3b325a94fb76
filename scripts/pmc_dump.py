"""Counters of the kernels matching a substring in a rocprofv3 --pmc database: summed over hardware instances, averaged over
the dispatches (last = the final dispatch).  usage: pmc_dump.py <results.db> <kernel-name-substring>"""
import collections, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
pat = sys.argv[2]
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
idc = next((c for c in cols if "dispatch" in c.lower()), None)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for row in cur.execute(f"select name, counter_name, counter_value{', ' + idc if idc else ''} from pmc_events"):
    if pat in row[0]:
        acc[row[1]][row[3] if idc else 0] += row[2]
for k, d in sorted(acc.items()):
    v = list(d.values())
    print(f"{k:44s} dispatches={len(v)} mean={sum(v) / len(v):.5g} last={v[-1]:.5g}")
