"""Counters of the kernels matching a substring in a rocprofv3 --pmc database, per kernel name: summed over hardware instances, averaged
over the dispatches.  usage: pmc_dump.py <results.db> <kernel-name-substring>"""
import collections, re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
pat = sys.argv[2]
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
idc = next((c for c in cols if "dispatch" in c.lower()), None)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for row in cur.execute(f"select name, counter_name, counter_value{', ' + idc if idc else ''} from pmc_events"):
    if pat in row[0]:
        acc[re.sub(r"\(.*", "", row[0])[:70]][row[1]][row[3] if idc else 0] += row[2]
for kern, cs in sorted(acc.items()):
    print(f"== {kern}")
    for k, d in sorted(cs.items()):
        v = list(d.values())
        print(f"  {k:42s} dispatches={len(v)} mean={sum(v) / len(v):.5g}")
