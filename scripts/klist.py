"""Per-launch durations of the kernels matching a pattern in a rocprofv3 kernel-trace DB, last step only.
usage: python scripts/klist.py <results.db> <substring> [n]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gcols = [c for c in cols if "grid" in c.lower() or "workgroup" in c.lower()]
sel = ", ".join(["name", "start", "end"] + gcols)
rows = [r for r in cur.execute(f"select {sel} from kernels order by start") if sys.argv[2] in r[0]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
print(gcols)
for r in rows[-n:]:
    print(f"{(r[2]-r[1])/1e3:9.1f} us  ", r[3:])
