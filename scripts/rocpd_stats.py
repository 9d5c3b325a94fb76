"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (markdown/CSV-ish).
usage: python scripts/rocpd_stats.py <results.db> [out.md]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
out = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for n, c, t, mn, mx in rows[:60]:
    n = re.sub(r"\(.*", "", n)
    out.append(f"| {n[:110]} | {c} | {t/1e6:.3f} | {t/c/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*t/total:.1f} |")
out.append(f"\ntotal kernel time {total/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
text = "\n".join(out)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
