"""Ordered kernel timeline of the LAST training step in a rocprofv3 kernel-trace DB (a step ends with opt_update_kernel):
start offset, duration, idle gap before the launch, short kernel name, grid.  usage: python scripts/timeline.py <results.db> [out.txt]"""
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gcols = [c for c in cols if c.lower() in ("grid_size_x", "grid_x", "workgroup_size_x", "grid_size", "workgroup_size")]
sel = ", ".join(["name", "start", "end"] + gcols)
rows = list(cur.execute(f"select {sel} from kernels order by start"))
ends = [i for i, r in enumerate(rows) if "opt_update_kernel" in r[0]]
lo, hi = (ends[-2] + 1, ends[-1] + 1) if len(ends) >= 2 else (0, len(rows))
step = rows[lo:hi]
t0 = step[0][1]
out = [f"# {len(step)} kernels, {(step[-1][2] - t0) / 1e6:.2f} ms wall, {sum(r[2] - r[1] for r in step) / 1e6:.2f} ms busy; columns: start_ms dur_us gap_us kernel {gcols}"]
prev_end = t0
for r in step:
    name = re.sub(r"\(.*", "", r[0]).replace("void ", "").replace("sh::", "").replace("unsigned short", "bf16")
    out.append(f"{(r[1] - t0) / 1e6:8.3f} {(r[2] - r[1]) / 1e3:8.1f} {(r[1] - prev_end) / 1e3:6.1f}  {name[:90]}  {r[3:]}")
    prev_end = max(prev_end, r[2])
text = "\n".join(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
else:
    print(text)
