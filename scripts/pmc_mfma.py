"""Matrix-core utilisation per kernel from one rocprofv3 --pmc pass over bench.py.
Counters: SQ_VALU_MFMA_BUSY_CYCLES (cycles the MFMA pipes were busy, summed over the chip's SIMDs),
SQ_INSTS_VALU_MFMA_MOPS_BF16 (bf16 MFMA work issued, units of 512 FLOP), SQ_BUSY_CYCLES, SQ_WAVE_CYCLES,
SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY (quad-cycles, MI355X_MICROARCH.md), GRBM_GUI_ACTIVE (chip-busy cycles).
  MFMA busy %   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs)      (gfx94x MfmaUtil formula)
  MFMA FLOP     = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512
  wave time split = WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY over WAVE_CYCLES
usage: pmc_mfma.py <pmc results.db> <out.md> [kernel-trace results.db for durations]"""
import collections
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
print("pmc_events columns:", cols)
idc = next((c for c in cols if "dispatch" in c.lower()), None)
# one row per (dispatch, counter, hardware instance -- e.g. one per XCD): counters that COUNT work add up over the instances,
# GRBM_GUI_ACTIVE (chip-busy cycles, the same wall clock seen by every instance) is taken once per dispatch (max)
per_disp = collections.defaultdict(lambda: collections.defaultdict(list))
q = f"select name, counter_name, counter_value{', ' + idc if idc else ''} from pmc_events"
auto_id = collections.Counter()
for row in cur.execute(q):
    name = re.sub(r"\(.*", "", row[0])
    did = row[3] if idc else None
    per_disp[(name, did)][row[1]].append(row[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for (name, did), d in per_disp.items():
    inst = max(1, len(d.get("GRBM_GUI_ACTIVE", [0])))
    n_disp = 1 if idc else max(1, inst // 8)  # without a dispatch id: assume the 8 XCD instances per dispatch
    for k, v in d.items():
        acc[name][k] += (max(v) if idc else sum(v) / (inst / n_disp)) if k == "GRBM_GUI_ACTIVE" else sum(v)
    cnt[name] += n_disp
dur = {}
if len(sys.argv) > 3:
    c2 = sqlite3.connect(sys.argv[3]).cursor()
    kc = [r[1] for r in c2.execute("pragma table_info(kernels)")]
    nc = "name" if "name" in kc else [c for c in kc if "name" in c][0]
    for n, c, t in c2.execute(f"select {nc}, count(*), sum(end-start) from kernels group by {nc}"):
        dur[re.sub(r"\(.*", "", n)] = (c, t)
rows = []
for name, d in acc.items():
    gui = d.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    wave = d.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    rows.append((gui, name, cnt[name], 100.0 * busy / (gui * 256 * 4), d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512,
                 100 * d.get("SQ_WAIT_ANY", 0) / wave, 100 * d.get("SQ_WAIT_INST_ANY", 0) / wave, 100 * d.get("SQ_ACTIVE_INST_ANY", 0) / wave))
rows.sort(reverse=True)
tot_gui = sum(r[0] for r in rows)
out = ["| kernel | launches | share of GPU-busy cycles % | MFMA busy % | MFMA TFLOP issued (bf16) | TFLOP/s (kernel-trace time) | wave cycles: wait % | issue-stall % | issuing % |",
       "|---|---|---|---|---|---|---|---|---|"]
for gui, name, n, util, fl, w1, w2, w3 in rows[:40]:
    tfs = ""
    if name in dur and dur[name][1] > 0 and n > 0:
        per_launch_ns = dur[name][1] / dur[name][0]
        tfs = f"{fl / n / per_launch_ns / 1e3:.0f}"
    out.append(f"| {name[:100]} | {n} | {100 * gui / tot_gui:.1f} | {util:.1f} | {fl / 1e12:.2f} | {tfs} | {w1:.0f} | {w2:.0f} | {w3:.0f} |")
busy_all = sum(acc[r[1]].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for r in rows)
out.append(f"\nwhole profiled run: MFMA busy {100.0 * busy_all / (tot_gui * 256 * 4):.1f} % of GPU-busy cycles, "
           f"{sum(r[4] for r in rows) / 1e12:.1f} TFLOP issued on the bf16 matrix pipes")
text = "\n".join(out)
print(text)
open(sys.argv[2], "w").write(text + "\n")
