"""Per dispatch of kernels matching <pattern>: duration, effective clock and matrix-pipe busy fraction from a rocprofv3 --kernel-trace --pmc db.
usage: clock_report.py <results.db> <pattern>"""
import collections, re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
pat = sys.argv[2]
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
idc = next(c for c in cols if "dispatch" in c.lower())
ctr = collections.defaultdict(dict)
names = {}
for name, cn, v, did in cur.execute(f"select name, counter_name, counter_value, {idc} from pmc_events"):
    if pat in name:
        ctr[did][cn] = ctr[did].get(cn, 0.0) + v
        names[did] = re.sub(r"\(.*", "", name)[:60]
kc = [r[1] for r in cur.execute("pragma table_info(kernels)")]
kid = next((c for c in kc if "dispatch" in c.lower()), None)
dur = {}
if kid:
    for did, s, e in cur.execute(f"select {kid}, start, end from kernels"):
        dur[did] = (e - s) * 1e-9
for did in sorted(ctr):
    c = ctr[did]
    t = dur.get(did)
    line = f"{did:6d} {names[did]:60s}"
    if t:
        line += f" {t*1e6:7.1f} us"
        if "GRBM_GUI_ACTIVE" in c:
            line += f"  clock {c['GRBM_GUI_ACTIVE'] / 8 / t / 1e9:5.2f} GHz"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        line += f"  MFMA busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (c['GRBM_GUI_ACTIVE'] / 8):5.1f} %"
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in c and "SQ_WAVE_CYCLES" in c:
            line += f"  {k[3:]} {100 * c[k] / c['SQ_WAVE_CYCLES']:4.1f}%"
    print(line)
