"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units).
FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for 16-B-per-lane coalesced reads on
gfx950; WRITE_SIZE is taken as reported (uncalibrated).  usage: pmc_traffic.py fetch.db write.db out.md out.json"""
import json, re, sqlite3, sys

def load(path):
    cur = sqlite3.connect(path).cursor()
    out = {}
    for name, cnt, tot in cur.execute("select name, count(*), sum(counter_value) from pmc_events group by name"):
        out[re.sub(r"\(.*", "", name)] = (cnt, tot)
    return out

f, w = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in f:
    cnt, fk = f[k]
    wk = w.get(k, (cnt, 0.0))[1]
    rd, wr = 2.0 * fk * 1024, wk * 1024
    rows.append((rd + wr, k, cnt, rd, wr))
rows.sort(reverse=True)
lines = ["| kernel | launches | HBM read GB (2xFETCH_SIZE) | HBM write GB (WRITE_SIZE) | bytes / launch (MB) |", "|---|---|---|---|---|"]
for tot, k, cnt, rd, wr in rows:  # EVERY kernel (round 4 cut the table at 25 rows: 95 GB per step could not be attributed from it)
    lines.append(f"| {k[:90]} | {cnt} | {rd/1e9:.2f} | {wr/1e9:.2f} | {tot/cnt/1e6:.1f} |")
lines.append(f"\nall kernels: read {sum(r[3] for r in rows)/1e9:.1f} GB, write {sum(r[4] for r in rows)/1e9:.1f} GB over the profiled run (2 steps: 1 warm-up + 1 timed)")
# the profiler classes bench.py reports (conv_fwd / conv_dgrad / conv_wgrad as its roofline classes; everything else by family)
cls = {"conv_fwd": (r"igemm_kernel<unsigned short, false", r"igemm256_kernel<false", r"gemm1x1_kernel<\d+, \d+, false", r"conv3x3_c64_kernel<[01]",
                    r"conv3x3_r128_kernel<[01]", r"gemm_n128_kernel<false", r"stem_ring_fwd_kernel", r"igemm_kernel<float, false"),
       "conv_dgrad": (r"igemm_kernel<unsigned short, true", r"igemm256_kernel<true", r"gemm1x1_kernel<\d+, \d+, true", r"conv3x3_c64_kernel<2",
                      r"conv3x3_r128_kernel<2", r"conv3x3_r128_s2dgrad_kernel", r"gemm_n128_kernel<true", r"igemm_kernel<float, true"),
       "conv_wgrad": (r"wgrad_kernel<", r"wgrad3x3_kernel", r"wgrad1x1_dma_kernel", r"stem_wgrad_ring_kernel", r"wgrad_reduce_kernel", r"wgrad_finish_kernel", r"stem_bwd_reduce_kernel")}
fam = {"batchnorm passes": (r"bn_apply_kernel", r"bn_bwd_apply", r"bn_bwd_partial", r"bn_partial", r"bn_relu_maxpool", r"pool_bn_bwd", r"bn_finalize", r"bn_bwd_finalize"),
       "folded-BatchNorm algebra": (r"fold_", r"bn_fold"),
       "pooling / layout / packing": (r"pool", r"subsample", r"scatter", r"pack_", r"stem_pad", r"nchw", r"cast"),
       "loss / post-process / head": (r"ntxent", r"dist_kernel", r"postprocess", r"proj_"),
       "optimizer": (r"opt_", r"lars"),
       "torch (fills, copies, cat)": (r"at::native", r"rocclr", r"elementwise")}
def in_any(k, groups):
    return any(re.search(p, k) for pats in groups.values() for p in pats)
steps_n = int(sys.argv[5]) if len(sys.argv) > 5 else 2
lines.append("\n## per class, GB per step (read + write)\n")
lines.append("| class | GB / step | read | write | launches / step |")
lines.append("|---|---|---|---|---|")
acc_all = 0.0
for name, pats in list(cls.items()) + list(fam.items()):
    sel = [r for r in rows if any(re.search(p, r[1]) for p in pats) and not (name in fam and in_any(r[1], cls))]
    if name in fam:  # first matching family wins
        prev = list(fam)[:list(fam).index(name)]
        sel = [r for r in sel if not any(re.search(p, r[1]) for q in prev for p in fam[q])]
    t = sum(r[0] for r in sel) / steps_n
    acc_all += t
    lines.append(f"| {name} | {t/1e9:.1f} | {sum(r[3] for r in sel)/steps_n/1e9:.1f} | {sum(r[4] for r in sel)/steps_n/1e9:.1f} | {sum(r[2] for r in sel)/steps_n:.0f} |")
rest = [r for r in rows if not in_any(r[1], cls) and not in_any(r[1], fam)]
lines.append(f"| unclassified | {sum(r[0] for r in rest)/steps_n/1e9:.1f} | {sum(r[3] for r in rest)/steps_n/1e9:.1f} | {sum(r[4] for r in rest)/steps_n/1e9:.1f} | {sum(r[2] for r in rest)/steps_n:.0f} |")
lines.append(f"| **step** | {sum(r[0] for r in rows)/steps_n/1e9:.1f} | | | |")
open(sys.argv[3], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
out = {}
for c, pats in cls.items():
    sel = [r for r in rows if any(re.search(p, r[1]) for p in pats)]
    n = sum(r[2] for r in sel)
    out[c] = {"bytes_per_launch": sum(r[0] for r in sel) / max(1, n), "launches_profiled": n,
              "read_bytes": sum(r[3] for r in sel), "write_bytes": sum(r[4] for r in sel)}
# whole-step view + provenance (bench.py quotes these only against the build they were measured on)
import os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 2  # bench.py --steps 1 --warmup 1
rd_all, wr_all = sum(r[3] for r in rows), sum(r[4] for r in rows)
out["step"] = {"bytes_per_step": (rd_all + wr_all) / steps, "read_bytes_per_step": rd_all / steps, "write_bytes_per_step": wr_all / steps,
               "steps_profiled": steps}
try:
    from bench import source_hash
    out["source_hash"] = source_hash()
except Exception as e:  # noqa: BLE001
    out["source_hash"] = None
try:
    out["commit"] = (subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip()
                     or os.environ.get("SIMHAND_COMMIT") or (open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".commit")).read().strip()
                                                             if os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".commit")) else None))
except Exception:  # noqa: BLE001 -- no git on the GPU box snapshot
    out["commit"] = os.environ.get("SIMHAND_COMMIT")
json.dump(out, open(sys.argv[4], "w"), indent=1)
