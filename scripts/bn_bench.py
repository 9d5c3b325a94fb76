"""Per-shape micro-benchmark of the BatchNorm streaming kernels on the ResNet-50 tensors at a given
image count: bn_apply (plain / residual+mask), bn_bwd_partial, bn_bwd_apply; prints ms and algorithmic GB/s."""
import sys
import time
import torch
sys.path.insert(0, ".")
from simhand_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
if len(sys.argv) > 2:
    ops._lib_dev().simhand_test_bn_set_nt(int(sys.argv[2]))
dtype = torch.bfloat16
# (channels, side, count of non-residual units, count of residual units (+ downsample, which has no relu))
SHAPES = [(64, 112, 1, 0), (64, 56, 6, 0), (256, 56, 1, 3), (128, 56, 1, 0), (128, 28, 7, 0), (512, 28, 1, 4), (256, 28, 1, 0),
          (256, 14, 11, 0), (1024, 14, 1, 6), (512, 14, 1, 0), (512, 7, 5, 0), (2048, 7, 1, 3)]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


tot = {}
print(f"{'shape':22s} {'MB':>6s} | {'apply':>7s} {'GB/s':>5s} | {'apply+res':>9s} {'GB/s':>5s} | {'bwd_part y':>10s} {'GB/s':>5s} | {'bwd_part m':>10s} {'GB/s':>5s} |"
      f" {'bwd_app y':>9s} {'GB/s':>5s} | {'bwd_app m':>9s} {'GB/s':>5s}")
for c, s, n_plain, n_res in SHAPES:
    m = N * s * s
    y = torch.randn(m, c, device="cuda").to(dtype)
    res = torch.randn(m, c, device="cuda").to(dtype)
    da = torch.randn(m, c, device="cuda").to(dtype)
    gamma = torch.ones(c, device="cuda")
    beta = torch.zeros(c, device="cuda")
    part = ops.bn_partial_stats(y, m, c)
    st = ops.bn_finalize(part, m, c, gamma, beta, None, None, None)
    one = 2.0 * m * c
    a, mask = ops.bn_apply(y, st, m, c, True, res, want_mask=True)
    lib = ops._lib_dev()
    nblk = lib.simhand_bn_stat_blocks(m, c)
    bp = torch.empty(nblk, 2, c, dtype=torch.float32, device="cuda")
    dg = torch.ones(c, device="cuda"); db = torch.ones(c, device="cuda")
    dy = torch.empty_like(y)
    P = ops._ptr; S = ops._stream

    def bwd_partial(mode):
        aa = mask if mode == 3 else None
        ops.check(lib.simhand_bn_bwd_partial(P(da), P(aa), P(y), P(st.mean), P(st.invstd), P(st.scale), P(st.shift), mode, m, c,
                                             ops.dt(dtype), P(bp), S()), "p")

    def bwd_apply(mode):
        aa = mask if mode == 3 else None
        ops.check(lib.simhand_bn_bwd_apply(P(da), P(aa), P(y), P(st.mean), P(st.invstd), P(gamma), P(dg), P(db), P(st.scale), P(st.shift),
                                           mode, P(dy), P(None), m, c, ops.dt(dtype), S()), "a")

    t = {}
    t["apply"] = timeit(lambda: ops.bn_apply(y, st, m, c, True, None, out=a))
    t["apply_res"] = timeit(lambda: ops.bn_apply(y, st, m, c, True, res, out=a, want_mask=False))
    t["bp_y"] = timeit(lambda: bwd_partial(2))
    t["bp_m"] = timeit(lambda: bwd_partial(3))
    t["ba_y"] = timeit(lambda: bwd_apply(2))
    t["ba_m"] = timeit(lambda: bwd_apply(3))
    byts = {"apply": 2 * one, "apply_res": 3 * one, "bp_y": 2 * one, "bp_m": 2 * one, "ba_y": 3 * one, "ba_m": 3 * one}
    cnt = {"apply": n_plain, "apply_res": n_res, "bp_y": n_plain, "bp_m": n_res, "ba_y": n_plain, "ba_m": n_res}
    for k in t:
        tot[k] = tot.get(k, 0.0) + t[k] * cnt[k]
    print(f"{str((c, s))+' x'+str(n_plain)+'+'+str(n_res):22s} {one/1e6:6.0f} | " +
          " | ".join(f"{t[k]*1e3:{w}.3f} {byts[k]/t[k]/1e9:5.0f}" for k, w in (("apply", 7), ("apply_res", 9), ("bp_y", 10), ("bp_m", 10), ("ba_y", 9), ("ba_m", 9))))
    del y, res, da, a, mask, dy
print("per-step totals (ms):", {k: round(v * 1e3, 2) for k, v in tot.items()}, "sum", round(sum(tot.values()) * 1e3, 2))
