"""Per-layer micro-benchmark of the conv kernels on the ResNet-50 shapes at a given image count.
Prints time, TFLOP/s and the algorithmic GB/s next to the HBM / MFMA bounds."""
import sys
import time
import torch
sys.path.insert(0, ".")
from simhand_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dtype = torch.bfloat16
which = sys.argv[2] if len(sys.argv) > 2 else "all"
# (cin, cout, k, stride, hin)
SHAPES = [(192, 64, 1, 1, 112), (64, 64, 1, 1, 56), (64, 64, 3, 1, 56), (64, 256, 1, 1, 56), (256, 64, 1, 1, 56), (256, 128, 1, 1, 56),
          (128, 128, 3, 2, 56), (128, 512, 1, 1, 28), (256, 512, 1, 2, 56), (512, 128, 1, 1, 28), (128, 128, 3, 1, 28),
          (512, 256, 1, 1, 28), (256, 256, 3, 2, 28), (256, 1024, 1, 1, 14), (512, 1024, 1, 2, 28), (1024, 256, 1, 1, 14),
          (256, 256, 3, 1, 14), (1024, 512, 1, 1, 14), (512, 512, 3, 2, 14), (512, 2048, 1, 1, 7), (1024, 2048, 1, 2, 14),
          (2048, 512, 1, 1, 7), (512, 512, 3, 1, 7)]
COUNT = [1, 1, 3, 4, 2, 1, 1, 4, 1, 3, 3, 1, 1, 6, 1, 5, 5, 1, 1, 3, 1, 2, 2]

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print(f"{'shape':34s} {'GF':>7s} {'MB':>7s} | {'fwd ms':>7s} {'TF/s':>6s} {'GB/s':>6s} | {'dgrad':>7s} {'TF/s':>6s} {'+bnsum':>7s} | {'wgrad':>7s} {'TF/s':>6s} | bound ms (hbm 5TB/s, mfma 1.5PF)")
for (cin, cout, k, s, h), cnt in zip(SHAPES, COUNT):
    pad = k // 2
    d = ops.conv_desc(N, h, h, cin, cout, k, k, s, pad, dtype)
    x = torch.randn(N, h, h, cin, device="cuda").to(dtype)
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    wk, wt = ops.pack_krsc(w, dtype), ops.pack_crsk(w, dtype)
    dy = torch.randn(N, d.ho, d.wo, cout, device="cuda").to(dtype)
    flops = 2.0 * N * d.ho * d.wo * cout * cin * k * k
    byts = 2.0 * (x.numel() + dy.numel())
    tf = timeit(lambda: ops.conv2d_fwd(d, x, wk, True)) if which in ("all", "fwd") else 0
    td = timeit(lambda: ops.conv2d_dgrad(d, dy, wt)) if which in ("all", "dgrad") and cin != 192 else 0
    tw = timeit(lambda: ops.conv2d_wgrad(d, x, dy)) if which in ("all", "wgrad") else 0
    tdf = 0
    if which in ("all", "dgrad", "fused") and cin != 192:
        # dgrad with the previous unit's BN-backward sums fused (mask recomputed from y): +1 read of a dx-sized tensor
        st = ops.BNState(cin, x.device)
        st.scale.fill_(1.0); st.shift.fill_(0.0)
        dxb = torch.empty_like(x)
        tdf = timeit(lambda: ops.conv2d_dgrad_fused(d, dy, wt, x, st, None, dx=dxb))
    tot["fwd"] += tf * cnt; tot["dgrad"] += td * cnt; tot["wgrad"] += tw * cnt; tot["dgrad_fused"] = tot.get("dgrad_fused", 0.0) + tdf * cnt
    f = lambda t: f"{t*1e3:7.3f} {flops/t/1e12 if t else 0:6.0f}"
    print(f"{str((cin,cout,k,s,h))+'x'+str(cnt):34s} {flops/1e9:7.0f} {byts/1e6:7.0f} | {f(tf)} {byts/tf/1e9 if tf else 0:6.0f} | {f(td)} {tdf*1e3:7.3f} | {f(tw)} | {byts/5e12*1e3:.3f} {flops/1.5e15*1e3:.3f}")
    del x, dy
print("per-step totals (ms):", {k: round(v * 1e3, 2) for k, v in tot.items()})
