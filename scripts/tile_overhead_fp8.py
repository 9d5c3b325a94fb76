"""scripts/tile_overhead.py for the e4m3 variant of the 256 x 256 kernel next to bf16: 3x3 conv fwd at fixed M (7 rounds of one tile per CU),
N = 256, Cin = 128..1024; fits t = rounds * (a + b * ksteps).  A k-step is 64 KB of operands in both variants: 64 channels of bf16, 128 of e4m3."""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
N, h = 2340, 14
for mode in ("bf16", "fp8", "bf16", "fp8"):
    res = []
    for cin in (128, 256, 512, 1024):
        d = ops.conv_desc(N, h, h, cin, 256, 3, 3, 1, 1, torch.bfloat16)
        x = torch.randn(N, h, h, cin, device="cuda").to(torch.bfloat16)
        w = torch.randn(256, cin, 3, 3, device="cuda") * 0.05
        if mode == "bf16":
            wk = ops.pack_krsc(w, torch.bfloat16)
            fn = lambda: ops.conv2d_fwd(d, x, wk, True)
        else:
            sx, sw = ops.FP8Scaler(x.device, False), ops.FP8Scaler(x.device, False)
            xq, wq = sx.quantize(x), sw.pack_weights(w)
            fn = lambda: ops.conv2d_fwd_fp8(d, xq, wq, sx, sw, True)
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10
        m = N * h * h
        blocks = (m + 255) // 256
        ke = 64 if mode == "bf16" else 128
        res.append((9 * cin // ke, t * 1e6 / (blocks / 256)))
        print(f"{mode} K={cin} ksteps={9*cin//ke} t={t*1e3:.3f} ms per-round {res[-1][1]:.1f} us  {2.0*m*256*cin*9/t/1e12:.0f} TF")
        del x
    (k0, t0_), (k1, t1_) = res[0], res[-1]
    b = (t1_ - t0_) / (k1 - k0)
    print(f"{mode}: slope b = {b:.3f} us per k-step, intercept a = {t0_ - b * k0:.1f} us per block")
