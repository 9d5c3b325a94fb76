"""The 3x3 layers of the fp8 set at BASELINE configs[4]'s per-GPU batch (4096 images): forward / data gradient / weight gradient, bf16 vs e4m3,
each looped 20 times after a warm-up (isolation numbers; A/B two builds with --lib PATH)."""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops

if "--lib" in sys.argv:  # another build of the library (scripts/build_variant.sh), before its first use
    from simhand_amd import _lib as _sh_lib

    _sh_lib.set_library_paths(sys.argv[sys.argv.index("--lib") + 1])
DEV = "cuda"
def timed(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tot = {"bf16": 0.0, "fp8": 0.0}
for cin, cout, h, stride, count in ((256, 256, 14, 1, 5), (512, 512, 7, 1, 2), (256, 256, 28, 2, 1), (512, 512, 14, 2, 1)):
    g = torch.Generator(device=DEV).manual_seed(cin + h)
    x = torch.randn(n, h, h, cin, device=DEV, generator=g).relu().to(torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) * (2.0 / (cin * 9)) ** 0.5
    d = ops.conv_desc(n, h, h, cin, cout, 3, 3, stride, 1, torch.bfloat16)
    dy = (torch.randn(n, d.ho, d.wo, cout, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    wk, wc = ops.pack_krsc(wt, torch.bfloat16), ops.pack_crsk(wt, torch.bfloat16)
    sx, sw, sdy, swt = ops.FP8Scaler(DEV, True), ops.FP8Scaler(DEV, False), ops.FP8Scaler(DEV, True), ops.FP8Scaler(DEV, False)
    xq, wq, dyq, wtq = sx.quantize(x), sw.pack_weights(wt), sdy.quantize(dy), ops.fp8_pack_crsk(swt, wt)
    r = {}
    r["fwd bf16"] = timed(lambda: ops.conv2d_fwd(d, x, wk, True))
    r["fwd fp8"] = timed(lambda: ops.conv2d_fwd_fp8(d, xq, wq, sx, sw))
    r["dgrad bf16"] = timed(lambda: ops.conv2d_dgrad_ex(d, dy, wc))
    r["dgrad fp8"] = timed(lambda: ops.conv2d_dgrad_ex(d, dy, wc, fp8=(dyq, wtq, sdy, swt)))
    r["wgrad bf16"] = timed(lambda: ops.conv2d_wgrad(d, x, dy))
    r["wgrad fp8"] = timed(lambda: ops.conv2d_wgrad_fp8(d, xq, dyq, sx.state, sdy.state)) if ops.conv2d_wgrad_fp8_pays(d) else float("nan")
    for k, v in r.items():
        if v == v: tot[k.split()[1]] += v * count
    print(f"({cin},{cout},3,{stride},{h}) x{count} n={n}: " + "  ".join(f"{k} {v:7.1f}" for k, v in r.items()) + " us")
    del x, dy, xq, dyq
print(f"count-weighted sums (where both exist): bf16 {tot['bf16']/1e3:.2f} ms  fp8 {tot['fp8']/1e3:.2f} ms")
