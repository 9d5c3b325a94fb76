"""Forward 1x1 convolutions with the BatchNorm + residual + ReLU epilogue on the 256 x 256 kernel (the conv3 / shortcut layers of stages 3-4), 2048 images:
plain forward next to the epilogue form (isolation; A/B two builds with --lib PATH)."""
import sys, time, math, torch
sys.path.insert(0, ".")
from simhand_amd import ops

if "--lib" in sys.argv:  # another build of the library (scripts/build_variant.sh), before its first use
    from simhand_amd import _lib as _sh_lib

    _sh_lib.set_library_paths(sys.argv[sys.argv.index("--lib") + 1])
DEV, dt = "cuda", torch.bfloat16
def timed(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6
n = 2048
g = torch.Generator(device=DEV).manual_seed(1)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
for cin, cout, h, cnt in ((512, 2048, 7, 3), (1024, 2048, 7, 1), (512, 1024, 14, 1), (256, 1024, 14, 6)):
    d = ops.conv_desc(n, h, h, cin, cout, 1, 1, 1, 0, dt)
    x = rnd(n, h, h, cin).to(dt)
    wk = ops.pack_krsc(rnd(cout, cin, 1, 1) / math.sqrt(cin), dt)
    st = ops.BNState(cout, DEV); st.scale.copy_(rnd(cout)); st.shift.copy_(rnd(cout) * 0.3)
    res = rnd(n, h, h, cout).to(dt)
    ops.route_reset()
    r = {"plain + stats": timed(lambda: ops.conv2d_fwd(d, x, wk, True)),
         "bn + relu": timed(lambda: ops.conv2d_fwd_bnact(d, x, wk, st, True)),
         "bn + residual + relu + mask": timed(lambda: ops.conv2d_fwd_bnact(d, x, wk, st, True, residual=res, want_mask=True))}
    rc = ops.route_counts()
    print(f"({cin} -> {cout}) 1x1 @ {h}^2 x{cnt}: " + "  ".join(f"{k} {v:7.1f}" for k, v in r.items()) + f" us   igemm256_fwd launches {rc.get('igemm256_fwd')}")
    del x, res
