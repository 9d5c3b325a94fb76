"""Data-gradient forms of the 256 x 256 kernel whose epilogues load per-chunk operands (bias, residual gradient + masks, fused BatchNorm-backward sums),
at 2048 images: time per launch next to the plain store of the same shape (isolation; A/B two builds with --lib PATH)."""
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
# (cin = destination channels, cout = reduction, h, count per step, label)
for cin, cout, h, cnt, label in ((256, 1024, 14, 6, "conv3 of stage 3 (folded: + bias + second segment + fused sums)"),
                                 (512, 2048, 7, 3, "conv3 of stage 4 (folded)"),
                                 (2048, 512, 7, 2, "conv1 of stage 4 (masked residual merge)"),
                                 (1024, 512, 14, 1, "conv1 of stage 4's entry block")):
    d = ops.conv_desc(n, h, h, cin, cout, 1, 1, 1, 0, dt)
    dy = rnd(n, h, h, cout).to(dt)
    wt = (rnd(cin, cout) / math.sqrt(cout)).to(dt)
    r = {}
    r["plain"] = timed(lambda: ops.conv2d_dgrad_ex(d, dy, wt))
    bias = rnd(cin)
    r["+bias"] = timed(lambda: ops.conv2d_dgrad_ex(d, dy, wt, bias=bias))
    if ops.conv2d_dgrad_concat_ok(d, cin):
        x2 = rnd(n, h, h, cin).to(dt); wt2 = (rnd(cin, cin) / math.sqrt(cin)).to(dt)
        y_prev = rnd(n, h, h, cin).to(dt)
        st = ops.BNState(cin, DEV); st.scale.copy_(rnd(cin)); st.shift.copy_(rnd(cin) * 0.3)
        r["+bias+seg2"] = timed(lambda: ops.conv2d_dgrad_ex(d, dy, wt, bias=bias, x2=x2, wt2=wt2))
        r["+bias+seg2+sums"] = timed(lambda: ops.conv2d_dgrad_ex(d, dy, wt, bias=bias, x2=x2, wt2=wt2, fuse_mode=2, prev_y=y_prev, prev_st=st))
        del x2, y_prev
    res = rnd(n, h, h, cin).to(dt)
    mask = torch.randint(0, 256, (n, h, h, cin // 8), device=DEV, dtype=torch.uint8, generator=g)
    r["masked residual"] = timed(lambda: ops.conv2d_dgrad_ex(d, dy, wt, res_grad=res, res_mask=mask))
    pm = torch.randint(0, 256, (n, h, h, cin // 8), device=DEV, dtype=torch.uint8, generator=g)
    r["masked residual + masked store (mode 4)"] = timed(lambda: ops.conv2d_dgrad_ex(d, dy, wt, res_grad=res, res_mask=mask, fuse_mode=4, prev_mask=pm, want_sums=False))
    rc = ops.route_counts()
    print(f"({cin} <- {cout}) 1x1 @ {h}^2 x{cnt}  [{label}]  igemm256 hits so far {rc.get('igemm256_dgrad')}")
    for k, v in r.items(): print(f"      {k:45s} {v:8.1f} us")
    del dy, res, mask, pm
