"""The layer-1 conv3 folded data gradient (64 <- 256 + 64 second segment + bias + fused BatchNorm-backward sums; the generic 128 x 64 tile kernel), 2048 images,
and the stage-2 / short-K siblings that share igemm_kernel: time per launch (isolation; A/B two builds with --lib PATH)."""
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
for cin, cout, h in ((64, 256, 56),):
    d = ops.conv_desc(n, h, h, cin, cout, 1, 1, 1, 0, dt)
    dy = rnd(n, h, h, cout).to(dt)
    wt = (rnd(cin, cout) / math.sqrt(cout)).to(dt)
    bias = rnd(cin)
    x2 = rnd(n, h, h, cin).to(dt); wt2 = (rnd(cin, cin) / math.sqrt(cin)).to(dt)
    y_prev = rnd(n, h, h, cin).to(dt)
    st = ops.BNState(cin, DEV); st.scale.copy_(rnd(cin)); st.shift.copy_(rnd(cin) * 0.3)
    ops.route_reset()
    r = {"single segment (activation-stationary 1x1 kernel)": timed(lambda: ops.conv2d_dgrad_ex(d, dy, wt)),
         "+bias+seg2": timed(lambda: ops.conv2d_dgrad_ex(d, dy, wt, bias=bias, x2=x2, wt2=wt2)),
         "+bias+seg2+sums (the step's form)": timed(lambda: ops.conv2d_dgrad_ex(d, dy, wt, bias=bias, x2=x2, wt2=wt2, fuse_mode=2, prev_y=y_prev, prev_st=st))}
    gb = {"single segment (activation-stationary 1x1 kernel)": 4.11 + 0.0, "+bias+seg2": 4.93, "+bias+seg2+sums (the step's form)": 5.75}
    print(f"({cin} <- {cout}) 1x1 @ {h}^2, {n} images")
    for k, v in r.items(): print(f"      {k:55s} {v:8.1f} us   {gb[k] / v * 1e3:5.2f} TB/s")
