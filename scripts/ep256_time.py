"""ON THE GPU BOX: what the BatchNorm / residual / ReLU / mask epilogue of igemm256_kernel costs on the stage-4 conv3 shape
(512 -> 2048 at 7 x 7, 2048 images) and on the other big-tile 1x1 forwards, piece by piece."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simhand_amd import ops, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
lib = _lib.load()
for (h, cin, cout) in ((7, 512, 2048), (14, 1024, 256), (14, 256, 1024), (28, 512, 256)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(N, h, h, cin, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 1, 1, device="cuda", generator=g) / math.sqrt(cin)).to(torch.bfloat16).float()
    d = ops.conv_desc(N, h, h, cin, cout, 1, 1, 1, 0, torch.bfloat16)
    wk = ops.pack_krsc(w, torch.bfloat16)
    res = torch.randn(N, h, h, cout, device="cuda", generator=g).to(torch.bfloat16)
    st = ops.BNState(cout, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.1)
    for mode in (2, 0):
        lib.simhand_test_igemm256_enable(mode)
        ops.route_reset()
        r = {}
        r["plain"] = t(lambda: ops.conv2d_fwd(d, x, wk, False))
        r["plain+stats"] = t(lambda: ops.conv2d_fwd(d, x, wk, True))
        r["bn"] = t(lambda: ops.conv2d_fwd_bnact(d, x, wk, st, False, None))
        r["bn+relu"] = t(lambda: ops.conv2d_fwd_bnact(d, x, wk, st, True, None))
        r["bn+res+relu"] = t(lambda: ops.conv2d_fwd_bnact(d, x, wk, st, True, res))
        r["bn+res+relu+mask"] = t(lambda: ops.conv2d_fwd_bnact(d, x, wk, st, True, res, True))
        routes = {k: v for k, v in ops.route_counts().items() if v}
        print(f"({cin},{cout},1,1,{h}) igemm256_enable={mode}: " + "  ".join(f"{k} {v:.0f}" for k, v in r.items()) + f"  us   routes {sorted(routes)}")
    lib.simhand_test_igemm256_enable(1)
