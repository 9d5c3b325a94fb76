"""Rows-per-block sweep of the short-K 1x1 kernel on the ResNet-50 shapes. usage: conv1x1_tune.py [N]"""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lib = _lib.load()
dt = torch.bfloat16
SH = [(64, 64, 56), (64, 256, 56), (256, 64, 56), (256, 128, 56), (128, 512, 28), (512, 128, 28), (512, 256, 28), (256, 1024, 14), (1024, 256, 14)]
def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
for cin, cout, h in SH:
    d = ops.conv_desc(N, h, h, cin, cout, 1, 1, 1, 0, dt)
    x = torch.randn(N, h, h, cin, device="cuda").to(dt); dy = torch.randn(N, h, h, cout, device="cuda").to(dt)
    w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.05
    wk, wt = ops.pack_krsc(w, dt), ops.pack_crsk(w, dt)
    byts = 2.0 * (x.numel() + dy.numel())
    out = []
    for mf in (4, 2, 1):
        for k in (64, 128, 256):
            if not (mf == 4 and k == 256): lib.simhand_test_conv1x1_set_rows(k, mf)
        tf = timeit(lambda: ops.conv2d_fwd(d, x, wk, True)) if cin <= 256 and not (mf == 4 and cin == 256) else float("nan")
        td = timeit(lambda: ops.conv2d_dgrad(d, dy, wt)) if cout <= 256 and not (mf == 4 and cout == 256) else float("nan")
        out.append(f"mf{mf}: fwd {tf:6.3f} dgrad {td:6.3f}")
    print(f"({cin},{cout},{h}) bound {byts/5e12*1e3:.3f} | " + " | ".join(out))
