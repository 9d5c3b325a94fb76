"""One activation-stationary 1x1 launch form, a few times (for rocprofv3 --pmc passes).
usage: one_1x1.py {fwd|ep|dgrad|dgrad_acc2} w h [N]   (Bottleneck width w at side h: the wide side has 4w channels)"""
import sys, torch
sys.path.insert(0, ".")
from simhand_amd import ops
mode, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
dt = torch.bfloat16
m, cw = N * h * h, 4 * w
one = ops.BNState(cw, "cuda"); one.scale.fill_(1.0); one.shift.fill_(0.0)
if mode in ("fwd", "ep"):
    d = ops.conv_desc(N, h, h, w, cw, 1, 1, 1, 0, dt)
    x = torch.randn(N, h, h, w, device="cuda").to(dt)
    wk = ops.pack_krsc(torch.randn(cw, w, 1, 1, device="cuda") * 0.05, dt)
    res = torch.randn(N, h, h, cw, device="cuda").to(dt)
    fn = (lambda: ops.conv2d_fwd(d, x, wk, want_stats=True)) if mode == "fwd" else (lambda: ops.conv2d_fwd_bnact(d, x, wk, one, True, res, want_mask=True))
else:
    d = ops.conv_desc(N, h, h, cw, w, 1, 1, 1, 0, dt)
    dy = torch.randn(N, h, h, w, device="cuda").to(dt)
    wt = ops.pack_crsk(torch.randn(w, cw, 1, 1, device="cuda") * 0.05, dt)
    g = torch.randn(N, h, h, cw, device="cuda").to(dt)
    _, mask = ops.bn_apply(g.view(m, cw), one, m, cw, True, None, want_mask=True)
    _, mask2 = ops.bn_apply(torch.randn(m, cw, device="cuda").to(dt), one, m, cw, True, None, want_mask=True)
    fn = (lambda: ops.conv2d_dgrad(d, dy, wt)) if mode == "dgrad" else \
         (lambda: ops.conv2d_dgrad_ex(d, dy, wt, res_grad=g, res_mask=mask, fuse_mode=4, prev_mask=mask2, want_sums=False))
for _ in range(4):
    fn()
torch.cuda.synchronize()
