"""ON THE GPU BOX: conv1's masked-store data gradient of a stage-entry block at 2048 images, with and without the merged shortcut gradient."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simhand_amd import ops
N = 2048
def t(fn, reps=8):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (h, cin, cw) in ((56, 256, 128), (28, 512, 256), (14, 1024, 512)):
    g = torch.Generator(device="cuda").manual_seed(1)
    d1 = ops.conv_desc(N, h, h, cin, cw, 1, 1, 1, 0, torch.bfloat16)
    dy1 = torch.randn(N, h, h, cw, device="cuda", generator=g).to(torch.bfloat16)
    w1 = (torch.randn(cw, cin, 1, 1, device="cuda", generator=g) / math.sqrt(cin)).to(torch.bfloat16).float()
    wt1 = ops.pack_crsk(w1, torch.bfloat16)
    m = N * h * h
    pmask = torch.randint(0, 256, (m, cin // 8), device="cuda", generator=g, dtype=torch.uint8)
    dsub = torch.randn(N, h // 2, h // 2, cin, device="cuda", generator=g).to(torch.bfloat16)
    ops.route_reset()
    t3 = t(lambda: ops.conv2d_dgrad_ex(d1, dy1, wt1, fuse_mode=4, prev_mask=pmask, want_sums=False, sub_grad=dsub))
    r3 = {k: v for k, v in ops.route_counts().items() if v}
    ops.route_reset()
    t2 = t(lambda: ops.conv2d_dgrad_ex(d1, dy1, wt1, fuse_mode=4, prev_mask=pmask, want_sums=False))
    dx = torch.empty(N, h, h, cin, device="cuda", dtype=torch.bfloat16)
    ts = t(lambda: ops.scatter2_add(dsub, dx, pmask))
    gb3 = (dy1.numel() * 2 + dsub.numel() * 2 + pmask.numel() + m * cin * 2) / 1e9
    print(f"conv1 dgrad {cw}->{cin} @ {h}^2: merged {t3:.0f} us ({gb3 / t3 * 1e6:.0f} GB/s of {gb3:.2f} GB), masked store only {t2:.0f} us, scatter-add pass {ts:.0f} us; routes {list(r3)}")
