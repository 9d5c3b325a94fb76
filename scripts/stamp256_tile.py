"""Diagnostic (SH_ABL256 == 31 build): where a tile of igemm256_kernel spends its cycles outside the k-loop -- prologue (decode + first DMAs issued),
loop, drain + BatchNorm sums, output-store ISSUE, output-store DRAIN (s_waitcnt vmcnt(0)); per wave, mean over the tiles of one launch."""
import sys, torch
sys.path.insert(0, ".")
from simhand_amd import ops
N, h = 2340, 14
for cin, stats in ((256, True), (256, False), (1024, True)):
    d = ops.conv_desc(N, h, h, cin, 256, 3, 3, 1, 1, torch.bfloat16)
    x = torch.randn(N, h, h, cin, device="cuda").to(torch.bfloat16)
    wk = ops.pack_krsc(torch.randn(256, cin, 3, 3, device="cuda") * 0.05, torch.bfloat16)
    for _ in range(3):
        y, _ = ops.conv2d_fwd(d, x, wk, stats)
    torch.cuda.synchronize()
    m = N * h * h
    tiles = m // 256
    st = y.view(m, 256)[: tiles * 256].view(tiles, 256, 256)[:, :8, :16].contiguous().view(torch.float32).view(tiles, 8, 8).double()
    nk = st[0, 0, 5].item()
    per = st[:, :, :5].mean(dim=0)
    print(f"cin={cin} nk={nk:.0f} stats={stats}: cycles per tile and wave (s_memtime ticks)")
    for w in range(8):
        print(f"   wave {w}: prologue {per[w,0]:8.0f}  loop {per[w,1]:9.0f} ({per[w,1]/nk:6.0f}/k-step)  drain+sums {per[w,2]:7.0f}  store issue {per[w,3]:7.0f}  store drain {per[w,4]:7.0f}")
    mean = per.mean(dim=0)
    print(f"   mean  : prologue {mean[0]:8.0f}  loop {mean[1]:9.0f}  drain+sums {mean[2]:7.0f}  store issue {mean[3]:7.0f}  store drain {mean[4]:7.0f}   outside the loop: {mean[0]+mean[2]+mean[3]+mean[4]:.0f}")
