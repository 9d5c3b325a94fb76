"""Diagnostic (SH_ABL256 == 30 build): cycles a wave of igemm256_kernel spends per k-step in the counted DMA wait and in the barrier."""
import sys, torch
sys.path.insert(0, ".")
from simhand_amd import ops
N, h = 2340, 14
for cin in (256, 1024):
    d = ops.conv_desc(N, h, h, cin, 256, 3, 3, 1, 1, torch.bfloat16)
    x = torch.randn(N, h, h, cin, device="cuda").to(torch.bfloat16)
    wk = ops.pack_krsc(torch.randn(256, cin, 3, 3, device="cuda") * 0.05, torch.bfloat16)
    for _ in range(3):
        y, _ = ops.conv2d_fwd(d, x, wk, False)
    torch.cuda.synchronize()
    m = N * h * h
    tiles = m // 256
    st = y.view(m, 256)[: tiles * 256].view(tiles, 256, 256)[:, :8, :8].contiguous().view(torch.float32).view(tiles, 8, 4).double()
    nk = st[0, 0, 3].item()
    per = st[:, :, :3].mean(dim=(0,)) / nk          # [wave][dma, bar, all] cycles (s_memtime ticks) per k-step
    print(f"cin={cin} nk={nk:.0f}: per k-step and wave (s_memtime ticks = 100 MHz? / shader cycles): ")
    for w in range(8):
        print(f"   wave {w}: dma-wait {per[w,0]:7.1f}  barrier {per[w,1]:7.1f}  loop total {per[w,2]:7.1f}")
    print(f"   mean  : dma-wait {per[:,0].mean():7.1f}  barrier {per[:,1].mean():7.1f}  loop total {per[:,2].mean():7.1f}")
