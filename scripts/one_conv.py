"""Run one conv shape a few times (for rocprofv3 --pmc runs). usage: one_conv.py cin cout k stride h [N] [which]"""
import sys, torch
sys.path.insert(0, ".")
from simhand_amd import ops
cin, cout, k, s, h = map(int, sys.argv[1:6])
N = int(sys.argv[6]) if len(sys.argv) > 6 else 2048
which = sys.argv[7] if len(sys.argv) > 7 else "fwd"
dtype = torch.bfloat16
d = ops.conv_desc(N, h, h, cin, cout, k, k, s, k // 2, dtype)
x = torch.randn(N, h, h, cin, device="cuda").to(dtype)
w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
wk, wt = ops.pack_krsc(w, dtype), ops.pack_crsk(w, dtype)
dy = torch.randn(N, d.ho, d.wo, cout, device="cuda").to(dtype)
for _ in range(3):
    if which == "fwd": ops.conv2d_fwd(d, x, wk, True)
    elif which == "dgrad": ops.conv2d_dgrad(d, dy, wt)
    else: ops.conv2d_wgrad(d, x, dy)
torch.cuda.synchronize()
