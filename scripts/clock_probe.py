"""One 3x3 256 -> 256 @ 14^2 forward (2048 images) per operand pattern, for a rocprofv3 --kernel-trace --pmc pass (scripts/clock_report.py):
effective clock = GRBM_GUI_ACTIVE / XCDs / duration, matrix-pipe busy fraction at THAT clock."""
import sys, torch
sys.path.insert(0, ".")
from simhand_amd import ops
N, h, cin, cout = 2048, 14, 256, 256
dtype = torch.bfloat16
d = ops.conv_desc(N, h, h, cin, cout, 3, 3, 1, 1, dtype)
for name, xs, ws in (("random", 1.0, 0.05), ("zero", 0.0, 0.0), ("random", 1.0, 0.05), ("zero", 0.0, 0.0)):
    x = (torch.randn(N, h, h, cin, device="cuda") * xs).to(dtype)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * ws
    wk = ops.pack_krsc(w, dtype)
    for _ in range(6):
        ops.conv2d_fwd(d, x, wk, True)
    torch.cuda.synchronize()
