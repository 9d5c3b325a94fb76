"""ON THE GPU BOX: the stage-entry shortcut forwards (1x1 + BatchNorm epilogue, no residual) at 2048 images: fast (EP == 3) vs generic epilogue."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simhand_amd import ops
N = 2048
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (h, cin, cout) in ((56, 64, 256), (28, 256, 512)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(N, h, h, cin, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 1, 1, device="cuda", generator=g) / math.sqrt(cin)).to(torch.bfloat16).float()
    d = ops.conv_desc(N, h, h, cin, cout, 1, 1, 1, 0, torch.bfloat16)
    wk = ops.pack_krsc(w, torch.bfloat16)
    st = ops.BNState(cout, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.1)
    new = t(lambda: ops.conv2d_fwd_bnact(d, x, wk, st, False, None))
    ops.test_switch("G1_PF", 0)
    old = t(lambda: ops.conv2d_fwd_bnact(d, x, wk, st, False, None))
    ops.test_switch("G1_PF", -1)
    gb = (x.numel() + N * h * h * cout) * 2 / 1e9
    print(f"({cin},{cout},1,1,{h}) fwd + BN epilogue at 2048 images: fast {new:.0f} us ({gb / new * 1e6:.0f} GB/s), generic {old:.0f} us")
