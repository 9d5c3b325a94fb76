"""ON THE GPU BOX: bn_apply + 3x3 forward as two launches vs simhand_conv2d_fwd_bnin (BatchNorm + ReLU applied in the kernel's LDS ring) at
2048 images."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simhand_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for (h, c) in ((56, 64), (28, 128)):
    d = ops.conv_desc(N, h, h, c, c, 3, 3, 1, 1, torch.bfloat16)
    if not ops.conv2d_fwd_bnin_ok(d):
        print(f"({c},{c},3,1,{h}): no bnin kernel"); continue
    g = torch.Generator(device="cuda").manual_seed(1)
    y_in = torch.randn(N, h, h, c, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(c, c, 3, 3, device="cuda", generator=g) / math.sqrt(9 * c)).to(torch.bfloat16).float()
    wk = ops.pack_krsc(w, torch.bfloat16)
    st = ops.BNState(c, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.1)
    m = N * h * h
    ta = t(lambda: ops.bn_apply(y_in.view(m, c), st, m, c, True, None))
    a = ops.bn_apply(y_in.view(m, c), st, m, c, True, None).view(N, h, h, c)
    tc = t(lambda: ops.conv2d_fwd(d, a, wk, True))
    tf = t(lambda: ops.conv2d_fwd_bnin(d, y_in, st, wk, True))
    a2, y2, p2 = ops.conv2d_fwd_bnin(d, y_in, st, wk, True)
    y1, p1 = ops.conv2d_fwd(d, a, wk, True)
    print(f"({c},{c},3,1,{h}) at {N} images: bn_apply {ta:.0f} us + conv {tc:.0f} us = {ta + tc:.0f} us; one launch {tf:.0f} us; "
          f"bit-identical: a {torch.equal(a, a2)} y {torch.equal(y1, y2)} sums {torch.equal(p1, p2)}")
