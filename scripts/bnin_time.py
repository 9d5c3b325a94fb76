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
# the backward twin: bn_bwd_apply + 3x3 data gradient (with the previous unit's fused sums) vs sh_dy_src in the ring
for (h, c) in ((28, 128),):
    d = ops.conv_desc(N, h, h, c, c, 3, 3, 1, 1, torch.bfloat16)
    if not ops.conv2d_dgrad_dysrc_ok(d):
        print(f"({c},{c},3,1,{h}): no dy_src ring kernel"); continue
    g = torch.Generator(device="cuda").manual_seed(2)
    da = torch.randn(N, h, h, c, device="cuda", generator=g).to(torch.bfloat16)
    y = torch.randn(N, h, h, c, device="cuda", generator=g).to(torch.bfloat16)
    py = torch.randn(N, h, h, c, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(c, c, 3, 3, device="cuda", generator=g) / math.sqrt(9 * c)).to(torch.bfloat16).float()
    wt = ops.pack_crsk(w, torch.bfloat16)
    st = ops.BNState(c, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.1); st.mean.fill_(0.0); st.invstd.fill_(1.0)
    gamma = torch.ones(c, device="cuda"); m = N * h * h
    dgm, dbt = torch.randn(c, device="cuda"), torch.randn(c, device="cuda")
    coefs = ops.bn_bwd_coefs(st, gamma, dgm, dbt, m)
    dy = torch.empty_like(da)
    def two():
        bw = ops.bn_backward(da.view(m, c), None, y.view(m, c), st, gamma, m, c, True, False, mask_from_y=True, raw_partial=None)
        return ops.conv2d_dgrad_fused(d, bw[0].view(N, h, h, c), wt, py, st, None)
    t2 = t(two)
    t1 = t(lambda: ops.conv2d_dgrad_ex(d, None, wt, fuse_mode=2, prev_y=py, prev_st=st, dy_src=(da, y, st, coefs, True, dy)))
    tb = t(lambda: ops.conv2d_dgrad_fused(d, dy, wt, py, st, None))
    print(f"({c},{c},3,1,{h}) dgrad at {N} images: bn_backward (sums + apply) + dgrad {t2:.0f} us (dgrad alone {tb:.0f}); dy_src in the ring {t1:.0f} us "
          f"(+ the sums-only BatchNorm pass it still needs)")
