"""ON THE GPU BOX: the parameter-sized launches of the folded BatchNorm (simhand_bn_fold_fwd / _bwd: centre, small GEMM, split sum, per-channel
algebra) per Bottleneck shape -- they sit on the step's critical path (20 + 20 calls per step)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simhand_amd import ops
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
tot = 0.0
for (cc, cw, cnt) in ((256, 64, 4), (512, 128, 4), (512, 256, 1), (1024, 256, 6), (1024, 512, 1), (2048, 512, 3), (2048, 1024, 1)):
    g = torch.Generator(device="cuda").manual_seed(cc + cw)
    m = 2048 * 49
    w = torch.randn(cc, cw, device="cuda", generator=g) / cw ** 0.5
    a = torch.randn(4096, cw, device="cuda", generator=g)
    s2 = (a.t() @ a) * (m / 4096.0); t2 = a.sum(0) * (m / 4096.0)
    gamma, beta = torch.ones(cc, device="cuda"), torch.zeros(cc, device="cuda")
    st, ws2 = ops.bn_fold_fwd(w, True, s2, t2, m, gamma, beta, None, None, None)
    gm = torch.randn(cc, cw, device="cuda", generator=g); s = torch.randn(cc, device="cuda", generator=g)
    tf = t(lambda: ops.bn_fold_fwd(w, True, s2, t2, m, gamma, beta, None, None, None))
    tb = t(lambda: ops.bn_fold_bwd(w, True, gm, s, ws2, t2, st, gamma, m, torch.bfloat16))
    tot += cnt * (tf + tb)
    print(f"cc {cc:5d} cw {cw:5d} x{cnt}: fold_fwd {tf:6.1f} us  fold_bwd {tb:6.1f} us   ({2.0 * cc * cw * cw / 1e9:.2f} GFLOP each)")
print(f"per step (counts of ResNet-50): {tot / 1e3:.2f} ms")
