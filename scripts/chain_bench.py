"""conv3 + BN + residual + ReLU followed by the next block's conv1: two launches vs the chained launch (stage 1 of ResNet-50)."""
import sys, torch
sys.path.insert(0, ".")
from simhand_amd import ops
N, dt = 2048, torch.bfloat16
def ev(fn, it=10):
    fn(); fn(); torch.cuda.synchronize(); ts = []
    for _ in range(it):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2] * 1e3
for h, w in ((56, 64), (28, 128)):
    cout = 4 * w
    d = ops.conv_desc(N, h, h, w, cout, 1, 1, 1, 0, dt)
    d1 = ops.conv_desc(N, h, h, cout, w, 1, 1, 1, 0, dt)
    a2 = torch.relu(torch.randn(N, h, h, w, device="cuda")).to(dt)
    res = torch.relu(torch.randn(N, h, h, cout, device="cuda")).to(dt)
    w3 = ops.pack_krsc(torch.randn(cout, w, 1, 1, device="cuda") * 0.1, dt)
    w1 = ops.pack_krsc(torch.randn(w, cout, 1, 1, device="cuda") * 0.05, dt)
    st = ops.BNState(cout, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.0)
    out, _ = ops.conv2d_fwd_bnact(d, a2, w3, st, True, res, want_mask=True)
    t3 = ev(lambda: ops.conv2d_fwd_bnact(d, a2, w3, st, True, res, want_mask=True))
    t1 = ev(lambda: ops.conv2d_fwd(d1, out, w1, want_stats=True))
    tc = ev(lambda: ops.conv2d_fwd_bnact_chain(d, a2, w3, st, res, w1))
    print(f"w={w}@{h}: conv3+bnact {t3:.0f} us + next conv1 {t1:.0f} us = {t3 + t1:.0f} us; chained {tc:.0f} us")
