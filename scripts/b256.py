"""Times the launches that take the 256x256 tile kernel (fwd + dgrad), median of event-timed repeats.  usage: b256.py [N]"""
import sys, torch
sys.path.insert(0, ".")
from simhand_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dt = torch.bfloat16
SH = [(256, 256, 3, 1, 14), (512, 512, 3, 1, 7), (1024, 256, 1, 1, 14), (2048, 512, 1, 1, 7), (1024, 512, 1, 1, 14), (512, 256, 1, 1, 28)]
def ev(fn, it=12):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2] * 1e3
out = []
for cin, cout, k, s, h in SH:
    d = ops.conv_desc(N, h, h, cin, cout, k, k, s, k // 2, dt)
    x = torch.randn(N, h, h, cin, device="cuda").to(dt); dy = torch.randn(N, d.ho, d.wo, cout, device="cuda").to(dt)
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    wk, wt = ops.pack_krsc(w, dt), ops.pack_crsk(w, dt)
    fl = 2.0 * N * d.ho * d.wo * cout * cin * k * k
    tf = ev(lambda: ops.conv2d_fwd(d, x, wk, True)); td = ev(lambda: ops.conv2d_dgrad(d, dy, wt))
    st = ops.BNState(cin, x.device); st.scale.fill_(1.0); st.shift.fill_(0.0)
    dxb = torch.empty_like(x)
    tdf = ev(lambda: ops.conv2d_dgrad_fused(d, dy, wt, x, st, None, dx=dxb))
    out.append(f"{(cin,cout,k,s,h)}: fwd {tf:6.1f} us {fl/tf/1e6:5.0f} TF | dgrad {td:6.1f} us {fl/td/1e6:5.0f} TF | dgrad+sums {tdf:6.1f} us")
print("\n".join(out))
