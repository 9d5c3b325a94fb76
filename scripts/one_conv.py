"""One conv forward shape repeated (for PMC passes): python scripts/one_conv.py cin cout k stride h [n_images] [iters]"""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
cin, cout, k, s, h = (int(v) for v in sys.argv[1:6])
N = int(sys.argv[6]) if len(sys.argv) > 6 else 2048
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dtype = torch.bfloat16
d = ops.conv_desc(N, h, h, cin, cout, k, k, s, k // 2, dtype)
x = torch.randn(N, h, h, cin, device="cuda").to(dtype)
w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
wk = ops.pack_krsc(w, dtype)
fn = lambda: ops.conv2d_fwd(d, x, wk, True)
fn(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters): fn()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / iters
print(f"{(cin, cout, k, s, h)} N={N}: {t*1e6:.1f} us  {2.0*N*d.ho*d.wo*cout*cin*k*k/t/1e12:.0f} TF")
