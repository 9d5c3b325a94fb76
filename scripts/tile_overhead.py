"""Per-block overhead of the 256x256 kernel: 3x3 conv fwd at fixed M (exactly 7 blocks per CU), N = 256, Cin = 128..1024;
fits t = rounds * (a + b * ksteps)."""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
dtype = torch.bfloat16
N, h = 2340, 14
res = []
for cin in (128, 256, 512, 1024):
    d = ops.conv_desc(N, h, h, cin, 256, 3, 3, 1, 1, dtype)
    x = torch.randn(N, h, h, cin, device="cuda").to(dtype)
    w = torch.randn(256, cin, 3, 3, device="cuda") * 0.05
    wk = ops.pack_krsc(w, dtype)
    fn = lambda: ops.conv2d_fwd(d, x, wk, True)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 10
    m = N * h * h
    blocks = (m + 255) // 256
    res.append((9 * cin // 64, t * 1e6 / (blocks / 256)))
    print(f"K={cin} ksteps={9*cin//64} t={t*1e3:.3f} ms per-round {res[-1][1]:.1f} us  {2.0*m*256*cin*9/t/1e12:.0f} TF")
    del x
(k0, t0_), (k1, t1_) = res[0], res[-1]
b = (t1_ - t0_) / (k1 - k0)
print(f"slope b = {b:.3f} us per k-step, intercept a = {t0_ - b * k0:.1f} us per block")
