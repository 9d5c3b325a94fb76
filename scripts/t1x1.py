"""1x1 forward (no statistics) over (cin, cout, side) shapes given as c:k:h arguments: ms and TB/s of x + y traffic."""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
N = 2048; dt = torch.bfloat16
shapes = [tuple(map(int, a.split(":"))) for a in sys.argv[1:]] or [(256, 1024, 14), (128, 512, 28), (64, 256, 56)]
for cin, cout, h in shapes:
    d = ops.conv_desc(N, h, h, cin, cout, 1, 1, 1, 0, dt)
    x = torch.randn(N, h, h, cin, device="cuda").to(dt); w = ops.pack_krsc(torch.randn(cout, cin, 1, 1, device="cuda") * 0.05, dt)
    fn = lambda: ops.conv2d_fwd(d, x, w, False)
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10 * 1e3
    gb = N * h * h * (cin + cout) * 2 / 1e9
    print(f"{cin}->{cout}@{h}: {t:.3f} ms  {gb / t:.2f} TB/s  {2.0 * N * h * h * cin * cout / t / 1e9:.0f} TF/s")
