import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
N, dtype = 2048, torch.bfloat16
for (cin, cout, h) in ((64, 256, 56), (128, 512, 28), (256, 1024, 14), (256, 64, 56)):
    d = ops.conv_desc(N, h, h, cin, cout, 1, 1, 1, 0, dtype)
    x = torch.randn(N, h, h, cin, device="cuda").to(dtype)
    wk = ops.pack_krsc(torch.randn(cout, cin, 1, 1, device="cuda") * 0.05, dtype)
    a = timeit(lambda: ops.conv2d_fwd(d, x, wk, True))
    b = timeit(lambda: ops.conv2d_fwd(d, x, wk, False))
    y = torch.empty(N, h, h, cout, device="cuda", dtype=dtype)
    c = timeit(lambda: y.copy_(y))  # pure streaming copy of the output size (read+write)
    z = timeit(lambda: y.zero_())
    print(f"{cin}->{cout}@{h}: with stats {a:.3f} ms, without {b:.3f} ms; torch copy of out {c:.3f} ms, memset {z:.3f} ms")
