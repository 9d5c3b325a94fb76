import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops, _lib
N, h, cin, cout = 2048, 14, 256, 256
dtype = torch.bfloat16
d = ops.conv_desc(N, h, h, cin, cout, 3, 3, 1, 1, dtype)
x = torch.zeros(N, h, h, cin, device="cuda", dtype=dtype)
wk = ops.pack_krsc(torch.zeros(cout, cin, 3, 3, device="cuda"), dtype)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
out = []
for ring in (0, 1, 0, 1):
    _lib.load().simhand_test_switch(16, ring)
    out.append(f"ring={ring} {timeit(lambda: ops.conv2d_fwd(d, x, wk, True))*1e6:6.1f} us")
print(" | ".join(out))
