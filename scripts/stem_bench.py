"""Micro-benchmark of the direct stem kernels at N images of 224x224. usage: stem_bench.py [N]"""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dt = torch.bfloat16
x = torch.randn(N, 3, 224, 224, device="cuda")
w = torch.randn(64, 3, 7, 7, device="cuda") * 0.05
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
xp = ops.stem_pad_input(x, dt); wp = ops.stem_pack_weights(w, dt)
y, _ = ops.stem_conv_fwd(xp, wp, 224, 224)
dy = torch.randn_like(y)
print("pad   %.3f ms" % timeit(lambda: ops.stem_pad_input(x, dt)))
print("fwd   %.3f ms" % timeit(lambda: ops.stem_conv_fwd(xp, wp, 224, 224)))
print("wgrad %.3f ms" % timeit(lambda: ops.stem_conv_wgrad(xp, dy, 224, 224)))
