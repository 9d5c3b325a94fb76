"""Micro-benchmark of the direct stem kernels at N images of 224x224. usage: stem_bench.py [N]"""
import os, sys, time, torch
sys.path.insert(0, ".")
if os.environ.get("SH_LIB"):  # A/B against another build of the library
    from simhand_amd import _lib
    _lib.LIB_PATH = os.environ["SH_LIB"]
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
m = y.shape[0] * y.shape[1] * y.shape[2]
gamma = torch.ones(64, device="cuda"); beta = torch.zeros(64, device="cuda")
st = ops.bn_finalize(ops.bn_partial_stats(y.view(m, 64), m, 64), m, 64, gamma, beta, None, None, None)
pooled, idx, ywin = ops.bn_relu_maxpool_fwd(y, st, want_winner=True)
dz = torch.randn_like(pooled)
print("bn+relu+pool fwd        %.3f ms" % timeit(lambda: ops.bn_relu_maxpool_fwd(y, st, want_winner=True)))
print("bn+pool bwd (pooled st) %.3f ms" % timeit(lambda: ops.maxpool_bn_backward(dz, idx, y, st, gamma, ywin=ywin)))
print("bn+pool bwd (gather st) %.3f ms" % timeit(lambda: ops.maxpool_bn_backward(dz, idx, y, st, gamma)))
