"""ON THE GPU BOX: every stem kernel at 2048 x 224^2 (bf16), one-pass chain vs the two-pass / fused-backward forms, HIP-event timed.
usage: python scripts/stem_bench.py [n]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simhand_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(n, 3, 224, 224, device=dev, generator=g)
wt = torch.randn(64, 3, 7, 7, device=dev, generator=g) / math.sqrt(147)
gamma = torch.rand(64, device=dev, generator=g) + 0.5
beta = torch.randn(64, device=dev, generator=g) * 0.2
xp = ops.stem_pad_input(x, torch.bfloat16)
wp = ops.stem_pack_weights(wt, torch.bfloat16)
del x


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out


m = n * 112 * 112
t_fwd, (y, part) = timed(lambda: ops.stem_conv_fwd(xp, wp, 224, 224))
st = ops.bn_finalize(part, m, 64, gamma, beta, None, None, None)
t_pool, (px, idx, ywin) = timed(lambda: ops.bn_relu_maxpool_fwd(y, st, want_winner=True))
dz = torch.randn(px.shape, device=dev, generator=g).to(torch.bfloat16)
t_bnb, (dy, dg, db) = timed(lambda: ops.maxpool_bn_backward(dz, idx, y, st, gamma, ywin=ywin))
t_wg, _ = timed(lambda: ops.stem_conv_wgrad(xp, dy, 224, 224))
del dy
t_stats, _ = timed(lambda: ops.stem_conv_stats(xp, wp, 224, 224))
t_pool2, _ = timed(lambda: ops.stem_conv_bn_relu_pool(xp, wp, st, 224, 224))
t_pool2n, _ = timed(lambda: ops.stem_conv_bn_relu_pool(xp, wp, st, 224, 224, want_winner=False))
t_bwd, _ = timed(lambda: ops.stem_backward_fused(xp, wp, dz, idx, ywin, st, gamma, 224, 224))
print(f"n = {n}: one-pass chain  conv+store {t_fwd:.3f}  bn+relu+pool {t_pool:.3f}  bn-bwd (partial+apply) {t_bnb:.3f}  wgrad {t_wg:.3f}  "
      f"= fwd {t_fwd + t_pool:.3f} / bwd {t_bnb + t_wg:.3f} ms")
print(f"n = {n}: two-pass        stats {t_stats:.3f}  conv+bn+relu+pool {t_pool2:.3f} (without ywin {t_pool2n:.3f})  fused backward (incl. pooled "
      f"partial + finalize + reduce) {t_bwd:.3f}  = fwd {t_stats + t_pool2:.3f} / bwd {t_bwd:.3f} ms")
