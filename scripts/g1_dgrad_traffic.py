"""Which operand of the stage-3 conv1 data gradient (1024 <- 256 @ 14^2, 2048 images; gemm1x1_kernel<256, 2, DGRAD, PF>) is fetched more often than once?
One variant per process (argv[1]) so that a rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass attributes the bytes:
  plain    : dy given, plain store                                   expected read  dy 0.2055 GB                      write dx 0.822
  masked   : dy given, store through the consumer's mask (PF = 2)    + fmask 0.051
  merge    : dy given, masked store + residual-gradient merge (PF=1) + res_grad 0.822 + res_mask 0.051
  step     : the step's form: dy derived on load (da, y), masked store + merge   + y 0.2055, + dy_out write 0.2055
usage (GPU box): rocprofv3 --pmc FETCH_SIZE -d /tmp/x -o f -- python scripts/g1_dgrad_traffic.py step ; python scripts/pmc_dump.py /tmp/x/f_results.db gemm1x1"""
import math
import sys

import torch

sys.path.insert(0, ".")
from simhand_amd import ops  # noqa: E402

variant = sys.argv[1]
n, h, cin, cout = 2048, 14, 1024, 256
dt, DEV = torch.bfloat16, "cuda"
g = torch.Generator(device=DEV).manual_seed(3)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
d = ops.conv_desc(n, h, h, cin, cout, 1, 1, 1, 0, dt)
m = n * h * h
dy = rnd(n, h, h, cout).to(dt)
y = rnd(n, h, h, cout).to(dt)
wt = ops.pack_crsk((rnd(cout, cin, 1, 1) / math.sqrt(cout)), dt)
res = rnd(n, h, h, cin).to(dt)
mask = torch.randint(0, 256, (m, cin // 8), dtype=torch.uint8, device=DEV, generator=g)
mask2 = torch.randint(0, 256, (m, cin // 8), dtype=torch.uint8, device=DEV, generator=g)
st = ops.BNState(cout, DEV)
st.scale.copy_(rnd(cout)); st.shift.copy_(rnd(cout) * 0.3)
coefs = (rnd(cout), rnd(cout) * 0.1, rnd(cout) * 0.01)
dy_out = torch.empty_like(dy)
fn = {"plain": lambda: ops.conv2d_dgrad_ex(d, dy, wt),
      "masked": lambda: ops.conv2d_dgrad_ex(d, dy, wt, fuse_mode=4, prev_mask=mask, want_sums=False),
      "merge": lambda: ops.conv2d_dgrad_ex(d, dy, wt, res_grad=res, res_mask=mask2, fuse_mode=4, prev_mask=mask, want_sums=False),
      "step": lambda: ops.conv2d_dgrad_ex(d, None, wt, res_grad=res, res_mask=mask2, fuse_mode=4, prev_mask=mask, want_sums=False,
                                          dy_src=(dy, y, st, coefs, True, dy_out))}[variant]
# flush the caches between launches with a 1-GB fill so that every launch starts cold, as in the step
junk = torch.empty(1 << 29, dtype=torch.float16, device=DEV)
for _ in range(6):
    junk.fill_(1.0)
    fn()
torch.cuda.synchronize()
print(variant, "done", ops.route_counts().get("gemm1x1_dgrad"))
