"""HBM bytes of single data-gradient launches of igemm256_kernel at 2048 images, one variant per process (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE):
  c3x3      : 3x3 256 @ 14^2, plain store                          operands: dy 205.5 MB -> dx 205.5 MB
  c3x3_sums : the same + the previous unit's BatchNorm-backward sums  + prev_y 205.5 MB read
  fold      : folded conv3 gradient 256 <- 1024 (+ 256) @ 14^2 + bias + sums   g 822 + a2 205.5 + y2 205.5 MB -> da2 205.5 MB
  n128_fold : the stage-2 twin on gemm_n128_kernel, 128 <- 512 (+ 128) @ 28^2      g 1644 + a2 411 + y2 411 MB -> da2 411 MB
usage (GPU box): rocprofv3 --pmc WRITE_SIZE -d /tmp/x -o f -- python scripts/i256_dgrad_traffic.py fold ; python scripts/pmc_dump.py /tmp/x/f_results.db igemm256"""
import math
import sys

import torch

sys.path.insert(0, ".")
if len(sys.argv) > 2:  # a variant build of the library (scripts/build_variant.sh)
    from simhand_amd import _lib
    _lib.set_library_paths(sys.argv[2], None)
from simhand_amd import ops  # noqa: E402

variant = sys.argv[1]
n, h = 2048, 14
dt, DEV = torch.bfloat16, "cuda"
g = torch.Generator(device=DEV).manual_seed(3)
rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
C = 128 if variant.startswith("n128") else 256
st = ops.BNState(C, DEV)
st.scale.copy_(rnd(C)); st.shift.copy_(rnd(C) * 0.3)
if variant.startswith("c3x3"):
    d = ops.conv_desc(n, h, h, 256, 256, 3, 3, 1, 1, dt)
    dy = rnd(n, h, h, 256).to(dt)
    wt = ops.pack_crsk(rnd(256, 256, 3, 3) / 48.0, dt)
    py = rnd(n, h, h, 256).to(dt)
    fn = (lambda: ops.conv2d_dgrad_ex(d, dy, wt)) if variant == "c3x3" else (lambda: ops.conv2d_dgrad_ex(d, dy, wt, fuse_mode=2, prev_y=py, prev_st=st))
elif variant == "n128_fold":
    h = 28
    d = ops.conv_desc(n, h, h, 128, 512, 1, 1, 1, 0, dt)
    gq = rnd(n, h, h, 512).to(dt)
    wa = (rnd(128, 512) / 24.0).to(dt)
    a2 = rnd(n, h, h, 128).to(dt)
    wm = (rnd(128, 128) / 12.0).to(dt)
    py = rnd(n, h, h, 128).to(dt)
    bias = rnd(128)
    fn = lambda: ops.conv2d_dgrad_ex(d, gq, wa, bias=bias, x2=a2, wt2=wm, fuse_mode=2, prev_y=py, prev_st=st)  # noqa: E731
else:
    d = ops.conv_desc(n, h, h, 256, 1024, 1, 1, 1, 0, dt)
    gq = rnd(n, h, h, 1024).to(dt)
    wa = (rnd(256, 1024) / 32.0).to(dt)
    a2 = rnd(n, h, h, 256).to(dt)
    wm = (rnd(256, 256) / 16.0).to(dt)
    py = rnd(n, h, h, 256).to(dt)
    bias = rnd(256)
    fn = lambda: ops.conv2d_dgrad_ex(d, gq, wa, bias=bias, x2=a2, wt2=wm, fuse_mode=2, prev_y=py, prev_st=st)  # noqa: E731
junk = torch.empty(1 << 29, dtype=torch.float16, device=DEV)
ops.route_reset()
for _ in range(6):
    junk.fill_(1.0)
    fn()
torch.cuda.synchronize()
if len(sys.argv) > 3:  # bit-identity of a variant build: dump (dx, partial) of one more call, or compare with an earlier dump
    import os
    dx, part = fn()
    torch.cuda.synchronize()
    if os.path.exists(sys.argv[3]):
        rdx, rpart = torch.load(sys.argv[3])
        print("identical to", sys.argv[3], ":", torch.equal(rdx, dx.cpu()), torch.equal(rpart, part.cpu()))
    else:
        torch.save((dx.cpu(), part.cpu()), sys.argv[3])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):  # isolated launch time, operands cold (the 1-GB fill in between), median of 5 x 4 calls
    junk.fill_(1.0)
    e0.record()
    for _ in range(4):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 4 * 1e3)
print(variant, "us per launch (median of 5):", round(sorted(ts)[2], 1))
print(variant, {k: v for k, v in ops.route_counts().items() if v})
