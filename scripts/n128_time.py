"""ON THE GPU BOX: gemm_n128_kernel against the 128-row tile kernel (switch N128 = 0) at 2048 images: conv3's folded data gradient of the
stage-2 Bottlenecks (128 <- 512 + 128, bias, fused sums) and conv1's forward (512 -> 128) with BatchNorm partial sums."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simhand_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
h, c, k1, k2 = 28, 128, 512, 128
g = torch.Generator(device="cuda").manual_seed(1)
d = ops.conv_desc(N, h, h, c, k1, 1, 1, 1, 0, torch.bfloat16)
gy = torch.randn(N, h, h, k1, device="cuda", generator=g).to(torch.bfloat16)
a2 = torch.randn(N, h, h, k2, device="cuda", generator=g).to(torch.bfloat16)
wa = (torch.randn(c, k1, device="cuda", generator=g) / math.sqrt(k1)).to(torch.bfloat16)
wm = (torch.randn(c, k2, device="cuda", generator=g) / math.sqrt(k2)).to(torch.bfloat16)
bias = torch.randn(c, device="cuda", generator=g)
y2 = torch.randn(N, h, h, c, device="cuda", generator=g).to(torch.bfloat16)
st = ops.BNState(c, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.1)
dx = torch.empty(N, h, h, c, device="cuda", dtype=torch.bfloat16)
f = lambda: ops.conv2d_dgrad_ex(d, gy, wa, dx=dx, bias=bias, x2=a2, wt2=wm, fuse_mode=2, prev_y=y2, prev_st=st)
new = t(f); ops.test_switch("N128", 0); old = t(f); ops.test_switch("N128", -1)
gb = (gy.numel() + a2.numel() + y2.numel() + dx.numel()) * 2 / 1e9
print(f"folded data gradient 128 <- 512 + 128 @ 28^2, {N} images ({gb:.2f} GB): n128 {new:.0f} us ({gb / new * 1e6:.0f} GB/s), 128-row tile kernel {old:.0f} us")
df = ops.conv_desc(N, h, h, k1, c, 1, 1, 1, 0, torch.bfloat16)
wk = ops.pack_krsc((torch.randn(c, k1, 1, 1, device="cuda", generator=g) / math.sqrt(k1)).to(torch.bfloat16).float(), torch.bfloat16)
f = lambda: ops.conv2d_fwd(df, gy, wk, True)
new = t(f); ops.test_switch("N128", 0); old = t(f); ops.test_switch("N128", -1)
gb = (gy.numel() + dx.numel()) * 2 / 1e9
print(f"forward 512 -> 128 @ 28^2 + BN sums ({gb:.2f} GB): n128 {new:.0f} us ({gb / new * 1e6:.0f} GB/s), 128-row tile kernel {old:.0f} us")
