import sys, math, torch
sys.path.insert(0, '.')
from simhand_amd import ops
N=2048
d = ops.conv_desc(N, 56, 56, 128, 128, 3, 3, 2, 1, torch.bfloat16)
g = torch.Generator(device='cuda').manual_seed(1)
dy = torch.randn(N, 28, 28, 128, device='cuda', generator=g).to(torch.bfloat16)
w = (torch.randn(128,128,3,3, device='cuda', generator=g)/34).to(torch.bfloat16).float()
wt = ops.pack_crsk(w, torch.bfloat16)
lib = ops._lib_dev()
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a,b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps*1e3
new = t(lambda: ops.conv2d_dgrad(d, dy, wt))
lib.simhand_test_conv3x3_r128_enable(0)
old = t(lambda: ops.conv2d_dgrad(d, dy, wt))
lib.simhand_test_conv3x3_r128_enable(-1)
print(f"(128,128,3,2,56) dgrad at 2048 images: ring {new:.0f} us, parity-class tile launches {old:.0f} us")
