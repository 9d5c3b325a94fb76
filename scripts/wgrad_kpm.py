"""Weight gradients per ResNet-50 shape for several split-K block targets (simhand_test_wgrad_target_blocks): ms incl. the reduce."""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
N = 2048; dt = torch.bfloat16
lib = ops._lib_dev()
targets = (256, 384, 512, 768, 1024)
shapes = [(64, 64, 3, 1, 56), (128, 128, 3, 1, 28), (256, 256, 3, 1, 14), (512, 512, 3, 1, 7), (128, 128, 3, 2, 56), (256, 256, 3, 2, 28), (512, 512, 3, 2, 14),
          (512, 128, 1, 1, 28), (1024, 256, 1, 1, 14), (2048, 512, 1, 1, 7)]
for cin, cout, k, st, h in shapes:
    d = ops.conv_desc(N, h, h, cin, cout, k, k, st, k // 2, dt)
    x = torch.randn(N, h, h, cin, device="cuda").to(dt); dy = torch.randn(N, d.ho, d.wo, cout, device="cuda").to(dt)
    r = []
    for tb in targets:
        lib.simhand_test_wgrad_target_blocks(tb, tb)
        fn = lambda: ops.conv2d_wgrad(d, x, dy)
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); r.append((time.perf_counter() - t0) / 10 * 1e3)
    lib.simhand_test_wgrad_target_blocks(0, 0)
    print(f"{cin:5d}->{cout:5d} k{k} s{st} @{h:3d}: " + "  ".join(f"{b}: {t:.3f}" for b, t in zip(targets, r)))
