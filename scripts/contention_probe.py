"""Is the 256 x 256 kernel's operand stream limited per CU (latency) or chip-wide (L2 / fabric bandwidth)?  The same tile work on 64 / 128 /
256 CUs (one round each, 3x3, 256 output channels, 16 x 16 images = one 256-row tile per image): equal times = per-CU limit."""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import _lib, ops
lib = _lib.load()
lib.simhand_test_igemm256_enable(2)
dt = torch.bfloat16
for cin in (256, 1024):
    for n in (32, 64, 128, 256):
        d = ops.conv_desc(n, 16, 16, cin, 256, 3, 3, 1, 1, dt)
        x = torch.randn(n, 16, 16, cin, device="cuda").to(dt)
        wk = ops.pack_krsc(torch.randn(256, cin, 3, 3, device="cuda") * 0.05, dt)
        fn = lambda: ops.conv2d_fwd(d, x, wk, True)
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 20 * 1e6
        print(f"cin {cin}: {n:3d} tiles (= CUs busy): {t:7.1f} us per launch, {t / (9 * cin // 64):.3f} us per k-step")
