"""Same as contention_probe.py at 256 tiles, with random vs all-zero operands (zero operands toggle no MFMA datapath bits: the power / clock share)."""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import _lib, ops
lib = _lib.load()
lib.simhand_test_igemm256_enable(2)
dt = torch.bfloat16
cin = 1024
for n in (128, 256):
    for zero in (False, True):
        d = ops.conv_desc(n, 16, 16, cin, 256, 3, 3, 1, 1, dt)
        x = (torch.zeros if zero else torch.randn)(n, 16, 16, cin, device="cuda").to(dt)
        wk = ops.pack_krsc((torch.zeros if zero else torch.randn)(256, cin, 3, 3, device="cuda") * 0.05, dt)
        fn = lambda: ops.conv2d_fwd(d, x, wk, True)
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 20 * 1e6
        print(f"{n} tiles, {'zero  ' if zero else 'random'} operands: {t:7.1f} us, {t / (9 * cin // 64):.3f} us per k-step")
