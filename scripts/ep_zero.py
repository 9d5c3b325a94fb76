"""BN + residual + ReLU epilogue kernel on random vs all-zero operands: how much of its time is the clock (DVFS gives ~5 % back on zeros:
1.51 -> 1.43 ms at 64 @ 56^2, 0.507 -> 0.482 at 256 @ 14^2) -- i.e. not what makes its matrix phase and its HBM phases add up."""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
N = 2048; dt = torch.bfloat16
for zero in (False, True):
    for w, h in ((64, 56), (256, 14)):
        cout = 4 * w
        d = ops.conv_desc(N, h, h, w, cout, 1, 1, 1, 0, dt)
        mk = (lambda *s: torch.zeros(*s, device="cuda")) if zero else (lambda *s: torch.randn(*s, device="cuda"))
        x = mk(N, h, h, w).to(dt); res = mk(N, h, h, cout).to(dt)
        wk = ops.pack_krsc(mk(cout, w, 1, 1) * 0.05, dt)
        st = ops.BNState(cout, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.0)
        fn = lambda: ops.conv2d_fwd_bnact(d, x, wk, st, True, res, want_mask=True)
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20 * 1e3
        print(f"zero={zero} w={w}@{h}: {t:.3f} ms")
