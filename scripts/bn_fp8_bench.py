"""bn_apply vs bn_apply_fp8 (fused e4m3 emission) vs bn_apply + quantize at the stage-3 / stage-4 conv2 inputs, 2048 images."""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
def timed(fn):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e6
for c, h in ((256, 14), (512, 7), (256, 28), (512, 14)):
    y = torch.randn(2048, h, h, c, device="cuda").to(torch.bfloat16)
    m = 2048 * h * h
    st = ops.BNState(c, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.1)
    sc = ops.FP8Scaler("cuda", delayed=True)
    sc.bn_apply_quantize(y, st, True)
    t_plain = timed(lambda: ops.bn_apply(y.view(m, c), st, m, c, True))
    t_fused = timed(lambda: sc.bn_apply_quantize(y, st, True))
    a = ops.bn_apply(y.view(m, c), st, m, c, True)
    t_q = timed(lambda: sc.quantize(a))
    print(f"c={c} h={h}: bn_apply {t_plain:.0f} us | fused bn_apply_fp8 (+ scale update) {t_fused:.0f} us | separate quantize {t_q:.0f} us")
