"""fp8 vs bf16 forward of the MFMA-bound layers at 2048 images (e4m3 variant of the 256 x 256 LDS-DMA kernel vs its bf16 form)."""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
dt = torch.bfloat16
for cin, cout, k, s, h in [(256, 256, 3, 1, 14), (512, 512, 3, 1, 7), (256, 256, 3, 2, 28), (512, 512, 3, 2, 14), (128, 128, 3, 1, 28), (1024, 256, 1, 1, 14)]:
    d = ops.conv_desc(2048, h, h, cin, cout, k, k, s, k // 2, dt)
    x = torch.randn(2048, h, h, cin, device="cuda").to(dt)
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    wk = ops.pack_krsc(w, dt)
    sx, sw = ops.FP8Scaler("cuda", delayed=True), ops.FP8Scaler("cuda", delayed=False)
    xq, wq = sx.quantize(x), sw.pack_weights(w)
    def timed(fn):
        fn(); fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 10
    tb = timed(lambda: ops.conv2d_fwd(d, x, wk, True))
    ok = ops.conv2d_fwd_fp8_supported(d)
    tf = timed(lambda: ops.conv2d_fwd_fp8(d, xq, wq, sx, sw)) if ok else float("nan")
    tq = timed(lambda: sx.quantize(x))
    fl = 2.0 * 2048 * d.ho * d.wo * cout * cin * k * k
    print(f"({cin},{cout},{k},{s},{h}): bf16 {tb*1e6:6.0f} us {fl/tb/1e12:5.0f} TF | fp8 {tf*1e6:6.0f} us {fl/tf/1e12:5.0f} TF | quantize pass {tq*1e6:5.0f} us")
