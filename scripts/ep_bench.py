"""conv3 forward of a Bottleneck as the folded forward runs it (BN + residual + ReLU + bit mask in the epilogue), per stage;
optional k:mf arguments set the rows per block of the activation-stationary kernel."""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
N = 2048; dt = torch.bfloat16
for c in sys.argv[1:]:
    k, mf = map(int, c.split(":"))
    ops._lib_dev().simhand_test_conv1x1_set_rows(k, mf)
for w, h in ((64, 56), (128, 28), (256, 14), (512, 7)):
    cout = 4 * w
    d = ops.conv_desc(N, h, h, w, cout, 1, 1, 1, 0, dt)
    x = torch.randn(N, h, h, w, device="cuda").to(dt)
    wk = ops.pack_krsc(torch.randn(cout, w, 1, 1, device="cuda") * 0.05, dt)
    res = torch.randn(N, h, h, cout, device="cuda").to(dt)
    st = ops.BNState(cout, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.0)
    fn = lambda: ops.conv2d_fwd_bnact(d, x, wk, st, True, res, want_mask=True)
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10 * 1e3
    m = N * h * h
    gb = (m * w * 2 + 2 * m * cout * 2 + m * cout / 8) / 1e9
    print(f"w={w:4d} @{h:3d}: {t:.3f} ms  {gb / t:.2f} TB/s")
    del x, res
# stage 4 again on the 128 x 128 tile kernel (three blocks per CU overlap each other's epilogues)
ops._lib_dev().simhand_test_igemm256_enable(0)
w, h = 512, 7
cout = 4 * w
d = ops.conv_desc(N, h, h, w, cout, 1, 1, 1, 0, dt)
x = torch.randn(N, h, h, w, device="cuda").to(dt)
wk = ops.pack_krsc(torch.randn(cout, w, 1, 1, device="cuda") * 0.05, dt)
res = torch.randn(N, h, h, cout, device="cuda").to(dt)
st = ops.BNState(cout, "cuda"); st.scale.fill_(1.0); st.shift.fill_(0.0)
fn = lambda: ops.conv2d_fwd_bnact(d, x, wk, st, True, res, want_mask=True)
fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): fn()
torch.cuda.synchronize(); print(f"w= 512 @  7 on the 128-row kernel: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
ops._lib_dev().simhand_test_igemm256_enable(1)
