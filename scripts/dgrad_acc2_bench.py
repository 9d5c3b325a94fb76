"""conv1 data gradient of a Bottleneck as the folded backward runs it: dx = dy W + res_grad * bit(res_mask), stored through the
previous block's ReLU mask (accumulate 2 + relu_mode 4), per ResNet-50 stage.  usage: dgrad_acc2_bench.py [N]"""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
N = 2048
for c in sys.argv[1:]:  # k:mf pairs -> rows per block of the activation-stationary kernel
    k, mf = map(int, c.split(":"))
    ops._lib_dev().simhand_test_conv1x1_set_rows(k, mf)
dt = torch.bfloat16
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
for w, h in ((64, 56), (128, 28), (256, 14), (512, 7)):
    cin = 4 * w
    d = ops.conv_desc(N, h, h, cin, w, 1, 1, 1, 0, dt)
    dy = torch.randn(N, h, h, w, device="cuda").to(dt)
    wt = ops.pack_crsk(torch.randn(w, cin, 1, 1, device="cuda") * 0.05, dt)
    g = torch.randn(N, h, h, cin, device="cuda").to(dt)
    one = ops.BNState(cin, "cuda"); one.scale.fill_(1.0); one.shift.fill_(0.0)
    m = N * h * h
    _, mask = ops.bn_apply(g.view(m, cin), one, m, cin, True, None, want_mask=True)
    _, mask2 = ops.bn_apply(torch.randn(m, cin, device="cuda").to(dt), one, m, cin, True, None, want_mask=True)
    t = timeit(lambda: ops.conv2d_dgrad_ex(d, dy, wt, res_grad=g, res_mask=mask, fuse_mode=4, prev_mask=mask2, want_sums=False))
    gb = (m * w * 2 + 2 * m * cin * 2 + 2 * m * cin / 8) / 1e9
    t2 = timeit(lambda: ops.conv2d_dgrad_ex(d, dy, wt, fuse_mode=4, prev_mask=mask2, want_sums=False))  # first block of a stage: no residual
    print(f"w={w:4d} @{h:3d}: {t:.3f} ms  {gb / t:.2f} TB/s   | masked store only: {t2:.3f} ms")
    del dy, g, mask, mask2
