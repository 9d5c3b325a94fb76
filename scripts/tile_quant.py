"""Time one conv layer (fwd) at several batch sizes: shows the tile-count quantisation of a one-block-per-CU kernel."""
import sys, time, torch
sys.path.insert(0, ".")
from simhand_amd import ops
cin, cout, k, s, h = map(int, sys.argv[1:6])
dtype = torch.bfloat16
for N in map(int, sys.argv[6:]):
    d = ops.conv_desc(N, h, h, cin, cout, k, k, s, k // 2, dtype)
    x = torch.randn(N, h, h, cin, device="cuda").to(dtype)
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    wk = ops.pack_krsc(w, dtype)
    for stats in (True, False):
        fn = lambda: ops.conv2d_fwd(d, x, wk, stats)
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10
        m = N * d.ho * d.wo
        blocks = ((m + 255) // 256) * (cout // 256)
        print(f"N={N} stats={stats} blocks={blocks} rounds={blocks/256:.3f} t={t*1e3:.3f} ms  {2.0*m*cout*cin*k*k/t/1e12:.0f} TF  per-round {t*1e6/(-(-blocks//256)):.1f} us")
