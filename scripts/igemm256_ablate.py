"""Times fwd / dgrad of a few MFMA-bound ResNet-50 layers with each ablation build (scripts/igemm_ablate.sh)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time
    import torch
    sys.path.insert(0, os.path.dirname(HERE))
    from simhand_amd import _lib
    _lib.LIB_PATH = sys.argv[2]
    from simhand_amd import ops
    N = 2048
    dtype = torch.bfloat16
    out = []
    for cin, cout, k, s, h in [(256, 256, 3, 1, 14), (1024, 256, 1, 1, 14), (512, 512, 3, 1, 7), (2048, 512, 1, 1, 7), (512, 2048, 1, 1, 7)]:
        d = ops.conv_desc(N, h, h, cin, cout, k, k, s, k // 2, dtype)
        x = torch.randn(N, h, h, cin, device="cuda").to(dtype)
        w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        wk = ops.pack_krsc(w, dtype)
        fn = lambda: ops.conv2d_fwd(d, x, wk, True)
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10
        fl = 2.0 * N * d.ho * d.wo * cout * cin * k * k
        out.append(f"{t*1e3:6.3f}ms {fl/t/1e12:5.0f}TF")
    print(" | ".join(out))
    sys.exit(0)

names = {0: "baseline", 2: "no A DMAs", 3: "no B DMAs", 4: "no DMA", 5: "no output stores", 6: "no sched_barrier"}
print("layers: 3x3 256@14 | 1x1 1024->256@14 | 3x3 512@7 | 1x1 2048->512@7 | 1x1 512->2048@7")
for n in [int(v) for v in os.environ.get("ABL_SET", "0 4 5").split()]:
    lib = os.path.join(HERE, "abl", f"libabl_{n}.so")
    r = subprocess.run([sys.executable, __file__, "child", lib], capture_output=True, text=True)
    print(f"{names[n]:18s}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}")
