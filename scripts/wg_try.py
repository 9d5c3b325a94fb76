"""bench.py with other split-K block targets: usage wg_try.py <generic/1x1 target> <3x3 target>  (0 0 = defaults)."""
import sys, json, io, contextlib
a, b = int(sys.argv[1]), int(sys.argv[2])
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "10", "--warmup", "3"]
sys.path.insert(0, ".")
from simhand_amd import ops
ops._lib_dev().simhand_test_wgrad_target_blocks(a, b)
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print((a, b), round(d["ms_per_step"], 2), {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items() if k.startswith("conv")})
