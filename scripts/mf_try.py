import sys, json, io, contextlib
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "3", "--warmup", "1"]
cfg = sys.stdin.read().split()
sys.path.insert(0, ".")
from simhand_amd import ops
for c in cfg:
    k, mf = map(int, c.split(":"))
    ops._lib_dev().simhand_test_conv1x1_set_rows(k, mf)
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(cfg, round(d["ms_per_step"], 2), {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items() if k.startswith("conv")})
