"""ON THE GPU BOX: the kernel routes the production dispatch takes for every Appendix-C shape x {forward, data gradient, weight gradient} at
1024 and 4096 images (BASELINE configs[3] / configs[4] per-GPU batches) -> tests/golden/fullsize_routes.json (copied back through
gpurun_out/).  tests/test_gpu_fullsize.py::test_other_batches_* asserts the dispatch still takes them.
usage: python scripts/dump_fullsize_routes.py > gpurun_out/fullsize_routes.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simhand_amd import ops  # noqa: E402
from tests.test_gpu_fullsize import DT, SHAPES, _operands_n, _routes  # noqa: E402

out = {}
for n in (1024, 4096):
    for shape in SHAPES:
        cin, cout, k, s, h = shape
        o = _operands_n(shape, n, ("x", "dy"))
        d = o["d"]
        ops.hooks_reset()
        rs = []
        ops.route_reset()
        ops.conv2d_fwd(d, o["x"], ops.pack_krsc(o["w"], DT), want_stats=True)
        torch.cuda.synchronize()
        rs.append(_routes(ops))
        ops.route_reset()
        ops.conv2d_dgrad(d, o["dy"], ops.pack_crsk(o["w"], DT))
        torch.cuda.synchronize()
        rs.append(_routes(ops))
        ops.route_reset()
        ops.conv2d_wgrad_oihw(d, o["x"], o["dy"], (cout, cin, k, k))
        torch.cuda.synchronize()
        rs.append(_routes(ops))
        out[f"{n}:" + "x".join(map(str, shape))] = rs
        del o
        torch.cuda.empty_cache()
print(json.dumps(out, indent=0, sort_keys=True))
