"""ON THE GPU BOX: one profiled ResNet-50 step at the benchmark configuration; every library launch with its elapsed time against the time its
OWN algorithmic bytes (at 5.5 TB/s) and FLOPs (at 1.15 PFLOP/s, what the matrix-bound kernels of this tree reach) allow, sorted by the
excess.  The method that found the 2.1-ms shortcut launch of round 4 (docs/lab-notes.md).  usage: python scripts/launch_outliers.py [top]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from simhand_amd import ops  # noqa: E402

top = int(sys.argv[1]) if len(sys.argv) > 1 else 40
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.make_model(args, 1).to(dev).train()


class _T:
    max_epochs, world_size = 100, 1


model.trainer = _T()
model.setup("fit")
(opt,), _ = model.configure_optimizers()
batch = bench.device_batch(args.per_gpu_batch, args.image_size, 5, dev)


def step():
    opt.zero_grad(set_to_none=True)
    model.training_step(batch, 0)["loss"].backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
ops.prof_reset()
ops.prof_set_classes(None)
ops.prof_enable(True)
step()
torch.cuda.synchronize()
recs = ops.prof_records()
ops.prof_enable(False)
rows = []
for i, (cls, ms, fl, by) in enumerate(recs):
    bound = max(by / 5.5e12, fl / 1.15e15) * 1e3
    rows.append((ms - bound, i, cls, ms, bound, by / 1e9, fl / 1e12))
tot = sum(r[3] for r in rows)
print(f"{len(rows)} launches, {tot:.1f} ms; sum of per-launch bounds {sum(r[4] for r in rows):.1f} ms; launches sorted by time above their own bound:")
print("   # class        ms   bound ms  excess     GB   TFLOP")
for ex, i, cls, ms, bound, gb, tf in sorted(rows, reverse=True)[:top]:
    print(f"{i:4d} {cls:10s} {ms:7.3f} {bound:8.3f} {ex:7.3f} {gb:7.2f} {tf:7.3f}")
