"""Which torch (ATen) operators does one benchmark step still launch next to the C-ABI kernels, and from where?  Groups the operators the
CPU-side profiler sees in ONE step of bench.py's workload by (operator, first stack frame inside this repository).
usage: python scripts/aten_census.py [--per-gpu-batch 1024] [bench.py flags]      (on the GPU box)"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda", 0)
model = bench.make_model(args, 1).to(dev).train()


class _T:
    max_epochs, world_size = 100, 1


model.trainer = _T()
model.setup("fit")
(opt,), (sched,) = model.configure_optimizers()
batch = bench.device_batch(args.per_gpu_batch, args.image_size, 5, dev)


def step(i):
    opt.zero_grad(set_to_none=True)
    out = model.training_step(batch, i)
    out["loss"].backward()
    opt.step()
    sched["scheduler"].step()


for i in range(2):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step(2)
    torch.cuda.synchronize()

root = os.path.abspath(".")
launching = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::div", "aten::sub",
             "aten::cat", "aten::sum", "aten::mean", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::neg", "aten::sqrt",
             "aten::index", "aten::index_select", "aten::where", "aten::eq", "aten::lt", "aten::gt", "aten::max", "aten::min",
             "aten::ones_like", "aten::zeros_like", "aten::exp", "aten::log", "aten::pow", "aten::reciprocal", "aten::rsqrt",
             "aten::div_", "aten::sub_", "aten::addcmul_", "aten::addcdiv_", "aten::lerp_", "aten::masked_fill_", "aten::stack")
count = collections.Counter()
for ev in prof.events():
    if ev.name not in launching:
        continue
    where = "?"
    for fr in ev.stack or ():
        if root in fr or "simhand_amd" in fr or "bench.py" in fr:
            where = fr.replace(root + "/", "")
            break
    shape = tuple(tuple(s) for s in (ev.input_shapes or ())[:1])
    count[(ev.name, where, shape)] += 1
for (name, where, shape), n in sorted(count.items(), key=lambda kv: -kv[1])[:80]:
    print(f"{n:5d}  {name:22s} {str(shape):28s} {where}")
