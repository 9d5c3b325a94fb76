"""Repeat ONE small fp32 ResNet-50 HandCLR_W step (8 images of 64 x 64: the per-rank shard of the 4-rank synchronised-BatchNorm test)
in a single process, optionally with the synchronised-BatchNorm code path switched on (identity all-reduce), and compare every
gradient of every repetition with the first one.  A kernel-level race (timing dependent, e.g. under competing GPU processes) shows up
as a repetition that differs; the multi-rank harness (gloo, ranks sharing the GPU) is not involved.
usage: race_probe.py <repetitions> <sync 0|1> [competitors]"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import step as orc  # noqa: E402  (test infrastructure: builds the seeded weights / batch)
from simhand_amd import ops  # noqa: E402
from tests.test_gpu_step import _product  # noqa: E402

reps, sync = int(sys.argv[1]), int(sys.argv[2])
ncomp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
LOAD = ("import torch,time\nx=torch.randn(8192,8192,device='cuda',dtype=torch.bfloat16)\nwhile True:\n"
        "    for _ in range(50): y=x@x\n    torch.cuda.synchronize(); time.sleep(0.05)\n")
comps = [subprocess.Popen([sys.executable, "-c", LOAD]) for _ in range(ncomp)]
try:
    AUG = ["color_jitter", "crop", "random_crop", "resize", "rotate"]
    wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
    batch = orc.synthetic_batch(16, size=64, seed=7)
    torch.manual_seed(7)
    om = orc.StepOracle("simhand_w", "50", AUG, **wcfg).train()
    with torch.no_grad():
        for k, p in om.named_parameters():
            if k.endswith("bn3.weight"):
                p.fill_(0.1)
    model = _product("HandCLR_W", "50", wcfg, om)
    if sync:
        ops.set_bn_sync(lambda t: t)
    shard = {k: v[:4].to("cuda") for k, v in batch.items()}
    first, bad = None, 0
    for it in range(reps):
        for p in model.parameters():
            p.grad = None
        loss = model.training_step(shard, 0)["loss"]
        loss.backward()
        torch.cuda.synchronize()
        cur = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        if first is None:
            first = cur
            continue
        diff = [(k, ((cur[k] - first[k]).norm() / (first[k].norm() + 1e-30)).item()) for k in cur if not torch.equal(cur[k], first[k])]
        big = [(k, round(e, 5)) for k, e in diff if e > 1e-3]
        if it % 50 == 0:
            print(f"rep {it}: loss {loss.item():.6f}, {len(diff)} tensors not bit-equal to the first repetition", flush=True)
        if big:
            bad += 1
            print(f"rep {it}: loss {loss.item():.6f}; {len(big)} tensors differ by > 1e-3 (of {len(diff)} not bit-equal): {big[:8]}", flush=True)
    print(f"race_probe sync={sync} competitors={ncomp}: {reps} repetitions, {bad} bad; last rep: {len(diff)} tensors not bit-equal to the first")
finally:
    for c in comps:
        c.kill()
