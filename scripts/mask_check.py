"""Debug helper: count ReLU-mask disagreements (sign of near-zero BN outputs) between the HIP forward and the oracle."""
import sys
import torch
sys.path.insert(0, ".")
from simhand_amd import ops
from oracle import step as orc
from tests.test_gpu_step import AUG, CASES, _product, _to_dev

mine = []
_apply = ops.bn_apply
def bn_apply(y, st, m, c, relu, residual=None, out=None):
    a = _apply(y, st, m, c, relu, residual, out)
    if relu:
        mine.append(a.float().cpu())
    return a
ops.bn_apply = bn_apply

size, b, img = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
exp, wcfg = CASES["HandCLR_W"]
batch = orc.synthetic_batch(b, size=img, seed=5)
torch.manual_seed(5)
om = orc.StepOracle(exp, size, AUG, **wcfg).train()
theirs = []
for mod in om.modules():
    if isinstance(mod, torch.nn.ReLU):
        mod.register_forward_hook(lambda m, i, o: theirs.append(o.detach()))
model = _product("HandCLR_W", size, wcfg, om, torch.float32)
loss = model.training_step(_to_dev(batch), 0)["loss"]
lo = om.contrastive_step(batch)
print(len(mine), len(theirs))
for i, (a, t) in enumerate(zip(mine, theirs)):
    if t.dim() == 4:
        t = t.permute(0, 2, 3, 1)
    t = t.reshape(a.shape)
    d = (a > 0) != (t > 0)
    print(i, tuple(a.shape), "flips", int(d.sum()), "max|a-t|", float((a - t).abs().max()), "vals", a[d][:3].tolist(), t[d][:3].tolist())
