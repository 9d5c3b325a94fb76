"""Debug helper: per-parameter gradient error of the HIP step vs the CPU oracle."""
import sys
import torch
sys.path.insert(0, ".")
from oracle import step as orc
from tests.test_gpu_step import AUG, CASES, _product, _to_dev

size, b, img = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dtype = torch.bfloat16 if len(sys.argv) > 4 and sys.argv[4] == "bf16" else torch.float32
exp, wcfg = CASES["HandCLR_W"]
batch = orc.synthetic_batch(b, size=img, seed=5)
torch.manual_seed(5)
om = orc.StepOracle(exp, size, AUG, **wcfg).train()
model = _product("HandCLR_W", size, wcfg, om, dtype)
loss = model.training_step(_to_dev(batch), 0)["loss"]
loss.backward()
lo = om.contrastive_step(batch)
lo.backward()
print("loss", loss.item(), lo.item())
og = dict(om.named_parameters())
rows = []
for k, p in model.named_parameters():
    if og[k].grad is None:
        continue
    e = (p.grad.cpu() - og[k].grad).abs().max().item()
    s = og[k].grad.abs().max().item()
    rows.append((e / max(s, 1e-30), e, s, k))
for r in rows:
    print(f"{r[0]:.3e} {r[1]:.3e} {r[2]:.3e} {r[3]}")
