"""Multi-step stability evidence for the mixed-precision policy: the SAME contrastive training run -- ResNet-50 handclr_w,
LARS + Adam with the reference's warm-up / cosine schedule, a fixed synthetic data set revisited every epoch, view 2 = view 1 + noise so
that there is something to learn -- in bf16 (storage bf16 / fp32 accumulate, no loss scaling) and in the fp32 parity mode, from the same
initial weights.  Logs both loss curves; exits non-zero unless every loss is finite, the bf16 loss falls, and bf16 tracks fp32.
usage: python scripts/stability_run.py [steps] [out.md]      (on the GPU box)"""
import argparse
import math
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402  (model construction shared with the benchmark)

ap = argparse.ArgumentParser()
ap.add_argument("steps", nargs="?", type=int, default=160)
ap.add_argument("out", nargs="?", default=None)
ap.add_argument("--pairs", type=int, default=64)
ap.add_argument("--size", type=int, default=112)
ap.add_argument("--batches", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda", 0)


def dataset():
    g = torch.Generator(device=dev).manual_seed(11)
    out = []
    for _ in range(a.batches):
        b = bench.device_batch(a.pairs, a.size, int(torch.randint(0, 1 << 30, (1,), generator=g, device=dev)), dev)
        b["transformed_image2"] = b["transformed_image1"] + 0.5 * torch.randn(b["transformed_image1"].shape, generator=g, device=dev)
        out.append(b)
    return out


def run(precision):
    args = SimpleNamespace(experiment="handclr_w", resnet="50", per_gpu_batch=a.pairs, precision=precision)
    model = bench.make_model(args, 1).to(dev).train()
    for kv in filter(None, os.environ.get("SIMHAND_ENGINE", "").split(",")):  # e.g. SIMHAND_ENGINE=fold_bn3=0,chain_conv1=0 (diagnostics)
        k, v = kv.split("=")
        setattr(model.encoder.engine, k, bool(int(v)))

    class _T:
        max_epochs, world_size = math.ceil(a.steps / a.batches), 1

    model.trainer = _T()
    model.setup("fit")
    (opt,), (sched,) = model.configure_optimizers()
    data = dataset()
    from simhand_amd.host.amp import GradScaler

    scaler = GradScaler(enabled=precision == "16")  # precision 16 = fp16 storage + the reference's native-AMP loss scaling
    params = list(model.parameters())
    losses = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        opt.zero_grad(set_to_none=True)
        loss = model.training_step(data[i % a.batches], i)["loss"]
        scaler.scale(loss).backward()
        scaler.unscale_(params)
        scaler.step(opt)
        scaler.update()
        sched["scheduler"].step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    if precision == "16":
        print(f"product fp16 run: {scaler.skipped_steps} skipped steps, final scale {scaler.get_scale():g}", flush=True)
    pmax = max(float(p.detach().abs().max()) for p in model.parameters())
    return losses, pmax, time.perf_counter() - t0


def _sr_bf16(x):
    """fp32 -> bf16 with STOCHASTIC rounding (uniform 16 random bits added to the magnitude before the truncation), back as fp32."""
    bits = x.contiguous().view(torch.int32)
    rnd = torch.randint(0, 1 << 16, x.shape, device=x.device, dtype=torch.int32)
    return ((bits + rnd) & -65536).view(torch.float32)


class _Round(torch.autograd.Function):
    """x -> storage dtype -> fp32 in the forward (if fwd) and / or the same rounding of the gradient in the backward (if bwd).
    fwd_dt: another storage type for the forward value (mitigation twins); sr: stochastic instead of nearest-even rounding (bf16 forward)."""

    @staticmethod
    def forward(ctx, x, dt, fwd, bwd, fwd_dt=None, sr=False):
        ctx.dt, ctx.bwd = dt, bwd
        if not fwd:
            return x
        if sr:
            return _sr_bf16(x)
        return x.to(fwd_dt or dt).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.dt).to(torch.float32) if ctx.bwd else g), None, None, None, None, None


def run_oracle(storage=None, parts="wag", scaler=False, act_storage=None, act_sr=False):
    """The CHECKER's curve: oracle.StepOracle (plain torch ops) on the same device, same initial weights / data / optimizer / schedule.
    storage = torch.bfloat16 / torch.float16: its encoder convolutions see storage-rounded operands -- the numerics model of a 16-bit
    storage path with fp32 accumulators, independent of any HIP kernel.  parts: which tensors are rounded -- "w" the conv weights
    (forward only: the weight GRADIENT stays fp32, as in the HIP path), "a" the conv inputs / outputs in the forward, "g" the
    gradients crossing those same edges in the backward.  scaler: torch.cuda.amp.GradScaler semantics around the backward (the
    reference's precision=16 policy, src/experiments/main.py:158-159: scale 2^16, halve + skip the step on inf / nan, double every
    2000 clean steps) -- needed for fp16's 5-bit exponent, pointless for bf16.  Mitigation twins: act_storage = another
    storage type for the FORWARD activations only (fp16 activations next to bf16 weights / gradients); act_sr = stochastic rounding of
    the forward activations' bf16 stores (the tensor the decomposition blames)."""
    import torch.nn as nn
    import torch.nn.functional as F

    from oracle import step as orc

    torch.backends.cudnn.allow_tf32 = False
    args = SimpleNamespace(experiment="handclr_w", resnet="50", per_gpu_batch=a.pairs, precision="32")
    prod = bench.make_model(args, 1)
    om = orc.StepOracle("simhand_w", "50", bench.AUG, weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
    om.load_state_dict(prod.state_dict(), strict=True)
    om = om.to(dev).train()
    if storage is not None:
        ra = lambda t: _Round.apply(t, storage, "a" in parts, "g" in parts, act_storage, act_sr)  # noqa: E731
        rw = lambda t: _Round.apply(t, storage, "w" in parts, False)         # noqa: E731
        for m in om.encoder.modules():
            if isinstance(m, nn.Conv2d):
                m.forward = (lambda x, m=m: ra(F.conv2d(ra(x), rw(m.weight), None, m.stride, m.padding)))

    class _T:
        max_epochs, world_size = math.ceil(a.steps / a.batches), 1

    prod.trainer = _T()
    prod.setup("fit")
    # the product's own optimizer / schedule construction, on the oracle's parameters (same names -> same groups)
    named = dict(om.named_parameters())
    prod.named_parameters = lambda *aa, **kk: iter(named.items())
    (opt,), (sched,) = prod.configure_optimizers()
    data = dataset()
    losses = []
    scale, clean, skipped = 65536.0, 0, 0
    for i in range(a.steps):
        opt.zero_grad(set_to_none=True)
        loss = om.contrastive_step(data[i % a.batches])
        if scaler:
            (loss * scale).backward()
            grads = [p.grad for p in om.parameters() if p.grad is not None]
            finite = all(bool(torch.isfinite(g).all()) for g in grads)
            if finite:
                for g in grads:
                    g.div_(scale)
                opt.step()
                clean += 1
                if clean % 2000 == 0:
                    scale *= 2.0
            else:  # GradScaler.step skips the optimizer, update() halves the scale
                scale *= 0.5
                clean = 0
                skipped += 1
        else:
            loss.backward()
            opt.step()
        sched["scheduler"].step()
        losses.append(float(loss.detach()))
    if scaler:
        print(f"GradScaler twin: {skipped} skipped steps, final scale {scale:g}", flush=True)
    return losses


bf, bf_pmax, bf_s = run("bf16")
fp, fp_pmax, fp_s = run("32")
h16, h16_pmax, h16_s = run("16")
ora = ora_bf = None
twins = {}
if os.environ.get("SIMHAND_STABILITY_ORACLE", "0") == "1":
    ora, ora_bf = run_oracle(None), run_oracle(torch.bfloat16)
    # the reference's own precision policy (fp16 storage under autocast + GradScaler) and the decomposition of the bf16 lag
    twins = {"fp16 storage + GradScaler (the reference's precision=16)": run_oracle(torch.float16, "wag", scaler=True),
             "bf16 weights only": run_oracle(torch.bfloat16, "w"),
             "bf16 forward activations only": run_oracle(torch.bfloat16, "a"),
             "bf16 backward gradients only": run_oracle(torch.bfloat16, "g"),
             # mitigations for the bf16 lag (it is the forward activations' 8-bit significand): do they close the gap to fp32?
             "MITIGATION bf16 storage, forward activations rounded STOCHASTICALLY": run_oracle(torch.bfloat16, "wag", act_sr=True),
             "MITIGATION fp16 forward activations, bf16 weights + gradients (no scaler)": run_oracle(torch.bfloat16, "wag", act_storage=torch.float16)}
lines = [f"# {a.steps} training steps, ResNet-50 handclr_w, {a.batches} fixed batches of {a.pairs} pairs @ {a.size}^2 revisited every epoch, LARS + Adam, "
         "linear warm-up + cosine schedule; same initial weights", "",
         "| step | loss bf16 (bf16 storage, fp32 accumulate / statistics / loss / optimizer, no loss scaling) | loss fp32 parity mode | bf16 / fp32 |"
         " loss precision=16 (fp16 storage build + GradScaler: the reference's policy) |"
         + (" oracle fp32 (torch ops) | oracle with bf16-rounded conv weights / inputs / outputs |" if ora else "")
         + "".join(f" oracle twin: {k} |" for k in twins), "|---|---|---|---|---|" + ("---|---|" if ora else "") + "---|" * len(twins)]
for i in list(range(0, a.steps, max(1, a.steps // 16))) + [a.steps - 1]:
    lines.append(f"| {i} | {bf[i]:.4f} | {fp[i]:.4f} | {bf[i] / fp[i]:.4f} | {h16[i]:.4f} |" + (f" {ora[i]:.4f} | {ora_bf[i]:.4f} |" if ora else "")
                 + "".join(f" {v[i]:.4f} |" for v in twins.values()))
k = max(1, a.steps // 10)
head_b, tail_b, tail_f = sum(bf[:k]) / k, sum(bf[-k:]) / k, sum(fp[-k:]) / k
tail_h = sum(h16[-k:]) / k
lines += ["", f"mean of the first {k} losses (bf16) {head_b:.4f}; mean of the last {k}: bf16 {tail_b:.4f}, fp32 {tail_f:.4f}, precision=16 (fp16) {tail_h:.4f}; "
          f"max |parameter| bf16 {bf_pmax:.3f} / fp32 {fp_pmax:.3f}; wall {bf_s:.1f} s / {fp_s:.1f} s"]
if ora:
    lines.append(f"oracle, mean of the last {k}: fp32 {sum(ora[-k:]) / k:.4f}, bf16-storage twin {sum(ora_bf[-k:]) / k:.4f}")
for name, v in twins.items():
    lines.append(f"oracle twin, mean of the last {k}: {name}: {sum(v[-k:]) / k:.4f}")
text = "\n".join(lines)
print(text)
if a.out:
    open(a.out, "w").write(text + "\n")
# gates: everything finite, bf16 learns and stays within 10 % of fp32, and the REFERENCE's policy (fp16 + GradScaler) tracks fp32 to 2.5 %
ok = (all(math.isfinite(v) for v in bf + fp + h16) and tail_b < head_b - 0.05 and abs(tail_b - tail_f) <= 0.1 * abs(tail_f) + 0.05
      and abs(tail_h - tail_f) <= 0.025 * abs(tail_f))
sys.exit(0 if ok else 1)
