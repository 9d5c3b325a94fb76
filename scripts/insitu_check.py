"""Debug helper: wrap ops.* so every backbone kernel call inside a real step is re-computed with torch
CPU ops from the same inputs and the discrepancy is printed (finds which call site misbehaves)."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from simhand_amd import ops
from oracle import step as orc
from tests.test_gpu_step import AUG, CASES, _product, _to_dev

calls = []
def rel(a, b):
    s = b.abs().max().item()
    return (a - b).abs().max().item() / max(s, 1e-30)

_bn_bwd = ops.bn_backward
def bn_backward(da, a, y, st, gamma, m, c, relu, want_dres):
    out = _bn_bwd(da, a, y, st, gamma, m, c, relu, want_dres)
    dy, dres, dg, db = out
    g = da.float().reshape(m, c).cpu().double()
    if relu:
        g = g * (a.float().reshape(m, c).cpu() > 0)
    yy = y.float().reshape(m, c).cpu().double()
    mean, invstd = st.mean.cpu().double(), st.invstd.cpu().double()
    xh = (yy - mean) * invstd
    db_ref, dg_ref = g.sum(0), (g * xh).sum(0)
    dy_ref = gamma.cpu().double() * invstd * (g - db_ref / m - xh * dg_ref / m)
    # also: are mean/invstd the true batch stats of y?
    mean_true = yy.mean(0); var_true = yy.var(0, unbiased=False)
    print(f"bn_bwd m={m} c={c} relu={relu} dres={want_dres}: dbeta {rel(db.cpu().double(), db_ref):.2e} dgamma {rel(dg.cpu().double(), dg_ref):.2e} "
          f"dy {rel(dy.float().reshape(m,c).cpu().double(), dy_ref):.2e} mean {rel(mean, mean_true):.2e} invstd {rel(invstd, 1/torch.sqrt(var_true+1e-5)):.2e}")
    if relu and not want_dres:
        yf = y.float().reshape(m, c).cpu()
        o = F.batch_norm(yf, None, None, gamma.cpu(), None, training=True, eps=1e-5)  # beta not passed: use shift
        beta = (st.shift.cpu() + st.mean.cpu() * st.scale.cpu())
        o = o + beta
        mine = a.float().reshape(m, c).cpu() > 0
        diff = (mine != (o > 0))
        if diff.any():
            idx = diff.nonzero()
            print('   RELU MASK FLIPS:', idx.shape[0], 'bn out there:', o[diff][:4].tolist(), 'da there:', da.float().reshape(m,c).cpu()[diff][:4].tolist())
    return out
ops.bn_backward = bn_backward

_dgrad = ops.conv2d_dgrad
def conv2d_dgrad(d, dy, wt, dx=None, accumulate=False):
    base = dx.float().cpu().clone() if accumulate else None
    out = _dgrad(d, dy, wt, dx=dx, accumulate=accumulate)
    w = wt.float().cpu().view(d.cin, d.r, d.s, d.cout).permute(3, 0, 1, 2).contiguous()  # OIHW
    dyc = dy.float().cpu().reshape(d.n, d.ho, d.wo, d.cout).permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_input((d.n, d.cin, d.h, d.w), w, dyc, stride=d.stride, padding=d.pad).permute(0, 2, 3, 1)
    if accumulate:
        ref = ref + base.reshape(ref.shape)
    print(f"dgrad n={d.n} {d.h}x{d.w} cin={d.cin} cout={d.cout} k={d.r} s={d.stride} acc={accumulate}: {rel(out.float().cpu().reshape(ref.shape), ref):.2e}")
    return out
ops.conv2d_dgrad = conv2d_dgrad

_wgrad = ops.conv2d_wgrad
def conv2d_wgrad(d, x, dy):
    out = _wgrad(d, x, dy)
    xc = x.float().cpu().reshape(d.n, d.h, d.w, d.cin).permute(0, 3, 1, 2)
    dyc = dy.float().cpu().reshape(d.n, d.ho, d.wo, d.cout).permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(xc, (d.cout, d.cin, d.r, d.s), dyc, stride=d.stride, padding=d.pad)
    ref = ref.permute(0, 2, 3, 1).reshape(d.cout, -1)
    print(f"wgrad n={d.n} {d.h}x{d.w} cin={d.cin} cout={d.cout} k={d.r} s={d.stride}: {rel(out.cpu(), ref):.2e}")
    return out
ops.conv2d_wgrad = conv2d_wgrad

size, b, img = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
exp, wcfg = CASES["HandCLR_W"]
batch = orc.synthetic_batch(b, size=img, seed=5)
torch.manual_seed(5)
om = orc.StepOracle(exp, size, AUG, **wcfg).train()
model = _product("HandCLR_W", size, wcfg, om, torch.float32)
loss = model.training_step(_to_dev(batch), 0)["loss"]
loss.backward()
