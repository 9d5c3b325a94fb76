"""TEST-SIDE stand-in for the loss entry points of simhand_amd.ops, written with the oracle.

It exists so the collective plumbing of simhand_amd/host/dist_loss.py (row-block sharding,
all-gather order, MAX/MIN/SUM of the distance statistics, gathered negative sums) can be exercised
on CPU with the gloo backend.  Each function mimics the SEMANTICS of the C-ABI call of the same name
(include/simhand_hip.h) on CPU tensors.  It is never importable from the product package.
"""
import torch

from oracle import step as orc


class NtxentPlan:
    def __init__(self, B, b_loc, pair_off, weight_type, use_wpos, use_wneg, temperature=0.5, lambda_pos=0.0, lambda_neg=0.0, dim=128):
        self.B, self.N, self.b_loc, self.pair_off, self.rows = B, 2 * B, b_loc, pair_off, 2 * b_loc
        self.weight_type, self.use_wpos, self.use_wneg = weight_type, use_wpos, use_wneg
        self.t, self.lp, self.ln = temperature, lambda_pos, lambda_neg

    def rows_idx(self):
        a = torch.arange(self.pair_off, self.pair_off + self.b_loc)
        return torch.cat((a, a + self.B))


def _split(J, B):
    F = J.shape[1]
    j = J.view(2 * B, -1, 2) if F % 2 == 0 and F > 14 else J
    return j[:B], j[B:]


def pos_dist(J, B, mode, stats):
    j1, j2 = _split(J, B)
    d = orc.pos_distance(j1, j2, mode)
    stats[3], stats[4], stats[5] = d.max().double(), d.min().double(), d.double().sum()
    return d


def neg_dist(J, B, mode, b_loc, pair_off, stats):
    j1, j2 = _split(J, B)
    full = orc.neg_distance(j1, j2, mode)
    a = torch.arange(pair_off, pair_off + b_loc)
    D = full[torch.cat((a, a + B))].contiguous()
    stats[0], stats[1], stats[2] = D.max().double(), D.min().double(), D.double().sum()
    return D


def _w(d, wtype, dmax, dmin, mu, lam):
    if wtype == "explicit":
        return d
    if wtype == "linear":
        return (dmax - d) / (dmax - dmin)
    return 1 / (1 + torch.exp(lam * (d - mu)))


def _weights(plan, D, dpos, stats):
    wn = wp = None
    if plan.use_wneg:
        wn = _w(D, plan.weight_type, stats[0].float(), stats[1].float(), (stats[2] / (plan.N * plan.N)).float(), plan.ln)
    if plan.use_wpos:
        wp = _w(dpos, plan.weight_type, stats[3].float(), stats[4].float(), (stats[5] / plan.B).float(), plan.lp)
    return wp, wn


def ntxent_fwd(plan, Z, D, dpos, stats):
    idx = plan.rows_idx()
    wp, wn = _weights(plan, D, dpos, stats)
    s = Z[idx] @ Z.t()
    e = torch.exp((s if wn is None else s * wn) / plan.t)
    e[torch.arange(plan.rows), idx] = 0
    neg = e.sum(1)
    pair = torch.where(idx < plan.B, idx + plan.B, idx - plan.B)
    sp = (Z[idx] * Z[pair]).sum(1)
    if wp is not None:
        sp = sp * wp[idx % plan.B]
    loss = (torch.log(neg) - sp / plan.t).sum() / plan.N
    return neg, loss.reshape(1)


def ntxent_bwd(plan, Z, D, dpos, stats, neg_all, dloss):
    idx = plan.rows_idx()
    wp, wn = _weights(plan, D, dpos, stats)
    s = Z[idx] @ Z.t()
    w = torch.ones_like(s) if wn is None else wn
    e = torch.exp(w * s / plan.t)
    e[torch.arange(plan.rows), idx] = 0
    P = w * e * (1 / neg_all[idx][:, None] + 1 / neg_all[None, :])
    pair = torch.where(idx < plan.B, idx + plan.B, idx - plan.B)
    wpv = torch.ones(plan.rows) if wp is None else wp[idx % plan.B]
    dz = (P @ Z - 2 * wpv[:, None] * Z[pair]) / (plan.N * plan.t)
    return dz * dloss.reshape(())
