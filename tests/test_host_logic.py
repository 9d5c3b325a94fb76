"""CPU: host-side logic that needs no kernel launch -- CLI / config surface against the values captured
from the reference's own parser (tests/golden/cli.json), model registry, state_dict keys, weight-decay
split quirk, LR schedule / LARS restatement, checkpoint naming, and the gloo world-size-2 runs of the
sharded loss and gradient all-reduce (with the test-side kernel stand-in of tests/_cpu_kernels.py)."""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cli(golden_dir):
    return json.load(open(os.path.join(golden_dir, "cli.json")))


@pytest.mark.parametrize("name", ["handclr_w", "peclr_w", "simclr_w"])
def test_cli_surface_matches_reference_parser(cli, name):
    from simhand_amd.host import config as C
    from simhand_amd.host import experiments_utils as eu
    from simhand_amd.host.config import edict, read_json

    ref = cli[name]
    args = eu.get_general_args("x", ref["argv"])
    got = {k: v for k, v in vars(args).items() if k not in eu.BUILD_ONLY_FLAGS}
    assert got == ref["args"]
    train_param = eu.update_train_params(args, edict(read_json(C.TRAINING_CONFIG_PATH)))
    assert json.loads(json.dumps(train_param)) == ref["train_param"]
    model_param = edict(read_json(eu.model_config_path(name)))
    model_param = eu.update_model_params(model_param, args, 1000000, train_param)
    model_param.augmentation = [k for k, v in train_param.augmentation_flags.items() if v]
    assert json.loads(json.dumps(model_param)) == ref["model_param"]
    assert eu.prepare_name(f"{args.experiment_type}_", train_param) == ref["experiment_name"]


def test_weight_flag_assertions():
    from simhand_amd.host import experiments_utils as eu
    from simhand_amd.host.config import edict

    tp = edict(batch_size=8, accumulate_grad_batches=1)
    for bad in (["--weight_type", "cubic", "--joints_type", "augmented", "--diff_type", "mpjpe", "--pos_neg", "pos"],
                ["--weight_type", "non_linear", "--joints_type", "augmented", "--diff_type", "mpjpe", "--pos_neg", "pos",
                 "--non_linear_lambda_pos", "3.0", "--non_linear_lambda_neg", "0.05"]):
        with pytest.raises(AssertionError):
            eu.update_model_params(edict(), eu.get_general_args("x", bad), 10, tp)


def test_registry(cli):
    from simhand_amd.host import experiments_utils as eu

    for key, cls in cli["get_model"].items():
        got = eu.get_model(key)
        if key == "handclr_w":  # unreachable in the reference (falls off the if-chain) -> alias here
            assert cls is None and got.__name__ == "HandCLR_W"
        elif cls in ("SiMHand_W", "HandCLR_W"):
            assert got.__name__ == "HandCLR_W"
        elif cls in ("SiMHand_VIS", "HandCLR_VIS"):
            assert got.__name__ == "HandCLR_VIS"
        else:
            assert got.__name__ == cls, key
    assert eu.get_model("nope") is None
    with pytest.raises(ValueError):
        eu.model_config_path("supervised")


def _cfg(size="18", **kw):
    from simhand_amd.host.config import edict

    base = dict(resnet_size=size, projection_head_input_dim=2048, projection_head_hidden_dim=512, output_dim=128, augmentation=[],
                lr=1e-4, opt_weight_decay=1e-6, warmup_epochs=10, num_of_mini_batch=1, optimizer="LARS", batch_size=128, num_samples=12800,
                weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg", joints_type="augmented", use_pca=False)
    base.update(kw)
    return edict(base)


def test_state_dict_keys_and_seeded_init_match_oracle(golden_dir):
    from oracle import step as orc
    from simhand_amd.host import unsupervised

    meta = json.load(open(os.path.join(golden_dir, "step_rn18.json")))["HandCLR_W"]
    torch.manual_seed(5)
    m = unsupervised.HandCLR_W(_cfg("18"), None, "train")
    assert list(m.state_dict().keys()) == meta["state_dict_keys"]
    with torch.no_grad():
        assert float(sum(p.double().abs().sum() for p in m.parameters())) == pytest.approx(meta["param_checksum"], rel=1e-9)
    torch.manual_seed(7)
    a = unsupervised.PeCLR_W(_cfg("50"), None, "train").state_dict()
    torch.manual_seed(7)
    b = orc.StepOracle("peclr_w", "50").state_dict()
    assert list(a.keys()) == list(b.keys()) and all(torch.equal(a[k], b[k]) for k in a)
    assert a["projection_head.0.weight"].shape == (512, 2048) and a["encoder.final_layer.0.weight"].shape == (64, 2048)


def test_weight_decay_split_quirk_and_optimizer_setup():
    from oracle.optim import exclude_from_wt_decay
    from simhand_amd.host import unsupervised

    m = unsupervised.HandCLR_W(_cfg("18"), None, "train")
    names = [n for n, _ in m.named_parameters()]
    groups = m.exclude_from_wt_decay(m.named_parameters(), 1e-6)
    decay, no_decay = exclude_from_wt_decay(names)
    assert len(groups[0]["params"]) == len(decay) and len(groups[1]["params"]) == len(no_decay)
    assert "encoder.features.1.weight" in decay          # stem BN weight is NOT matched by "bn" -> decayed (quirk kept)
    assert "encoder.features.4.0.bn1.weight" in no_decay
    assert "encoder.features.5.0.downsample.1.weight" in decay and "encoder.features.5.0.downsample.1.bias" in no_decay

    class T:
        max_epochs, world_size = 50, 1

    m.trainer = T()
    m.setup("fit")
    assert m.train_iters_per_epoch == 100
    (opt,), (sch,) = m.configure_optimizers()
    assert sch["interval"] == "step"
    assert opt.defaults["lr"] == pytest.approx(1e-4 * math.sqrt(1024))
    assert opt.param_groups[0]["lr"] == 0.0  # warmup_start_lr = 0 at step 0
    s = sch["scheduler"]
    assert s.warmup_epochs == 1000 and s.max_epochs == 5000


def test_lr_schedule_matches_restated_pl_bolts():
    from oracle.optim import linear_warmup_cosine_lr
    from simhand_amd.host.optim import LinearWarmupCosineAnnealingLR

    class FakeOpt:
        param_groups = [{"lr": 3.2e-3}]

    s = LinearWarmupCosineAnnealingLR(FakeOpt(), warmup_epochs=10, max_epochs=100)
    for t in range(0, 100):
        assert FakeOpt.param_groups[0]["lr"] == pytest.approx(linear_warmup_cosine_lr(t, 3.2e-3, 10, 100), abs=1e-12)
        s.step()
    assert linear_warmup_cosine_lr(9, 3.2e-3, 10, 100) == pytest.approx(3.2e-3)
    assert linear_warmup_cosine_lr(100, 3.2e-3, 10, 100) == pytest.approx(0.0, abs=1e-12)


def test_checkpoint_filename_template():
    from simhand_amd.host.lightning import ModelCheckpoint

    ck = ModelCheckpoint(save_top_k=3, monitor="contrastive_loss", mode="min",
                         filename="handclr_w_pretrain_{epoch:02d}_train_['ego4d-1m']_bs_8.0_1024_lr_3.2e-03_{contrastive_loss:.6f}")
    assert ck.format_name(7, {"contrastive_loss": 6.123456789}) == \
        "handclr_w_pretrain_epoch=07_train_['ego4d-1m']_bs_8.0_1024_lr_3.2e-03_contrastive_loss=6.123457.ckpt"


def test_checkpoint_pruning_never_adopts_foreign_files(tmp_path):
    """A resumed run prunes to save_top_k over ITS OWN files: the list its checkpoint recorded (exact scores), or -- for checkpoints
    without one -- files whose whole name matches this callback's template; another run's file in the same directory that merely
    contains 'contrastive_loss=<float>' is neither adopted nor deleted."""
    from simhand_amd.host.lightning import ModelCheckpoint

    tmpl = "handclr_w_pretrain_{epoch:02d}_train_['ego4d-1m']_bs_8.0_1024_lr_3.2e-03_{contrastive_loss:.6f}"
    mine = ModelCheckpoint(save_top_k=2, monitor="contrastive_loss", mode="min", filename=tmpl, dirpath=str(tmp_path))
    own = [tmp_path / mine.format_name(e, {"contrastive_loss": v}) for e, v in ((0, 6.5), (1, 6.25))]
    foreign = [tmp_path / "peclr_w_pretrain_epoch=03_train_['ego4d-1m']_bs_8.0_1024_lr_3.2e-03_contrastive_loss=0.100000.ckpt",
               tmp_path / ("x" + own[0].name)]
    for f in own + foreign:
        f.write_bytes(b"")
    mine.rescan()
    assert sorted(p for _, p in mine.kept) == sorted(str(f) for f in own) and mine.best_model_path == str(own[1])

    class _T:
        global_rank = 0

        def checkpoint_dict(self, module, epoch):
            return {"kept": list(mine.kept)}

    mine.on_epoch_end(_T(), None, 2, {"contrastive_loss": 6.0})   # a better third file: the worst of OUR files goes, the foreign ones stay
    assert not own[0].exists() and own[1].exists() and all(f.exists() for f in foreign)
    assert len(mine.kept) == 2 and mine.best_model_path.endswith("contrastive_loss=6.000000.ckpt")
    # the saved dict lists the file being written (a resume from it restores exactly this set, with exact scores)
    import torch

    rec = torch.load(mine.best_model_path, weights_only=False)["kept"]
    again = ModelCheckpoint(save_top_k=2, monitor="contrastive_loss", mode="min", filename=tmpl, dirpath=str(tmp_path))
    again.restore(rec, "")
    assert again.kept == mine.kept and again.best_model_path == mine.best_model_path
    # a resumed run writing to ANOTHER directory starts its own top-k: the run it was started from is never pruned (GPU test
    # tests/test_gpu_main.py resumes from b/checkpoints into c/checkpoints and then exports the checkpoint it resumed from)
    (tmp_path / "elsewhere").mkdir()
    other = ModelCheckpoint(save_top_k=1, monitor="contrastive_loss", mode="min", filename=tmpl, dirpath=str(tmp_path / "elsewhere"))
    other.restore(rec, "")
    assert other.kept == [] and other.best_model_path == ""


def test_checkpoint_rewritten_file_keeps_one_entry_and_is_never_pruned_under_its_better_score(tmp_path):
    """A resumed run that re-runs an epoch rewrites a filename its restored kept-list already holds (default template '{epoch:02d}').
    There must be ONE entry per path afterwards, and pruning the worse entry must not delete the file the better one names."""
    import torch
    from simhand_amd.host.lightning import ModelCheckpoint

    class _T:
        global_rank = 0

        def checkpoint_dict(self, module, epoch):
            return {"epoch": epoch}

    ck = ModelCheckpoint(save_top_k=2, monitor="contrastive_loss", mode="min", dirpath=str(tmp_path))
    ck.on_epoch_end(_T(), None, 0, {"contrastive_loss": 6.5})
    ck.on_epoch_end(_T(), None, 1, {"contrastive_loss": 6.4})
    resumed = ModelCheckpoint(save_top_k=2, monitor="contrastive_loss", mode="min", dirpath=str(tmp_path))
    resumed.restore(list(ck.kept), ck.best_model_path)
    resumed.on_epoch_end(_T(), None, 1, {"contrastive_loss": 6.0})   # epoch 1 again, better: same file name 'epoch=01.ckpt'
    paths = [p for _, p in resumed.kept]
    assert len(paths) == len(set(paths)) == 2
    assert resumed.best_model_path.endswith("epoch=01.ckpt") and resumed.kept[0][0] == 6.0
    resumed.on_epoch_end(_T(), None, 2, {"contrastive_loss": 6.2})   # pushes out epoch 0, must not touch epoch 1's file
    assert (tmp_path / "epoch=01.ckpt").exists() and not (tmp_path / "epoch=00.ckpt").exists()
    assert torch.load(resumed.best_model_path, weights_only=False)["epoch"] == 1


def test_sharding_helpers():
    from simhand_amd.host.dist import shard_pairs

    assert shard_pairs(8192, 3, 8) == (3072, 1024)
    with pytest.raises(ValueError):
        shard_pairs(10, 0, 4)


def test_main_requires_synthetic_and_rejects_unknown_models(monkeypatch):
    from simhand_amd.host import main as M

    with pytest.raises(ValueError):
        M.main(["--experiment_type", "supervised", "--synthetic"])
    with pytest.raises(SystemExit):
        M.main(["--experiment_type", "handclr_w", "-sources", "ego4d", "-batch_size", "8"])


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from simhand_amd.host import dist as shdist
from simhand_amd.host.dist_loss import LossConfig, ShardedNtxent
from tests import _cpu_kernels
from oracle import step as orc
rank, local, world = shdist.init_from_env("gloo")
B = 12
g = torch.Generator().manual_seed(3)
z1 = torch.nn.functional.normalize(torch.randn(B, 128, generator=g)); z2 = torch.nn.functional.normalize(torch.randn(B, 128, generator=g))
j1 = torch.rand(B, 21, 2, generator=g) * 224; j2 = j1 + torch.randn(B, 21, 2, generator=g) * 8
off, b = shdist.shard_pairs(B, rank, world)
for wt, diff, mode in (("linear", "mpjpe", "pos_neg"), ("non_linear", "w_abs", "neg"), (None, "mpjpe", "pos_neg"), ("linear", "w_o_abs", "pos")):
    cfg = LossConfig(weight_type=wt, diff_type=diff, use_wpos=wt is not None and mode in ("pos_neg", "pos"),
                     use_wneg=wt is not None and mode in ("pos_neg", "neg"), lambda_pos=2.5, lambda_neg=0.01, kernels=_cpu_kernels)
    zl = torch.cat((z1[off:off + b], z2[off:off + b])).clone().requires_grad_(True)
    jl = torch.cat((j1[off:off + b], j2[off:off + b])).reshape(2 * b, -1)
    loss = ShardedNtxent.apply(zl, jl if wt else None, cfg, None, None, None)
    loss.backward()
    wp = wn = None
    if wt == "linear": wp, wn = orc.weights_linear(j1, j2, diff)
    if wt == "non_linear": wp, wn = orc.weights_nonlinear(j1, j2, 2.5, 0.01, diff)
    if mode == "pos": wn = None
    if mode == "neg": wp = None
    a, c = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    want = orc.ntxent(a, c, wp, wn); want.backward()
    assert abs(loss.item() - want.item()) < 1e-5 * abs(want.item()), (rank, wt, loss.item(), want.item())
    gw = torch.cat((a.grad[off:off + b], c.grad[off:off + b]))
    assert (zl.grad - gw).abs().max() < 1e-4 * gw.abs().max() + 1e-7, (rank, wt)
# gradient all-reduce: per-rank gradients ADD to the single-process gradient
p = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(3, 2))]
p[0].grad = torch.full((5,), float(rank + 1)); p[1].grad = torch.full((3, 2), 10.0 * (rank + 1))
shdist.allreduce_gradients(p, bucket_bytes=16)
tot = sum(range(1, world + 1))
assert torch.equal(p[0].grad, torch.full((5,), float(tot))) and torch.equal(p[1].grad, torch.full((3, 2), 10.0 * tot))
# the agreed gradient pattern is cached: the second call issues no MAX all-reduce / host read, and re-points p.grad at bucket views
p[0].grad = torch.full((5,), float(rank + 1)); p[1].grad = None     # a rank-local missing gradient is zero-filled from the cached plan
calls = []
orig = dist.all_reduce
dist.all_reduce = lambda t, *a, **k: (calls.append(k.get("op", a[0] if a else None)), orig(t, *a, **k))[1]
shdist.allreduce_gradients(p, bucket_bytes=16)
dist.all_reduce = orig
assert dist.ReduceOp.MAX not in calls, calls
assert torch.equal(p[0].grad, torch.full((5,), float(tot))) and torch.equal(p[1].grad, torch.zeros(3, 2))
r = torch.nn.Parameter(torch.zeros(4)); r.requires_grad_(True)
# overlapped reducer: groups handed over "block by block" go out as asynchronous buckets; finish() returns the reduced gradients as
# views of the buckets (no copy back); what it reduced is skipped (once) by allreduce_gradients
q = [torch.nn.Parameter(torch.zeros(n)) for n in (7, 3, 11, 2)]
gr = [torch.full((p_.numel(),), float((rank + 1) * (i + 1))) for i, p_ in enumerate(q)]
red = shdist.OverlappedGradReducer(bucket_bytes=32)
assert red.active()
red.submit([(q[0], gr[0]), (q[1], gr[1])]); red.submit([(q[2], gr[2])]); out = red.finish()
assert set(out) == {q[0], q[1], q[2]}
for i in range(3):
    assert torch.equal(out[q[i]], torch.full_like(gr[i], float(tot * (i + 1)))), (rank, i, out[q[i]])
    q[i].grad = out[q[i]]
assert out[q[0]].untyped_storage().data_ptr() == out[q[1]].untyped_storage().data_ptr()   # one bucket, two views
q[3].grad = gr[3]
assert red.reduced == {id(q[0]), id(q[1]), id(q[2])}
shdist.allreduce_gradients(q, bucket_bytes=16, skip=red.reduced)
assert not red.reduced
assert torch.equal(q[0].grad, torch.full((7,), float(tot))) and torch.equal(q[3].grad, torch.full((2,), 4.0 * tot))
# bf16 wire format: same sums where they are exactly representable
w = [torch.nn.Parameter(torch.zeros(6))]
w[0].grad = torch.full((6,), 0.5 * (rank + 1))
shdist.allreduce_gradients(w, wire="bf16")
assert w[0].grad.dtype == torch.float32 and torch.equal(w[0].grad, torch.full((6,), 0.5 * tot))
red2 = shdist.OverlappedGradReducer(bucket_bytes=8, wire="bf16")
g2 = torch.full((5,), 0.25 * (rank + 1)); red2.submit([(w[0], g2)]); o2 = red2.finish()
assert o2[w[0]].dtype == torch.float32 and torch.equal(o2[w[0]], torch.full((5,), 0.25 * tot))
# a gradient outside the agreed pattern is an error, not a silent desynchronisation
s2 = [torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2))]
s2[0].grad = torch.ones(2)
shdist.allreduce_gradients(s2)
s2[1].grad = torch.ones(2)
try:
    shdist.allreduce_gradients(s2)
    raise SystemExit("expected a RuntimeError")
except RuntimeError:
    pass
# synchronised BatchNorm switch (SURVEY 8e, optional): the callable it installs SUMs over the ranks; the packed forward message
# (2 C sums + the position count, double) gives every rank the global mean / variance of ragged shards
from simhand_amd import ops
assert not ops.bn_sync_active()
assert shdist.enable_sync_bn() and ops.bn_sync_active()
gx = torch.Generator().manual_seed(11)
full = torch.randn(40, 6, generator=gx, dtype=torch.float64) * 3 + 1
cuts = [0] + [40 * (r + 1) // world - (1 if r + 1 < world else 0) * (r % 2) for r in range(world)]   # ragged shards
mine = full[cuts[rank]:cuts[rank + 1]]
msg = torch.cat((mine.sum(0), (mine * mine).sum(0), torch.tensor([float(mine.shape[0])], dtype=torch.float64)))
ops._BN_SYNC(msg)
m = int(round(float(msg[-1]))); assert m == 40
mean, var = msg[:6] / m, msg[6:12] / m - (msg[:6] / m) ** 2
assert torch.allclose(mean, full.mean(0), atol=1e-12) and torch.allclose(var, full.var(0, unbiased=False), atol=1e-10)
# an asynchronous bucket stays in flight while BatchNorm sums are exchanged (no serialisation, no retry: DESIGN 4), and the collective
# audit (what scripts/dist_stress.py runs with) re-derives every recorded collective on the host: clean run -> no finding; a result
# tampered with after the fact -> exactly that record is reported
from tests._gloo_staging import CollectiveAudit
audit = CollectiveAudit(); shdist.set_collective_audit(audit)
shdist.enable_sync_bn()                                                          # re-install: the audited callable
red3 = shdist.OverlappedGradReducer(bucket_bytes=16)
q3 = [torch.nn.Parameter(torch.zeros(n)) for n in (6, 5)]
red3.submit([(q3[0], torch.full((6,), float(rank + 1)))])
assert red3._pending and red3._pending[0][0] is not None                         # asynchronous work object
both = torch.tensor([1.0 + rank, 2.0]); ops._BN_SYNC(both)                      # a BatchNorm exchange between two buckets
assert torch.equal(both, torch.tensor([float(tot), 2.0 * world]))
red3.submit([(q3[1], torch.full((5,), 2.0 * (rank + 1)))]); o3 = red3.finish()
assert torch.equal(o3[q3[0]], torch.full((6,), float(tot))) and torch.equal(o3[q3[1]], torch.full((5,), 2.0 * tot))
assert [t for t, _, _ in audit.records] == ["bucket0", "bn_sync", "bucket1"]
assert audit.verify() == []
o3[q3[1]][3] += 64.0                                                             # bucket views alias the audited result
found = audit.verify()
assert len(found) == 1 and found[0]["tag"] == "bucket1" and found[0]["first_bad"] == found[0]["last_bad"] == 3 and found[0]["n_bad"] == 1, found
shdist.set_collective_audit(None)
shdist.disable_sync_bn()
assert not ops.bn_sync_active()
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_loss_and_grad_allreduce_gloo(tmp_path, world):
    """world_size 2 / 4 over gloo on CPU: sharded loss (packed [Z | J] all-gather, packed scalar exchanges; + its backward) ==
    single-process oracle on the global batch; gradient reducers (cached pattern, bucket views, bf16 wire)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29609 + world), WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o


TRANSPORT_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from simhand_amd.host import dist as shdist
from tests import _gloo_staging
rank, local, world = shdist.init_from_env("gloo")
tot = world * (world + 1) // 2
for mode in (None, "all", "buckets", "thread", "off"):
    shdist.set_transport(None if mode is None else _gloo_staging.GlooStagingTransport(mode))   # "thread": the twin group is created HERE, on every rank
    t = torch.full((5,), float(rank + 1)); shdist.all_reduce_(t)
    assert torch.equal(t, torch.full((5,), float(tot))), (mode, t)
    out = torch.empty(world * 3); shdist.all_gather_into(out, torch.full((3,), float(rank)))
    assert torch.equal(out, torch.arange(world, dtype=torch.float32).repeat_interleave(3)), (mode, out)
    m = torch.nn.Linear(3, 2); shdist.broadcast_module_state(m)
    ref = [p.detach().clone() for p in m.parameters()]
    for q in ref: dist.broadcast(q, src=0)
    assert all(torch.equal(a, b) for a, b in zip(m.parameters(), ref)), mode
    red = shdist.OverlappedGradReducer(bucket_bytes=16)
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (6, 5, 3)]
    red.submit([(p, torch.full((p.numel(),), float(rank + 1) * (i + 1))) for i, p in enumerate(ps)])
    got = red.finish()
    assert all(torch.equal(got[p], torch.full((p.numel(),), float(tot) * (i + 1))) for i, p in enumerate(ps)), mode
    assert red.side_buckets == 0
shdist.set_transport(None)
try:
    _gloo_staging.GlooStagingTransport("sometimes")
    raise SystemExit("expected a ValueError")
except ValueError:
    pass
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_transports_of_the_data_parallel_module_gloo_world2(tmp_path):
    """host/dist.py hands tensors to torch.distributed through a replaceable transport (round 6): the product transport and every mode of the
    shared-GPU test transport (tests/_gloo_staging.py; host tensors pass through untouched, "thread" creates its twin group eagerly on all
    ranks) carry all_reduce_ / all_gather_into / broadcast_module_state / the overlapped bucket reducer to the same results."""
    script = tmp_path / "tworker.py"
    script.write_text(TRANSPORT_WORKER)
    world = 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29627", WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o


def test_product_modules_read_only_the_documented_environment_variables():
    """the package reads the rendezvous variables and SAVED_MODELS_BASE_PATH (the reference's own), nothing else."""
    import re

    allowed = {"WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY", "SAVED_MODELS_BASE_PATH", "PL_GLOBAL_SEED"}
    found = set()
    for dirpath, _, files in os.walk(os.path.join(ROOT, "simhand_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                found |= set(re.findall(r"os\.environ(?:\.get|\.setdefault)?\s*[\[(]\s*[\"']([A-Za-z0-9_]+)[\"']", src))
                found |= set(re.findall(r"getenv\(\s*[\"']([A-Za-z0-9_]+)[\"']", src))
                if "HSA_IPC_ENV" in src:
                    found.add("HSA_ENABLE_IPC_MODE_LEGACY")
    assert found <= allowed, sorted(found - allowed)
    for f in os.listdir(os.path.join(ROOT, "simhand_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(ROOT, "simhand_amd", "csrc", f)).read(), f


def _torchvision_resnet_keys(layers, bottleneck):
    """The state-dict key list of torchvision.models.resnet{18,34,50,101,152} (v0.13), written out from its published
    module structure: conv1, bn1, layer1..4 (blocks with conv/bn pairs and an optional downsample = Sequential(conv, bn)), fc."""
    bn = lambda p: [f"{p}.weight", f"{p}.bias", f"{p}.running_mean", f"{p}.running_var", f"{p}.num_batches_tracked"]  # noqa: E731
    keys = ["conv1.weight"] + bn("bn1")
    inplanes, exp = 64, 4 if bottleneck else 1
    for li, (n, planes, stride) in enumerate(zip(layers, (64, 128, 256, 512), (1, 2, 2, 2)), start=1):
        for b in range(n):
            p = f"layer{li}.{b}"
            for c in range(1, (3 if bottleneck else 2) + 1):
                keys += [f"{p}.conv{c}.weight"] + bn(f"{p}.bn{c}")
            if b == 0 and (stride != 1 or inplanes != planes * exp):
                keys += [f"{p}.downsample.0.weight"] + bn(f"{p}.downsample.1")
            inplanes = planes * exp
    return keys + ["fc.weight", "fc.bias"]


def test_checkpoint_lookup_and_torchvision_export(tmp_path, monkeypatch):
    """src/models/utils.py:504-540 (get_latest_checkpoint / get_encoder_state_dict) and src/models/port_model.py:7-48
    (positional copy onto torchvision's key order), on a Lightning-style checkpoint of the step model's state dict."""
    import torch

    from oracle import step as orc
    from simhand_amd.host import export

    for size, layers, bott in (("50", [3, 4, 6, 3], True), ("18", [2, 2, 2, 2], False), ("152", [3, 8, 36, 3], True)):
        assert list(export.torchvision_resnet(size).state_dict().keys()) == _torchvision_resnet_keys(layers, bott), size
    assert len(_torchvision_resnet_keys([3, 4, 6, 3], True)) == 320  # torchvision resnet50: 320 entries
    torch.manual_seed(0)
    om = orc.StepOracle("simhand_w", "50", [], weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
    monkeypatch.setenv("SAVED_MODELS_BASE_PATH", str(tmp_path))
    ckdir = tmp_path / "exp1" / "checkpoints"
    ckdir.mkdir(parents=True)
    for ep in (0, 3, 12):
        sd = {k: v.clone() + ep for k, v in om.state_dict().items()}
        torch.save({"state_dict": sd, "epoch": ep}, ckdir / f"epoch={ep}.ckpt")
    assert export.get_latest_checkpoint("exp1") == str(ckdir / "epoch=12.ckpt")  # numeric, not lexicographic, order
    assert export.get_latest_checkpoint("exp1", "epoch=3.ckpt") == str(ckdir / "epoch=3.ckpt")
    enc = export.get_encoder_state_dict("exp1", "epoch=3.ckpt")
    want = {k[8:]: v + 3 for k, v in om.state_dict().items() if "encoder" in k}
    assert list(enc.keys()) == list(want.keys()) and all(k.startswith(("features.", "final_layer.")) for k in enc)
    assert all(torch.equal(enc[k], want[k]) for k in enc)
    out = tmp_path / "resnet50_simhand.pth"
    tv = export.export_torchvision_state_dict(str(ckdir / "epoch=0.ckpt"), str(out), 50)
    assert list(tv.keys()) == _torchvision_resnet_keys([3, 4, 6, 3], True)
    src = [(k, v) for k, v in om.state_dict().items() if "features" in k]
    for (dk, dv), (sk, sv) in zip(tv.items(), src):  # 318 encoder tensors by position; fc keeps its own init
        assert torch.equal(dv, sv), (dk, sk)
    assert torch.equal(torch.load(out)["layer1.0.downsample.0.weight"], om.state_dict()["encoder.features.4.0.downsample.0.weight"])
    with pytest.raises(ValueError):  # a ResNet-50 checkpoint does not fit a ResNet-18
        export.peclr_to_torchvision(export.torchvision_resnet(18), str(ckdir / "epoch=0.ckpt"))
    hub = export.resnet50_simhand(pretrained=True, path=str(out))
    assert torch.equal(hub.state_dict()["conv1.weight"], om.state_dict()["encoder.features.0.weight"])
    with pytest.raises(ValueError):
        export.resnet50_simhand(pretrained=True)
