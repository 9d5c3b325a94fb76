"""Worker for tests/test_gpu_dist.py::test_sync_batchnorm_*.  R ranks each run their shard of a HandCLR_W step through the HIP library
with SYNCHRONISED BatchNorm (host.dist.enable_sync_bn: every BatchNorm's sums all-reduced in its finalize step) + the gradient
all-reduce.  With statistics over the global batch the sharded step IS the single-process step on the concatenated batch, so rank 0
compares with the oracle's plain full-batch step (loss, per-tensor gradients, BatchNorm running statistics).
argv: repo root, backend, ResNet size ("18": BasicBlock, unfused paths; "50": Bottleneck -- the Gram-matrix folds switch themselves off)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT, BACKEND, SIZE = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, ROOT)
from oracle import step as orc  # noqa: E402
from simhand_amd import ops  # noqa: E402
from simhand_amd.host import dist as shdist  # noqa: E402
from tests import _gloo_staging  # noqa: E402
from tests.test_gpu_step import _product  # noqa: E402

# "gloo": the ranks SHARE one GPU (device tensors staged through host memory by the test transport); "nccl": the product path
rank, local, world = _gloo_staging.init_shared_gpu() if BACKEND == "gloo" else shdist.init_from_env()
if os.environ.get("SIMHAND_POISON_WORKER"):  # torch.empty returns NaN patterns (tests/_poison.py)
    from tests._poison import poison

    poison(float(os.environ["SIMHAND_POISON_WORKER"]))
if os.environ.get("SIMHAND_POISON_EVERY"):  # every torch.empty of the step returns NaN patterns, re-used blocks included
    from tests._poison import poison_every

    poison_every()
if os.environ.get("SIMHAND_CANARY"):  # every torch.empty of the step carries a guard tail (tests/_poison.py guard_every): wild writes show
    from tests._poison import guard_every

    guard_every()
audit = None
if os.environ.get("SIMHAND_DIST_DIAG"):  # scripts/dist_stress.py: every collective's input kept and re-derived on the host afterwards
    audit = _gloo_staging.CollectiveAudit()
    shdist.set_collective_audit(audit)
dev = torch.device("cuda", torch.cuda.current_device())
AUG = ["color_jitter", "crop", "random_crop", "resize", "rotate"]
wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
B, size, seed = 16, 64, 7
batch = orc.synthetic_batch(B, size=size, seed=seed)
torch.manual_seed(seed)
om = orc.StepOracle("simhand_w", SIZE, AUG, **wcfg).train()
if SIZE == "50":  # as tests/test_gpu_configs.py: gamma_3 = 0.1 keeps the random-init residual net out of its chaotic regime, where
    with torch.no_grad():  # ReLU-kink flips alone put ~2 % on every gradient (measured: the same 2 % at world 1 without any sync)
        for k, p in om.named_parameters():
            if k.endswith("bn3.weight"):
                p.fill_(0.1)
torch.manual_seed(1000 + rank)
model = _product("HandCLR_W", SIZE, wcfg, om)
shdist.broadcast_module_state(model)
assert shdist.enable_sync_bn() == (world > 1) and ops.bn_sync_active() == (world > 1)
if world == 1 and os.environ.get("SIMHAND_FORCE_SYNC_PATH"):  # one rank: the synchronised code path (unfused BatchNorm passes) with an
    ops.set_bn_sync(lambda t: None)                          # identity "all-reduce" -- same kernels, same buffers as with R ranks
off, b = shdist.shard_pairs(B, rank, world)
shard = {k: v[off:off + b].to(dev) for k, v in batch.items()}
reducer = shdist.OverlappedGradReducer(bucket_bytes=int(os.environ.get("SIMHAND_TEST_BUCKET", 1 << 20)))
if not os.environ.get("SIMHAND_TEST_NO_REDUCER"):
    model.encoder.engine.grad_reducer = reducer
loss = model.training_step(shard, 0)["loss"]
loss.backward()
shdist.allreduce_gradients(model.parameters(), bucket_bytes=1 << 20, skip=reducer.reduced)
torch.cuda.synchronize()
if os.environ.get("SIMHAND_CANARY"):
    from tests._poison import check_guards

    bad_guards, live_guards = check_guards()
    print(f"CANARY rank {rank}: {live_guards} live guarded allocations, {len(bad_guards)} overwritten {bad_guards[:8]}", flush=True)
    assert not bad_guards, bad_guards
if audit is not None and world > 1:
    import json

    shdist.set_collective_audit(None)
    findings = audit.verify()
    for f in findings:
        print(f"AUDIT rank {rank}: " + json.dumps(f), flush=True)
    print(f"AUDIT rank {rank}: {len(audit.records)} collectives re-derived on the host, {len(findings)} wrong", flush=True)
    AUDIT_BAD = len(findings)
else:
    AUDIT_BAD = 0
if os.environ.get("SIMHAND_DIST_RERUN") and audit is not None and world > 1:
    # scripts/dist_stress.py --rerun: the SAME step once more on every rank (the kernels are deterministic), and each rank compares what it
    # fed into every collective -- its LOCAL BatchNorm sums and its local gradient buckets -- between the two runs.  A rank whose first run
    # went wrong locally (the round-3..5 irregularity: all collectives exact, reduced gradients wrong) shows WHICH tensors differed first.
    first, first_owners = audit.records, audit.owners
    grads1 = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    bufs1 = {k: b.detach().clone() for k, b in model.named_buffers()}  # the second step must not move the running statistics the checks below read
    audit2 = _gloo_staging.CollectiveAudit()
    shdist.set_collective_audit(audit2)
    model.zero_grad()
    reducer2 = shdist.OverlappedGradReducer(bucket_bytes=int(os.environ.get("SIMHAND_TEST_BUCKET", 1 << 20)))
    if not os.environ.get("SIMHAND_TEST_NO_REDUCER"):
        model.encoder.engine.grad_reducer = reducer2
    loss2 = model.training_step(shard, 0)["loss"]
    loss2.backward()
    shdist.allreduce_gradients(model.parameters(), bucket_bytes=1 << 20, skip=reducer2.reduced)
    torch.cuda.synchronize()
    shdist.set_collective_audit(None)
    with torch.no_grad():
        for k, b in model.named_buffers():
            b.copy_(bufs1[k])
        for k, p in model.named_parameters():  # and the checks read the FIRST run's reduced gradients
            if k in grads1:
                g2 = p.grad.detach().clone()
                p.grad.copy_(grads1[k])
                grads1[k] = g2
    names = {id(p): k for k, p in model.named_parameters()}
    ndiff, first_lines = 0, []
    if len(first) != len(audit2.records):
        print(f"LOCALDIFF rank {rank}: {len(first)} collectives in run 1, {len(audit2.records)} in run 2", flush=True)
    for i, ((tag, pre1, post1), (tag2, pre2, post2)) in enumerate(zip(first, audit2.records)):
        for what, a, b in (("local input", pre1, pre2), ("result", post1, post2)):
            a, b = a.detach().float().cpu().reshape(-1), b.detach().float().cpu().reshape(-1)
            if a.shape == b.shape and torch.equal(a, b):
                continue
            ndiff += 1
            if a.shape != b.shape:
                print(f"LOCALDIFF rank {rank}: record {i} {tag} / {tag2} {what}: shapes {tuple(a.shape)} vs {tuple(b.shape)}", flush=True)
                continue
            d = (a - b).abs()
            idx = d.nonzero().reshape(-1)
            where = ""
            if first_owners[i]:
                off_, parts = 0, []
                for pid, n_ in first_owners[i]:
                    seg = d[off_:off_ + n_]
                    if bool((seg != 0).any()):
                        parts.append(f"{names.get(pid, '?')} {float(seg.max()) / (float(b[off_:off_ + n_].abs().max()) + 1e-30):.2e}")
                    off_ += n_
                where = " | " + "; ".join(parts[:12])
            line = (f"LOCALDIFF rank {rank}: record {i} of {len(first)} {tag} {what}: {idx.numel()} of {a.numel()} elements differ between run 1 and run 2, first {int(idx[0])} last "
                    f"{int(idx[-1])}, max abs {float(d.max()):.3e} (run-2 abs max {float(b.abs().max()):.3e}){where}")
            if len(first_lines) < 6:
                first_lines.append(line)
            if ndiff <= 12:
                print(line, flush=True)
    for line in first_lines:  # once more at the END of the output (scripts/dist_stress.py keeps the tail): where the two runs part
        print(line.replace("LOCALDIFF", "LOCALFIRST"), flush=True)
    gd = [(k, float((p.grad - grads1[k]).abs().max()), float(grads1[k].abs().max())) for k, p in model.named_parameters() if k in grads1 and not torch.equal(p.grad, grads1[k])]  # run 1 (p.grad) vs run 2
    print(f"RERUN rank {rank}: {ndiff} collective inputs / results differ between the two runs; loss {loss.item():.7f} vs {loss2.item():.7f}; {len(gd)} reduced "
          f"gradients differ {[(k, f'{a:.2e}/{b:.2e}') for k, a, b in gd[:8]]}", flush=True)
# every rank normalised with the same statistics: the running buffers agree bit for bit across ranks
for k, buf in model.named_buffers():
    if buf.dtype.is_floating_point:
        lo, hi = buf.detach().clone(), buf.detach().clone()
        if world > 1:
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), k
if rank == 0:
    lo = om.contrastive_step(batch)  # ONE process, the whole batch: BatchNorm over all 2 B images
    lo.backward()
    assert abs(loss.item() - lo.item()) <= 2e-4 * abs(lo.item()), (loss.item(), lo.item())
    og = dict(om.named_parameters())
    errs, named = [], {}
    for k, p in model.named_parameters():
        if og[k].grad is None or og[k].grad.abs().max() < 1e-6:
            continue
        named[k] = ((p.grad.cpu() - og[k].grad).norm() / og[k].grad.norm()).item()
        errs.append(named[k])
    errs.sort()
    worst = sorted(named.items(), key=lambda kv: -kv[1])[:6]
    bad = [(k, round(v, 4)) for k, v in named.items() if not v <= 6e-2]
    nonfinite = [k for k, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    assert errs[len(errs) // 2] <= 1e-2 and errs[-1] <= 6e-2, (errs[len(errs) // 2], errs[-1], worst, len(bad), len(named), bad[:40], nonfinite[:10])
    ob = dict(om.named_buffers())
    worst = 0.0
    for k, buf in model.named_buffers():
        if buf.dtype.is_floating_point:
            ref = ob[k]
            worst = max(worst, ((buf.cpu() - ref).abs().max() / (ref.abs().max() + 1e-6)).item())
    assert worst <= 2e-3, worst
    # and per-rank statistics do NOT reproduce the full-batch step (the test can tell the two apart)
    print(f"rank 0 [sync BN, {BACKEND} x{world}, ResNet-{SIZE}]: loss {loss.item():.6f} == full-batch oracle {lo.item():.6f}; grad rel-L2 median "
          f"{errs[len(errs)//2]:.2e} max {errs[-1]:.2e}; running stats within {worst:.1e}")
if BACKEND == "nccl" and world > 1:
    # the same step with every exchange through the C ABI's RCCL wrappers.  Under synchronised BatchNorm the reducer keeps the gradient
    # buckets IN ORDER on the launch stream (first ncclComm, next to the BatchNorm sums): the second communicator / side stream is the
    # arrangement of the per-rank-statistics step (tests/_dist_worker.py asserts it is used there) -- reduced gradients tensor by tensor
    comm = shdist.RcclComm.from_torch_distributed()
    torch_grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    model.process_group = comm
    assert shdist.enable_sync_bn(comm)
    red_abi = shdist.OverlappedGradReducer(comm, bucket_bytes=1 << 20)
    assert comm.side_stream() is not None
    model.encoder.engine.grad_reducer = red_abi
    loss_abi = model.training_step(shard, 0)["loss"]
    loss_abi.backward()
    shdist.allreduce_gradients(model.parameters(), group=comm, bucket_bytes=1 << 20, skip=red_abi.reduced)
    torch.cuda.synchronize()
    assert red_abi.side_buckets == 0, red_abi.side_buckets
    assert abs(loss_abi.item() - loss.item()) <= 1e-6 * abs(loss.item()), (loss_abi.item(), loss.item())
    for k, p in model.named_parameters():
        if k in torch_grads:
            ref = torch_grads[k]
            assert (p.grad - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-8, k
    model.process_group = None
    model.encoder.engine.grad_reducer = None
    comm.close()
shdist.disable_sync_bn()
if world > 1:
    # control: the same shards with per-rank statistics give a different loss (the comparison above can tell the two modes apart)
    with torch.no_grad():
        loss_local = model.training_step(shard, 1)["loss"]
    assert abs(loss_local.item() - loss.item()) > 1e-3 * abs(loss.item()), (loss_local.item(), loss.item())
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
assert AUDIT_BAD == 0, f"{AUDIT_BAD} collectives returned a wrong result from right inputs"
print("rank", rank, "ok")
