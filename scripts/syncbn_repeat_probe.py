"""Fast reproduction harness for the shared-GPU synchronised-BatchNorm case (known issue): 4 ranks share the GPU over gloo
and REPEAT the sharded fp32 ResNet-50 step (synchronised BatchNorm + overlapped gradient buckets) in one process each; every rank
compares every reduced gradient of every repetition with its first repetition.  No oracle, no process start-up per sample:
hundreds of samples per minute instead of three.
usage: syncbn_repeat_probe.py [repetitions]       (parent: spawns the ranks + one competing GPU process)
env:   SIMHAND_GLOO_ASYNC_BUCKETS=1  the asynchronous buckets the harness showed the failure with (default: one collective at a time)
       PROBE_COLD=1                  empty the allocator caches before every repetition (first-touch allocations, as in the test)"""
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(reps):
    os.environ["SIMHAND_SHARE_GPU"] = "1"
    import torch.distributed as dist

    from oracle import step as orc  # test infrastructure: seeded weights and batch
    from simhand_amd import ops
    from simhand_amd.host import dist as shdist
    from tests.test_gpu_step import _product

    rank, local, world = shdist.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    aug = ["color_jitter", "crop", "random_crop", "resize", "rotate"]
    wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
    batch = orc.synthetic_batch(16, size=64, seed=7)
    torch.manual_seed(7)
    om = orc.StepOracle("simhand_w", "50", aug, **wcfg).train()
    with torch.no_grad():
        for k, p in om.named_parameters():
            if k.endswith("bn3.weight"):
                p.fill_(0.1)
    model = _product("HandCLR_W", "50", wcfg, om)
    shdist.broadcast_module_state(model)
    assert shdist.enable_sync_bn() and ops.bn_sync_active()
    off, b = shdist.shard_pairs(16, rank, world)
    shard = {k: v[off:off + b].to(dev) for k, v in batch.items()}
    cold = bool(os.environ.get("PROBE_COLD"))
    first, bad, diff = None, 0, []
    for it in range(reps):
        for p in model.parameters():
            p.grad = None
        if cold:
            torch.cuda.empty_cache()
        reducer = shdist.OverlappedGradReducer(bucket_bytes=1 << 20)
        model.encoder.engine.grad_reducer = reducer
        loss = model.training_step(shard, 0)["loss"]
        loss.backward()
        shdist.allreduce_gradients(model.parameters(), bucket_bytes=1 << 20, skip=reducer.reduced)
        torch.cuda.synchronize()
        cur = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        if first is None:
            first = cur
            continue
        diff = [(k, ((cur[k] - first[k]).norm() / (first[k].norm() + 1e-30)).item()) for k in cur if not torch.equal(cur[k], first[k])]
        big = [(k, round(e, 4)) for k, e in diff if e > 1e-3]
        if rank == 0 and it % 25 == 0:
            print(f"rep {it}: loss {loss.item():.6f}, {len(diff)} tensors not bit-equal to the first repetition", flush=True)
        if big:
            bad += 1
            print(f"rank {rank} rep {it}: loss {loss.item():.6f}; {len(big)} tensors differ by > 1e-3: {big[:5]}", flush=True)
    print(f"rank {rank}: {reps} repetitions, {bad} bad; last repetition: {len(diff)} tensors not bit-equal to the first", flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main(reps):
    world = 4
    load = subprocess.Popen([sys.executable, "-c",
                             "import torch,time\nx=torch.randn(8192,8192,device='cuda',dtype=torch.bfloat16)\nwhile True:\n"
                             "    for _ in range(50): y=x@x\n    torch.cuda.synchronize(); time.sleep(0.05)\n"])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(reps)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(world)]
    rc = [p.wait() for p in procs]
    load.kill()
    print(f"syncbn_repeat_probe: async buckets {bool(os.environ.get('SIMHAND_GLOO_ASYNC_BUCKETS'))}, cold {bool(os.environ.get('PROBE_COLD'))}: "
          f"return codes {rc}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker(int(sys.argv[2]))
    else:
        main(int(sys.argv[1]) if len(sys.argv) > 1 else 300)
