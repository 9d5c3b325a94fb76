"""Is it gloo?  R processes share the GPU (as tests/test_gpu_dist.py arranges them) and run, with NO simhand kernel involved, the
collective pattern of the synchronised-BatchNorm backward: an ASYNCHRONOUS 4-MB device-tensor all-reduce in flight while small
synchronous device-tensor all-reduces run, then wait and check both against the closed-form sums.
usage: gloo_concurrency_probe.py            (parent: spawns 4 ranks + one competing GPU process)
       gloo_concurrency_probe.py worker     (a rank; RANK / WORLD_SIZE / MASTER_* from the environment)
PROBE_ITERS (1500), PROBE_COLD=1: empty the device and pinned-host caches every iteration (first-touch allocations, as in a one-step test)."""
import os
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ITERS = int(os.environ.get("PROBE_ITERS", "1500"))


def worker():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    n_big, n_small = 1 << 20, 2 * 2048
    base_big = torch.arange(n_big, device=dev, dtype=torch.float32) % 1024
    base_small = torch.arange(n_small, device=dev, dtype=torch.float32) % 64
    tot = sum(r + 1 for r in range(world))
    bad = 0
    filler = torch.randn(2048, 2048, device=dev)
    cold = bool(os.environ.get("PROBE_COLD"))     # every iteration on freshly hipMalloc'd / freshly pinned memory (a one-step process)
    for it in range(ITERS):
        if cold:
            torch.cuda.empty_cache()
            try:
                torch._C._host_emptyCache()
            except AttributeError:
                pass
        big = (base_big + it) * (rank + 1)          # produced by a kernel right before the collective, as a gradient is
        work = dist.all_reduce(big, async_op=True)
        smalls = []
        for k in range(3):                          # three BatchNorm-sized synchronous all-reduces while the bucket is in flight
            filler = filler @ filler * 1e-3         # compute between them
            s = (base_small + it + k) * (rank + 1)
            dist.all_reduce(s)
            smalls.append((k, s))
        work.wait()
        ok = torch.equal(big, (base_big + it) * tot)
        for k, s in smalls:
            ok = ok and torch.equal(s, (base_small + it + k) * tot)
        if not ok:
            bad += 1
            print(f"rank {rank} iteration {it}: WRONG (big max err {(big - (base_big + it) * tot).abs().max().item():.3g})", flush=True)
    torch.cuda.synchronize()
    print(f"rank {rank}: {ITERS} iterations, {bad} wrong", flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    world = 4
    load = subprocess.Popen([sys.executable, "-c",
                             "import torch,time\nx=torch.randn(8192,8192,device='cuda',dtype=torch.bfloat16)\nwhile True:\n"
                             "    for _ in range(50): y=x@x\n    torch.cuda.synchronize(); time.sleep(0.05)\n"])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29733", WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker"], env=dict(env, RANK=str(r))) for r in range(world)]
    rc = [p.wait() for p in procs]
    load.kill()
    print("gloo_concurrency_probe: return codes", rc, f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    worker() if len(sys.argv) > 1 and sys.argv[1] == "worker" else main()
