"""GPU: the N > 1 path on real kernels.  World size 2 and 4 on ONE GPU (gloo transports the collectives because
RCCL refuses two ranks per device): sharded HIP step + gradient all-reduce == oracle with per-shard BatchNorm
statistics and the global NT-Xent."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_step_matches_oracle(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29620 + world), WORLD_SIZE=str(world), OMP_NUM_THREADS="4",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "_dist_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o
