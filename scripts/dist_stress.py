"""Stress of the multi-rank GPU test arrangement (ON THE GPU BOX): G concurrent groups of 4 ranks each repeat
tests/_syncbn_worker.py (ResNet-50, fp32, synchronised BatchNorm + overlapped gradient buckets, ranks sharing the one GPU over
gloo) as FRESH processes for a fixed number of minutes; the groups are each other's competing GPU load.  Every repetition is the
whole test (loss, per-tensor gradients and running statistics against the oracle's full-batch step); with --diag the worker also
keeps the input of every collective and re-derives each result on the host afterwards (host.dist.CollectiveAudit), which tells a
collective that returned a wrong result from right inputs apart from wrong inputs.

usage: python scripts/dist_stress.py --tag off --staging off --minutes 8 --groups 3 --diag
writes gpurun_out/dist_stress_<tag>.log (one line per repetition + the output tail of every failure + a summary line)."""
import argparse
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--tag", required=True)
ap.add_argument("--staging", default="all", choices=["all", "buckets", "thread", "off"])
ap.add_argument("--canary", action="store_true", help="guard tails behind every torch.empty of the worker, checked after the step (tests/_poison.py)")
ap.add_argument("--minutes", type=float, default=8.0)
ap.add_argument("--groups", type=int, default=3)
ap.add_argument("--world", type=int, default=4)
ap.add_argument("--size", default="50")
ap.add_argument("--max-reps", type=int, default=10 ** 6)
ap.add_argument("--diag", action="store_true")
ap.add_argument("--rerun", action="store_true", help="with --diag: every rank runs the step twice and compares what it fed into each collective (LOCALDIFF lines)")
args = ap.parse_args()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", f"dist_stress_{args.tag}.log"), "w")
lock = threading.Lock()
stats = {"reps": 0, "failed": 0, "audit_findings": 0}
deadline = time.time() + 60.0 * args.minutes
worker = os.path.join(ROOT, "tests", "_syncbn_worker.py")


def say(msg):
    with lock:
        log.write(msg + "\n")
        log.flush()


def group(gi):
    it = 0
    while time.time() < deadline:
        with lock:
            if stats["reps"] >= args.max_reps:
                return
            stats["reps"] += 1
            rep = stats["reps"]
        port = 30000 + gi * 1000 + (it % 900)
        it += 1
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(args.world), OMP_NUM_THREADS="4",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", SIMHAND_GLOO_STAGING=args.staging)
        if args.diag:
            env["SIMHAND_DIST_DIAG"] = "1"
        if args.canary:
            env["SIMHAND_CANARY"] = "1"
        if args.rerun:
            env["SIMHAND_DIST_RERUN"] = "1"
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, worker, ROOT, "gloo", args.size], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(args.world)]
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=600)[0])
            except subprocess.TimeoutExpired:
                p.kill()
                outs.append(p.communicate()[0] + "\n[dist_stress] TIMEOUT")
        bad = [r for r, p in enumerate(procs) if p.returncode != 0]
        audit = [ln for o in outs for ln in o.splitlines() if (ln.startswith("AUDIT") and '"tag"' in ln) or ln.startswith("LOCALFIRST")]
        with lock:
            stats["failed"] += 1 if bad else 0
            stats["audit_findings"] += len(audit)
        say(f"rep {rep} group {gi} {time.time() - t0:.1f}s {'FAILED ranks ' + str(bad) if bad else 'ok'} audit_findings {len(audit)}")
        if bad or audit:
            for r in (bad or range(args.world)):
                say(f"--- rep {rep} rank {r} output tail ---\n{outs[r][-6000:]}")


threads = [threading.Thread(target=group, args=(g,)) for g in range(args.groups)]
for t in threads:
    t.start()
for t in threads:
    t.join()
summary = (f"SUMMARY tag={args.tag} staging={args.staging} diag={int(args.diag)} canary={int(args.canary)} world={args.world} resnet={args.size} groups={args.groups} "
           f"minutes={args.minutes} reps={stats['reps']} failed={stats['failed']} audit_findings={stats['audit_findings']}")
say(summary)
print(summary)
