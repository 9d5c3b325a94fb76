#!/usr/bin/env python
"""Headline benchmark: hand-image-pairs/s of the SiMHand contrastive training step.

    python bench.py --gpus N --steps K --warmup W          # N > 1 without WORLD_SIZE: spawns its own N ranks (RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (BASELINE.json configs[1] / [2]): ResNet-50 `handclr_w`, bf16 MFMA kernels, per-GPU batch
1024 pairs of synthetic 224x224x3 images (2048 images per GPU), linear MPJPE weighting, crop+rotate
un-warp, global negatives over all ranks (weak scaling: global batch = 1024 * N).  One step = forward,
backward, gradient all-reduce (N > 1) and the LARS/Adam update -- nothing is skipped or cached.
Inputs are resident in HBM before the timed region.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel class (the implicit-GEMM
convolutions), measured live with HIP events on the launch stream inside the timed region;
`cpu_baseline` is the CPU oracle (a port of the reference step, validated against the reference's
golden vectors) timed on this host on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
FP8_DENSE_PEAK_TFLOPS = 5000.0   # ~5 PF dense fp8 (block-scaled K = 128 MFMA)
F32_PEAK_TFLOPS = 157.3
AUG = ["color_jitter", "crop", "random_crop", "resize", "rotate"]
TRAIN_GFLOP_PER_PAIR = {"18": 21.77, "50": 49.06, "152": 138.15}  # BASELINE.md section 3 (@224)


def source_hash() -> str:
    """sha1 over the sources that decide what the step launches (kernels, dispatch, engine): profiles/hbm_traffic.json records the
    hash it was measured at, so a static PMC figure can never be quoted against a different build without saying so."""
    import glob
    import hashlib

    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "simhand_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "simhand_amd", "csrc", "*.h"))
                   + [os.path.join(ROOT, "simhand_amd", "ops.py"), os.path.join(ROOT, "simhand_amd", "host", "resnet_model.py")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--per-gpu-batch", type=int, default=1024, help="pairs per GPU (BASELINE config: 1024)")
    ap.add_argument("--resnet", default="50", choices=["18", "34", "50", "101", "152"])
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "32", "fp8", "16"],
                    help="bf16 = BASELINE's dtype (default); 16 = the reference's own policy: fp16 storage + dynamic loss scaling (fp16 build of the "
                         "library); fp8 = BASELINE configs[4]: bf16 storage, e4m3 operands where they pay (parity n/a); 32 = exact-fp32 parity mode")
    ap.add_argument("--experiment", default="handclr_w", choices=["handclr_w", "peclr_w", "simclr"])
    ap.add_argument("--comm", default="abi", choices=["abi", "torch"],
                    help="N > 1: the step's collectives through the C ABI's own RCCL wrappers (simhand_comm_*: the communicator the product documents -- "
                         "loss exchanges on the compute stream's ncclComm, gradient buckets overlapped on a second ncclComm's side stream; default) or "
                         "through torch.distributed (nccl = RCCL; the fallback).  Rendezvous and the timing barrier use torch.distributed either way")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST ARRANGEMENT ONLY (tests/_gloo_staging.py): the N ranks share the visible GPU(s) over gloo with host staging -- what a "
                         "1-GPU box can run of the N > 1 code path; never a measurement")
    ap.add_argument("--grad-wire", default="fp32", choices=["fp32", "bf16"], help="N > 1: wire format of the gradient all-reduce buckets")
    ap.add_argument("--event-every", type=int, default=1,
                    help="HIP events bracket every launch of the roofline's kernel class in every Nth timed step (1 = every step, the default; "
                         "measured: sampling every 4th step moves the step time by < 0.1 ms)")
    ap.add_argument("--sync-bn", action="store_true", help="N > 1: synchronised BatchNorm (statistics over the global batch; not the headline configuration)")
    ap.add_argument("--switch", action="append", default=[], metavar="NAME=V",
                    help="A/B timing only: a kernel-selection test hook of the library (simhand_test_switch; names in simhand_amd.ops.TEST_SWITCHES), "
                         "e.g. --switch R128=0.  The library reads no environment variable; the default run sets none of these")
    ap.add_argument("--engine", action="append", default=[], metavar="ATTR=V",
                    help="A/B timing only: an engine-level fusion attribute of host/resnet_model.py ResNetEngine (chain_conv1, dense_shortcut, "
                         "merge_shortcut, fuse_apply_gram, fuse_bwd_apply_dgrad, fuse_bwd_apply_wgrad, fp8_all, bn_on_load), e.g. --engine chain_conv1=0")
    ap.add_argument("--emulate-world", type=int, default=1, metavar="R",
                    help="ONE process runs the rank-local work of an R-rank step (BASELINE configs[2] at R = 8): its own pairs through the backbone, "
                         "the loss row block 2 b_loc x 2 b_loc R against a synthetic gathered Z_all / J_all, gradient buckets flattened and cast as on "
                         "the wire -- NO collective is executed (an upper bound on data-parallel efficiency where no multi-GPU node is available)")
    ap.add_argument("--lib", default=None, metavar="PATH", help="A/B timing only: another build of libsimhand_hip.so (scripts/build_variant.sh)")
    ap.add_argument("--lib-f16", default=None, metavar="PATH", help="A/B timing only: another build of libsimhand_hip_f16.so")
    ap.add_argument("--no-loss-scaling", action="store_true", help="--precision 16 without the GradScaler (timing split only)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline AND the precision-16 companion line (profiling / A-B runs)")
    ap.add_argument("--no-parity-probe", action="store_true",
                    help="skip the smoke-size oracle comparison in front of the run (rocprofv3 passes: its ResNet-18 launches would be counted into the tables)")
    ap.add_argument("--cpu-pairs", type=int, default=32,
                    help="pairs in the CPU-baseline sample (default = the SURVEY 8d point B = 32; ~40 s on the box's host for ResNet-50)")
    ap.add_argument("--cpu-full", action="store_true",
                    help="also time the third SURVEY 8d CPU point (ResNet-50 B=128): minutes of CPU time")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks here, one process per GPU, before this process
    has touched the GPU (children are fresh interpreters, nothing is exec'ed over an initialised runtime).  Rank 0
    inherits stdout and prints the JSON line."""
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE=str(args.gpus), HSA_ENABLE_IPC_MODE_LEGACY="0")
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    procs = []
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # poll all ranks: the first failure ends the siblings (they would otherwise sit in a collective until the backend's watchdog
    # fires); an overall limit bounds the whole run
    deadline = time.monotonic() + float(os.environ.get("SIMHAND_BENCH_TIMEOUT_S", "3600"))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code)
        if rc != 0 or time.monotonic() > deadline:
            for p in live:  # exact PIDs this process started, nothing else
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            return rc or 124
        time.sleep(0.05)
    return rc


class DeviceStateSampler:
    """Clock / power of the GPU WHILE the timed steps run, read from the amdgpu sysfs nodes by a host thread (no device work, outside
    every kernel): hwmon freq1_input (gfx clock), power1_average / power1_input (socket power), gpu_busy_percent.  The matrix-bound kernels
    of this step run at the clock the power limit leaves (1.7-2.0 GHz on random operands, 2.25 GHz on zeros: profiles/
    r05_igemm256_clock_and_ring.txt), so a slow box and a regression are told apart by these numbers on the bench line."""

    def __init__(self, device_index: int, period_s: float = 0.02):
        import glob
        import threading

        self.period, self.samples, self._stop, self._thr = period_s, [], threading.Event(), None
        cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))
        want = None
        try:  # match the HIP device's PCI address when torch exposes it, else the index-th amdgpu card
            pr = torch.cuda.get_device_properties(device_index)
            want = f"{getattr(pr, 'pci_domain_id'):04x}:{getattr(pr, 'pci_bus_id'):02x}:{getattr(pr, 'pci_device_id'):02x}"
        except Exception:  # noqa: BLE001
            want = None
        pick = next((c for c in cards if want and os.path.basename(os.path.realpath(c)).startswith(want)), None)
        if pick is None and cards:
            pick = cards[min(device_index, len(cards) - 1)]
        self.dir = pick
        hw = sorted(glob.glob(os.path.join(pick, "hwmon", "hwmon*"))) if pick else []
        self.hw = hw[0] if hw else None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except Exception:  # noqa: BLE001
            return None

    def _sample(self):
        clk = pw = busy = None
        if self.hw:
            v = self._read(os.path.join(self.hw, "freq1_input"))
            clk = float(v) / 1e6 if v else None
            v = self._read(os.path.join(self.hw, "power1_average")) or self._read(os.path.join(self.hw, "power1_input"))
            pw = float(v) / 1e6 if v else None
        if clk is None and self.dir:
            v = self._read(os.path.join(self.dir, "pp_dpm_sclk")) or ""
            cur = [ln for ln in v.splitlines() if ln.rstrip().endswith("*")]
            if cur:
                try:
                    clk = float(cur[0].split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                except Exception:  # noqa: BLE001
                    clk = None
        if self.dir:
            v = self._read(os.path.join(self.dir, "gpu_busy_percent"))
            busy = float(v) if v else None
        return clk, pw, busy

    def start(self):
        import threading

        if self.dir is None:
            return self

        def run():
            while not self._stop.is_set():
                self.samples.append(self._sample())
                self._stop.wait(self.period)

        self._thr = threading.Thread(target=run, daemon=True)
        self._thr.start()
        return self

    def stop(self) -> dict:
        self._stop.set()
        if self._thr is not None:
            self._thr.join(timeout=1.0)
        if self.dir is None:
            return {"available": False, "why": "no amdgpu sysfs node with pp_dpm_sclk readable by this user"}

        def agg(i):
            v = [s[i] for s in self.samples if s[i] is not None]
            return None if not v else {"mean": sum(v) / len(v), "min": min(v), "max": max(v)}

        return {"available": True, "samples": len(self.samples), "period_ms": self.period * 1e3, "sclk_mhz": agg(0), "socket_power_w": agg(1),
                "gpu_busy_percent": agg(2), "source": f"{self.dir} (sysfs, host thread, during the timed steps)"}


def make_emulated_comm(world: int):
    """--emulate-world: a communicator whose collectives move NO data between devices.  all_gather_into fills every rank's slot with this
    rank's own rows (a device-local copy of the size RCCL would deliver: the loss then runs its full 2 b_loc x N row block, N = world x
    the local rows); all_reduce_ is a no-op (the buckets were already flattened / cast to the wire format by the caller)."""
    from simhand_amd.host import dist as shdist

    class EmulatedComm(shdist.RcclComm):
        def __init__(self, world):  # no ncclComm is created
            self.world, self.rank, self._side = world, 0, None

        def side_stream(self):
            if self._side is None:
                self._side = torch.cuda.Stream(priority=-1)
            return self._side

        def all_gather_into(self, out, x):
            out.view(self.world, -1).copy_(x.reshape(1, -1).expand(self.world, -1))

        def all_reduce_(self, t, op="sum", side=False):
            return t

        def close(self):
            pass

    return EmulatedComm(world)


def make_model(args, world):
    from simhand_amd.host import unsupervised
    from simhand_amd.host.config import edict

    wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg") if args.experiment != "simclr" else {}
    cfg = edict(resnet_size=args.resnet, projection_head_input_dim=2048, projection_head_hidden_dim=512, output_dim=128,
                augmentation=list(AUG), joints_type="augmented", use_pca=False, lr=1e-4, opt_weight_decay=1e-6,
                warmup_epochs=10, num_of_mini_batch=1, optimizer="LARS", batch_size=args.per_gpu_batch * world,
                num_samples=1_000_000, **wcfg)
    cls = {"handclr_w": unsupervised.HandCLR_W, "peclr_w": unsupervised.PeCLR_W, "simclr": unsupervised.SimCLR}[args.experiment]
    torch.manual_seed(5)
    model = cls(cfg, None, "train")
    model.set_compute_dtype({"32": torch.float32, "16": torch.float16}.get(args.precision, torch.bfloat16), fp8=args.precision == "fp8")
    return model


def device_batch(b, size, seed, device):
    """SURVEY 8d synthetic batch, generated directly on the device (same distributions as
    oracle.step.synthetic_batch; values differ because the generator is the device's)."""
    g = torch.Generator(device=device).manual_seed(seed)
    j1 = torch.rand(b, 21, 3, generator=g, device=device) * size
    j1[:, :, 2] = 1.0
    j2 = j1.clone()
    j2[:, :, :2] += torch.randn(b, 21, 2, generator=g, device=device) * 8.0
    ri = lambda lo, hi: torch.randint(lo, hi, (b,), generator=g, device=device)  # noqa: E731
    return {
        "transformed_image1": torch.randn(b, 3, size, size, generator=g, device=device),
        "transformed_image2": torch.randn(b, 3, size, size, generator=g, device=device),
        "joints1_aug": j1, "joints2_aug": j2, "joints1_ori": j1 / size, "joints2_ori": j2 / size,
        "angle_1": ri(-45, 46).to(torch.float64), "angle_2": ri(-45, 46).to(torch.float64),
        "jitter_x_1": -ri(0, 15), "jitter_x_2": -ri(0, 15), "jitter_y_1": -ri(0, 15), "jitter_y_2": -ri(0, 15),
    }


def host_cpu():
    """CPU model, sockets x cores (physical) and hardware threads of this host, from lscpu."""
    info = {"model": "unknown", "physical_cores": None, "threads": os.cpu_count()}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {ln.split(":", 1)[0].strip(): ln.split(":", 1)[1].strip() for ln in out.splitlines() if ":" in ln}
        info["model"] = kv.get("Model name", "unknown")
        info["physical_cores"] = int(kv.get("Socket(s)", "1")) * int(kv.get("Core(s) per socket", "0")) or None
    except Exception:  # noqa: BLE001 -- lscpu missing: keep the defaults
        pass
    return info


def _cpu_point(exp, resnet, pairs, size, wcfg, budget_s, warm=2, max_timed=5, min_timed=3):
    """One CPU reference point: the oracle step (fwd + bwd, then LARS/Adam timed separately) -- median of `min_timed` .. `max_timed`
    timed steps after `warm` warm-ups; beyond `min_timed` the timed steps stop once `budget_s` of wall time is spent (a median of
    fewer than three samples is not a median: VERDICT r3 weak #8)."""
    from oracle import step as orc
    from oracle.optim import LARSWrapperOracle

    torch.manual_seed(5)
    model = orc.StepOracle(exp, resnet, AUG, **wcfg).train()
    adam = torch.optim.Adam(model.parameters(), lr=3.2e-3, weight_decay=1e-6)
    opt = LARSWrapperOracle(adam)
    batch = orc.synthetic_batch(pairs, size=size, seed=5)
    fb, op = [], []
    t_start = time.perf_counter()
    for i in range(warm + max_timed):
        t0 = time.perf_counter()
        adam.zero_grad(set_to_none=True)
        loss = model.contrastive_step(batch)
        loss.backward()
        t1 = time.perf_counter()
        opt.step()
        t2 = time.perf_counter()
        if i >= warm:
            fb.append(t1 - t0)
            op.append(t2 - t1)
        if len(fb) >= min_timed and time.perf_counter() - t_start > budget_s:
            break
    fb.sort()
    op.sort()
    tf, to = fb[len(fb) // 2], op[len(op) // 2]
    return {"pairs_per_s": pairs / tf, "pairs_per_s_incl_optimizer": pairs / (tf + to), "fwd_bwd_ms": tf * 1e3, "optimizer_ms": to * 1e3,
            "pairs": pairs, "timed_steps": len(fb), "warmups": warm, "resnet": resnet, "image_size": size}


def cpu_baseline(args):
    """The oracle (CPU port of the reference step, torch fp32, validated against the reference's golden vectors) on the
    host cores.  Headline point: the benchmarked network at 32 pairs.  --cpu-full adds BASELINE configs[0]
    (ResNet-18, 32 pairs) and the 128-pair point."""
    wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg") if args.experiment != "simclr" else {}
    exp = {"handclr_w": "simhand_w", "peclr_w": "peclr_w", "simclr": "simclr"}[args.experiment]
    cpu = host_cpu()
    threads = torch.get_num_threads()
    # bounded sample: 1 warm-up + 3 timed steps of the benchmarked network at the SURVEY 8d batch (B = 32; ~15 s per ResNet-50 step on
    # the GPU box's 128 cores), and BASELINE configs[0] (ResNet-18 handclr_w, B = 32 -- the reference's own CPU-runnable case) next to
    # it: ~75 s of host time in the default run
    pt = _cpu_point(exp, args.resnet, args.cpu_pairs, args.image_size, wcfg, budget_s=45.0, warm=1, max_timed=3, min_timed=3)
    res = {"value": pt["pairs_per_s"], "unit": "pairs/s", "cores": threads, "kind": "port",
           "cpu_model": cpu["model"], "physical_cores": cpu["physical_cores"], "hardware_threads": cpu["threads"],
           "optimizer_ms": pt["optimizer_ms"], "fwd_bwd_ms": pt["fwd_bwd_ms"],
           "sample": f"oracle.StepOracle ResNet-{args.resnet} {args.experiment} fp32, {pt['pairs']} pairs of {args.image_size}x{args.image_size}, "
                     f"fwd+bwd (optimizer timed separately: {pt['optimizer_ms']:.0f} ms), median of {pt['timed_steps']} steps after {pt['warmups']} warm-up "
                     f"({pt['fwd_bwd_ms']:.0f} ms/step), {threads} torch threads on {cpu['model']} ({cpu['physical_cores']} physical cores)"}
    res["points"] = {f"ResNet-{args.resnet} B={args.cpu_pairs}": pt}
    if args.experiment == "handclr_w":
        res["points"]["configs[0] ResNet-18 B=32"] = _cpu_point(exp, "18", 32, args.image_size, wcfg, budget_s=15.0, warm=1, max_timed=3, min_timed=3)
    if args.cpu_full:
        res["points"][f"ResNet-{args.resnet} B=128"] = _cpu_point(exp, args.resnet, 128, args.image_size, wcfg, budget_s=400.0)
    return res


def main():
    args = parse()
    from simhand_amd import _lib, ops
    from simhand_amd.host import dist as shdist

    if args.lib or args.lib_f16:
        _lib.set_library_paths(args.lib, args.lib_f16)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))  # nothing above touched the GPU: the ranks are ordinary child processes
    # measured parity of THIS build on THIS device (VERDICT r5 "Next" 7): the smoke-size step against the CPU oracle, before anything is timed --
    # and before torch.distributed exists in this process (with a process group up the loss would exchange its rows with ranks that are
    # not running the probe)
    parity = None
    env_rank, env_local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if env_rank == 0 and args.precision != "fp8" and not args.no_parity_probe:
        import __graft_entry__ as entry

        _lib.require_device()
        torch.cuda.set_device(env_local % max(1, torch.cuda.device_count()) if args.share_gpu else env_local)
        parity = entry.parity_probe()
    if args.share_gpu:  # the shared-GPU test arrangement (ranks on one device over gloo, device tensors staged through host memory)
        from tests import _gloo_staging

        rank, local, world = _gloo_staging.init_shared_gpu()
    else:
        rank, local, world = shdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    emu = max(1, args.emulate_world)
    if emu > 1 and world != 1:
        raise SystemExit("--emulate-world runs in ONE process (--gpus 1)")
    _lib.require_device()
    if world > 1:
        # the N > 1 line must be what it says: one rank per GPU over RCCL with every rank present -- anything else fails loudly here
        got = dist.get_world_size()
        if got != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the {dist.get_backend()} backend reports {got} ranks")
        if not args.share_gpu:
            if dist.get_backend() != "nccl":
                raise SystemExit(f"bench.py: N > 1 runs over RCCL (backend nccl); torch.distributed came up with '{dist.get_backend()}'")
            if torch.cuda.device_count() < args.gpus:
                raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} device(s) are visible (one rank per GPU)")
    device = torch.device("cuda", torch.cuda.current_device())  # = LOCAL_RANK (init_from_env set it)
    model = make_model(args, world * emu).to(device).train()
    for kv in args.switch:   # A/B hooks (never set in the default run; recorded in the line's config)
        name, v = kv.split("=", 1)
        ops.test_switch(name, int(v))
    for kv in args.engine:
        name, v = kv.split("=", 1)
        if not hasattr(model.encoder.engine, name):
            raise SystemExit(f"--engine {name}: ResNetEngine has no such attribute")
        setattr(model.encoder.engine, name, bool(int(v)))
    shdist.broadcast_module_state(model)  # replicas start from rank 0's parameters (a no-op at N = 1)

    class _T:
        max_epochs, world_size = 100, world

    model.trainer = _T()
    model.setup("fit")
    (opt,), (sched,) = model.configure_optimizers()
    batch = device_batch(args.per_gpu_batch, args.image_size, 5 + rank, device)
    params = [p for p in model.parameters()]
    reducer, group = None, None
    if emu > 1:  # the rank-local work of an `emu`-rank step, no collective executed
        group = make_emulated_comm(emu)
        model.process_group = group
        reducer = shdist.OverlappedGradReducer(group, wire=args.grad_wire)
        model.encoder.engine.grad_reducer = reducer
    if world > 1:  # backbone gradients go out block by block during the backward pass; the head's follow in allreduce_gradients
        comm_note = None
        if args.comm == "abi" and dist.get_backend() == "nccl":  # bootstrap over torch.distributed, data path on the ABI communicator
            # the two ncclComms (compute stream + side stream) and a first all-reduce through each; EVERY rank must succeed, otherwise all ranks
            # fall back to torch.distributed together (a line that says so is worth more than no line: this path has never met > 1 GPU)
            ok, err = 1, ""
            try:
                group = shdist.RcclComm.from_torch_distributed()
                probe = torch.ones(8, dtype=torch.float32, device=device)
                group.all_reduce_(probe, "sum")
                side = group.side_stream()
                if side is not None:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        group.all_reduce_(probe, "sum", side=True)
                    torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                if abs(float(probe[0]) - float(world) ** (2 if side is not None else 1)) > 1e-3:
                    raise RuntimeError(f"probe all-reduce returned {float(probe[0])} on {world} ranks")
            except Exception as e:  # noqa: BLE001
                ok, err, group = 0, repr(e)[:300], None
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag) == 0:
                if group is not None:
                    group.close()
                group = None
                comm_note = f"torch.distributed (FALLBACK: the ABI communicator failed on at least one rank{': ' + err if err else ''})"
                print(f"bench.py[rank {rank}]: --comm abi unavailable, falling back to torch.distributed: {err or 'another rank failed'}", file=sys.stderr, flush=True)
            else:
                model.process_group = group
        reducer = shdist.OverlappedGradReducer(group, wire=args.grad_wire)
        model.encoder.engine.grad_reducer = reducer
        if args.sync_bn:
            shdist.enable_sync_bn(group)

    from simhand_amd.host.amp import GradScaler

    scaler = GradScaler(enabled=args.precision == "16" and not args.no_loss_scaling)  # the reference's native-AMP loss scaling; a no-op for the other precisions

    def step(i):
        opt.zero_grad(set_to_none=True)
        out = model.training_step(batch, i)
        scaler.scale(out["loss"]).backward()
        shdist.allreduce_gradients(params, group=group, skip=reducer.reduced if reducer is not None else None, wire=args.grad_wire)
        scaler.unscale_(params)
        scaler.step(opt)
        scaler.update()
        sched["scheduler"].step()
        return out["loss"]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events around every launch cost ~1-3 us of queue time each (~1100 launches per step: 1-3 % of the step), so
    # the full per-class breakdown is taken on the LAST WARM-UP step and the timed region records only the class the
    # roofline is quoted on (the conv class with the most time in that warm-up step).
    conv_classes = ("conv_fwd", "conv_dgrad", "conv_wgrad")
    loss = None
    warm_prof = None
    for i in range(args.warmup):
        if i == args.warmup - 1:
            ops.prof_reset()
            ops.prof_set_classes(None)
            ops.prof_enable(True)
        loss = step(i)
    barrier()
    if args.warmup > 0:
        ops.prof_enable(False)
        warm_prof = ops.prof_collect()
        dom = max(conv_classes, key=lambda k: warm_prof[k]["ms"])
        ops.prof_set_classes([dom])
    else:
        ops.prof_set_classes(None)
    ops.prof_reset()
    every = max(1, args.event_every)
    sampler = DeviceStateSampler(torch.cuda.current_device()).start() if rank == 0 else None
    t0 = time.perf_counter()
    for i in range(args.steps):
        ops.prof_enable(i % every == 0)  # live HIP events on the dominant class's launches, in every `every`-th timed step
        loss = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    device_state = sampler.stop() if sampler is not None else None
    ops.prof_enable(False)
    prof = ops.prof_collect()
    ops.prof_set_classes(None)
    final_loss = float(loss.detach())
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # every rank must have timed the same number of steps on the same per-rank batch: a rank that fell out of step would
        # otherwise inflate the aggregate silently
        did = torch.tensor([args.steps, args.warmup, args.per_gpu_batch], dtype=torch.int64, device=device)
        lo, hi = did.clone(), did.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit(f"bench.py: ranks disagree on (steps, warmup, per-gpu batch): min {lo.tolist()} max {hi.tolist()}")
    elapsed = float(t)

    if rank == 0:
        global_pairs = args.per_gpu_batch * world
        value = global_pairs * args.steps / elapsed
        conv = {k: prof[k] for k in conv_classes}
        dom = max(conv, key=lambda k: conv[k]["ms"])
        d = conv[dom]  # measured in the timed region
        peak = {"bf16": BF16_DENSE_PEAK_TFLOPS, "16": BF16_DENSE_PEAK_TFLOPS, "fp8": FP8_DENSE_PEAK_TFLOPS, "32": F32_PEAK_TFLOPS}[args.precision]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        breakdown, bsteps = (warm_prof, 1) if warm_prof is not None else (prof, len(range(0, args.steps, every)))
        all_conv_flops = sum(breakdown[k]["flops"] for k in conv_classes)
        all_conv_ms = sum(breakdown[k]["ms"] for k in conv_classes)
        # HBM view of the step.  The PMC counters (FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes) need rocprofv3's
        # own passes (scripts/refresh_profiles.sh), so the byte counts are STATIC: profiles/hbm_traffic.json records the source hash
        # they were taken at; against a different build they are reported as stale (null), never silently.
        traffic, hbm = None, None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf) and args.per_gpu_batch == 1024 and args.resnet == "50" and args.precision == "bf16" and args.image_size == 224:
            tj = json.load(open(tf))
            stale = tj.get("source_hash") != source_hash()
            src = f"profiles/hbm_traffic.json (static: rocprofv3 --pmc passes of this command at source hash {tj.get('source_hash')}, commit {tj.get('commit')})"
            if stale:
                print(f"bench.py: profiles/hbm_traffic.json was measured at source hash {tj.get('source_hash')}, this build is {source_hash()}: "
                      f"HBM traffic figures withheld (re-run scripts/refresh_profiles.sh)", file=sys.stderr, flush=True)
                hbm = {"bytes_per_step": None, "gbps": None, "peak": 8000.0, "frac": None, "stale": True, "source": src}
            else:
                traffic = (tj.get(dom) or {}).get("bytes_per_launch")
                bps = (tj.get("step") or {}).get("bytes_per_step")
                if bps:
                    gbps = bps / (elapsed / args.steps) / 1e9
                    hbm = {"bytes_per_step": bps, "gbps": gbps, "peak": 8000.0, "frac": gbps / 8000.0, "stale": False, "source": src}
        res = {
            "metric": "hand-image-pairs/sec", "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"bf16": "bf16", "32": "f32", "16": "f16 (loss-scaled)", "fp8": "fp8-e4m3 operands where they pay / bf16"}[args.precision], "data": "synthetic",
            "config": {"workload": f"ResNet-{args.resnet} {args.experiment} contrastive step (fwd+bwd+allreduce+LARS/Adam), "
                                   f"{args.per_gpu_batch} pairs/GPU of 2x{args.image_size}x{args.image_size}x3, linear MPJPE weighting, "
                                   f"crop+rotate un-warp, global negatives",
                       "global_batch": global_pairs, "per_gpu_batch": args.per_gpu_batch, "image_size": args.image_size,
                       "parallelism": f"dp{world}", "loss": final_loss,
                       "parity": parity if parity is not None else ("n/a (the reference has no fp8 path)" if args.precision == "fp8" else "not probed (--no-parity-probe)"),
                       "world_size_backend": dist.get_world_size() if world > 1 else 1,
                       "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if world > 1 else "none",
                       "comm": (comm_note or ("abi (simhand_comm_*: two ncclComms, gradient buckets on the side stream)" if group is not None else "torch.distributed")) if world > 1 else "none",
                       "grad_wire": args.grad_wire if world > 1 else "n/a",
                       "batchnorm": "synchronised" if (world > 1 and args.sync_bn) else "per-rank statistics",
                       "ab_hooks": (args.switch + args.engine + ([f"lib={args.lib}"] if args.lib else []) + ([f"lib_f16={args.lib_f16}"] if args.lib_f16 else []))
                                   or "none (production dispatch)"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "hbm": hbm, "launches": d["count"], "avg_launch_ms": d["ms"] / max(1, d["count"]),
                         "event_steps": f"{len(range(0, args.steps, every))} of the {args.steps} timed steps (every {every})",
                         "all_conv_tflops": all_conv_flops / (all_conv_ms * 1e-3) / 1e12 if all_conv_ms > 0 else 0.0,
                         "step_tflops_per_gpu": TRAIN_GFLOP_PER_PAIR.get(args.resnet, 0.0) * (args.image_size / 224.0) ** 2
                                                * args.per_gpu_batch * args.steps / elapsed / 1e3},
            "device_state": device_state,
            "kernel_ms_per_step": {k: v["ms"] / bsteps for k, v in breakdown.items()},
            "kernel_ms_source": "last warm-up step (events on every launch)" if warm_prof is not None else "timed region",
        }
        if emu > 1:
            # what the wire would add if NOTHING overlapped -- ring all-reduce of the gradient bytes over xGMI (per-link bound,
            # 2 (R - 1) / R x bytes / 153 GB/s) + three latency-bound small all-gathers
            nparam = sum(p.numel() for p in params if p.requires_grad)
            gbytes = nparam * (2 if args.grad_wire == "bf16" else 4)
            ring_ms = 2.0 * (emu - 1) / emu * gbytes / 153e9 * 1e3
            t_emu = elapsed / args.steps * 1e3
            res["emulation"] = {
                "what": f"rank-local work of ONE rank of a {emu}-rank step in one process: {args.per_gpu_batch} pairs through the backbone, loss row "
                        f"block {2 * args.per_gpu_batch} x {2 * args.per_gpu_batch * emu} against a gathered buffer filled by device-local copies, gradient "
                        f"buckets flattened ({args.grad_wire} wire) -- NO collective executed, NO multi-GPU measurement",
                "emulated_world": emu, "ms_per_step_rank_local": t_emu, "grad_bytes_on_wire": gbytes,
                "ring_allreduce_ms_not_overlapped_estimate": ring_ms,
                "small_collectives": "3 all-gathers of <= 1.4 MB per rank (latency-bound, ~0.1 ms together: estimate)",
                "projected_pairs_per_s_upper_bound": args.per_gpu_batch * emu / (t_emu * 1e-3),
                "projected_pairs_per_s_if_nothing_overlaps": args.per_gpu_batch * emu / ((t_emu + ring_ms + 0.1) * 1e-3),
                "note": "compare ms_per_step_rank_local with the N = 1 line of the same box: the ratio is the UPPER bound on the data-parallel "
                        "efficiency of the compute side (global-negatives loss + bucket handling); RCCL time and contention come on top",
            }
            res["config"]["parallelism"] = f"dp1 emulating rank 0 of dp{emu} (no collective executed)"
            res["config"]["global_batch"] = args.per_gpu_batch  # what THIS process processed per step
        if not args.no_cpu_baseline and world == 1 and emu == 1:  # the CPU reference is timed at N = 1 only (the other ranks would idle behind it)
            res["cpu_baseline"] = cpu_baseline(args)
        if args.precision == "bf16" and world == 1 and emu == 1 and not args.no_cpu_baseline and not (args.switch or args.engine):
            # bf16 is the dtype BASELINE names and this line reports; the RECOMMENDED training mode is --precision 16 (the reference's own
            # policy: fp16 storage + dynamic loss scaling) -- bf16 activations cost training quality (profiles/r05_stability_160steps.md:
            # last-16 mean loss 4.47 vs 4.12 after 160 steps).  Its step time on this box, from a child process started after every timed
            # region of this one has finished (an ordinary child: nothing is exec'ed over the initialised runtime).
            res["recommended_precision"] = {"precision": "16", "why": "fp16 storage + GradScaler (src/experiments/main.py:158-159) tracks the fp32 loss curve; "
                                            "bf16 storage does not (profiles/r05_stability_160steps.md)", "ms_per_step": None, "pairs_per_s": None}
            try:
                cmd = [sys.executable, os.path.abspath(__file__), "--precision", "16", "--steps", str(max(3, min(args.steps, 8))), "--warmup", "2",
                       "--no-cpu-baseline", "--per-gpu-batch", str(args.per_gpu_batch), "--resnet", args.resnet, "--image-size", str(args.image_size),
                       "--experiment", args.experiment]
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
                d16 = json.loads(line)
                res["recommended_precision"].update(ms_per_step=d16["ms_per_step"], pairs_per_s=d16["value"], dtype=d16["dtype"])
            except Exception as e:  # noqa: BLE001 -- the companion figure must never cost the headline line
                res["recommended_precision"]["error"] = repr(e)[:200]
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
