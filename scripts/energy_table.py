"""Steady-state power, clock and ENERGY per launch of the convolution kernels on the ResNet-50 shapes (2048 images): every op is
looped for ~1.2 s while a host thread samples the amdgpu sysfs nodes (bench.DeviceStateSampler) -- the step runs at the socket's power limit
(profiles/r05_power_wall.md), so joules per launch, not cycles, are what a kernel change has to lower.  ON THE GPU BOX:

    python scripts/energy_table.py [--out profiles/r05_energy_table.md] [--seconds 1.2]

J = mean socket power x time per launch (idle power is not subtracted; the idle reading is printed first)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from simhand_amd import ops  # noqa: E402
from scripts.layer_table import SHAPES  # noqa: E402


def loop(fn, seconds):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    # calibrate the launch count for `seconds`
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 10
    n = max(20, int(seconds / per))
    s = bench.DeviceStateSampler(torch.cuda.current_device(), period_s=0.01).start()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    st = s.stop()
    # drop the ramp: the first 30 % of the samples
    k = len(s.samples) * 3 // 10
    pw = [x[1] for x in s.samples[k:] if x[1] is not None]
    ck = [x[0] for x in s.samples[k:] if x[0] is not None]
    return dt, (sum(pw) / len(pw) if pw else None), (sum(ck) / len(ck) if ck else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2048)
    ap.add_argument("--seconds", type=float, default=1.2)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dt_ = torch.bfloat16
    N = a.images
    lines = []
    s = bench.DeviceStateSampler(torch.cuda.current_device(), period_s=0.01).start()
    time.sleep(1.0)
    idle = s.stop()
    lines.append(f"idle (1 s, nothing queued): {idle.get('socket_power_w')} W, sclk {idle.get('sclk_mhz')}")
    lines += ["", "| layer (cin,cout,k,s,Hin) x count | op | us | TFLOP/s | W (steady) | sclk MHz | J / launch | pJ / algorithmic FLOP | J / step (x count) |", "|---|---|---|---|---|---|---|---|---|"]
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for cin, cout, k, st_, h, cnt in SHAPES:
        pad = k // 2
        d = ops.conv_desc(N, h, h, cin, cout, k, k, st_, pad, dt_)
        x = torch.randn(N, h, h, cin, device="cuda").to(dt_).relu_()   # post-ReLU statistics: half of the activations are zero
        w = torch.randn(cout, cin, k, k, device="cuda") * (1.0 / (cin * k * k) ** 0.5)
        wk, wt = ops.pack_krsc(w, dt_), ops.pack_crsk(w, dt_)
        dy = torch.randn(N, d.ho, d.wo, cout, device="cuda").to(dt_)
        fl = 2.0 * N * d.ho * d.wo * cout * cin * k * k
        for name, fn in (("fwd", lambda: ops.conv2d_fwd(d, x, wk, True)), ("dgrad", lambda: ops.conv2d_dgrad(d, dy, wt)),
                         ("wgrad", lambda: ops.conv2d_wgrad(d, x, dy))):
            t, pw, ck = loop(fn, a.seconds)
            j = pw * t if pw else float("nan")
            tot[name] += j * cnt
            lines.append(f"| ({cin},{cout},{k},{st_},{h}) x{cnt} | {name} | {t*1e6:.0f} | {fl/t/1e12:.0f} | {pw:.0f} | {ck:.0f} | {j:.3f} | {j/fl*1e12:.2f} | {j*cnt:.2f} |")
            print(lines[-1], flush=True)
        del x, dy
    lines.append("")
    lines.append(f"sum over the 52 non-stem convolutions (count-weighted, plain forms): fwd {tot['fwd']:.1f} J, dgrad {tot['dgrad']:.1f} J, wgrad {tot['wgrad']:.1f} J per step")
    text = "\n".join(lines)
    print(text)
    if a.out:
        open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
