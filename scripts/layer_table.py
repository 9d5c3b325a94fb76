"""Per-layer table for the 23 distinct convolution shapes of ResNet-50 @224^2 at the benchmarked
batch (2048 images): forward / data gradient / weight gradient through the library's own dispatch -- kernel route,
microseconds, TFLOP/s, algorithmic GB/s, the bound that applies (HBM below the 397 FLOP/B ridge of 2.5 PF / 6.3 TB/s,
MFMA above) and the fraction of that bound.  Runs ON THE GPU BOX:

    python scripts/layer_table.py [--images 2048] [--out profiles/r02_layer_table.md]

Timing: HIP events on torch's current stream (the stream the kernels are launched on), median of 5 after 2 warm-ups.
The forward of the layers whose BatchNorm is folded (1x1, cout >= 2 cin) is timed in the form the engine runs
(simhand_conv2d_fwd_bnact: BN + residual + ReLU epilogue); the data gradient is the plain store form (the engine's
variants add epilogue reads that are listed in DESIGN 3)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from simhand_amd import ops  # noqa: E402

SHAPES = [  # (cin, cout, k, stride, hin, count)
    (64, 64, 1, 1, 56, 1), (64, 64, 3, 1, 56, 3), (64, 256, 1, 1, 56, 4), (256, 64, 1, 1, 56, 2), (256, 128, 1, 1, 56, 1),
    (128, 128, 3, 2, 56, 1), (128, 512, 1, 1, 28, 4), (256, 512, 1, 2, 56, 1), (512, 128, 1, 1, 28, 3), (128, 128, 3, 1, 28, 3),
    (512, 256, 1, 1, 28, 1), (256, 256, 3, 2, 28, 1), (256, 1024, 1, 1, 14, 6), (512, 1024, 1, 2, 28, 1), (1024, 256, 1, 1, 14, 5),
    (256, 256, 3, 1, 14, 5), (1024, 512, 1, 1, 14, 1), (512, 512, 3, 2, 14, 1), (512, 2048, 1, 1, 7, 3), (1024, 2048, 1, 2, 14, 1),
    (2048, 512, 1, 1, 7, 2), (512, 512, 3, 1, 7, 2),
]


def shapes_for(resnet: str, image: int):
    """The distinct convolution shapes (stem excluded) of a Bottleneck ResNet at `image`^2 with their counts -- SHAPES above is
    shapes_for("50", 224).  Same enumeration order as the table of SURVEY Appendix C."""
    counts = {"50": [3, 4, 6, 3], "101": [3, 4, 23, 3], "152": [3, 8, 36, 3]}[resnet]
    out, cin, h = [], 64, image // 4
    for i, nb in enumerate(counts):
        p, s = 64 << i, (1 if i == 0 else 2)
        ho = h // s
        out.append((cin, p, 1, 1, h, 1))                       # entry conv1
        out.append((p, p, 3, s, h, 1))                          # entry conv2 (v1.5: the stride sits here)
        out.append((p, 4 * p, 1, 1, ho, nb))                    # conv3 of every block
        out.append((cin, 4 * p, 1, s, h, 1))                    # shortcut
        if nb > 1:
            out.append((4 * p, p, 1, 1, ho, nb - 1))            # conv1 of the identity blocks
            out.append((p, p, 3, 1, ho, nb - 1))                # conv2 of the identity blocks
        cin, h = 4 * p, ho
    merged = {}
    for c_in, c_out, k, s, hh, n in out:
        merged[(c_in, c_out, k, s, hh)] = merged.get((c_in, c_out, k, s, hh), 0) + n
    return [key + (n,) for key, n in merged.items()]


PEAK_TF, PEAK_GB = 2500.0, 6300.0  # dense bf16 MFMA; achievable HBM copy rate (MI355X_MICROARCH.md)


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def route_of(fn):
    ops.route_reset()
    fn()
    torch.cuda.synchronize()
    skip = ("fwd_bnact", "dgrad_fused_sums", "dgrad_parity", "dgrad_concat", "wgrad_colsum", "subsample2", "scatter2_add")
    return "+".join(k for k, v in ops.route_counts().items() if v and k not in skip)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2048)
    ap.add_argument("--out", default=None)
    ap.add_argument("--resnet", default="50", choices=["50", "101", "152"])
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--alternates", action="store_true", help="also time every op under the other kernel routes (tuning hooks)")
    args = ap.parse_args()
    from simhand_amd import _lib
    lib = _lib.load()
    alts = {"default": lambda: None, "no256": lambda: lib.simhand_test_igemm256_enable(0), "force256": lambda: lib.simhand_test_igemm256_enable(2),
            "no_c64": lambda: lib.simhand_test_conv3x3_c64_enable(0), "no_wgrad3": lambda: lib.simhand_test_wgrad3x3_enable(0),
            "notail": lambda: lib.simhand_test_igemm256_split_tail(0)}
    n, dt, dev = args.images, torch.bfloat16, "cuda"
    rows = ["| layer (cin,cout,k,s,Hin) x count | op | kernel route | us | TFLOP/s | GB/s (algorithmic) | bound | fraction of bound |",
            "|---|---|---|---|---|---|---|---|"]
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    shapes = SHAPES if (args.resnet, args.image_size) == ("50", 224) else shapes_for(args.resnet, args.image_size)
    for cin, cout, k, s, h, cnt in shapes:
        pad = 1 if k == 3 else 0
        d = ops.conv_desc(n, h, h, cin, cout, k, k, s, pad, dt)
        x = torch.randn(n, h, h, cin, device=dev).to(dt)
        w = torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5
        wk, wc = ops.pack_krsc(w, dt), ops.pack_crsk(w, dt)
        dy = torch.randn(n, d.ho, d.wo, cout, device=dev).to(dt)
        m = n * d.ho * d.wo
        flops = 2.0 * m * cout * cin * k * k
        folded = k == 1 and cout >= 2 * cin
        if folded:
            st = ops.BNState(cout, dev)
            st.scale.fill_(1.0)
            st.shift.fill_(0.0)
            res = torch.randn(n, d.ho, d.wo, cout, device=dev).to(dt) if s == 1 and cout == 4 * cin else None
            f_fwd = lambda: ops.conv2d_fwd_bnact(d, x, wk, st, res is not None, res, want_mask=res is not None)  # noqa: E731
            b_fwd = 2.0 * (x.numel() + m * cout * (2 if res is not None else 1) + wk.numel()) + (m * cout / 8 if res is not None else 0)
        else:
            f_fwd = lambda: ops.conv2d_fwd(d, x, wk, want_stats=True)  # noqa: E731
            b_fwd = 2.0 * (x.numel() + m * cout + wk.numel())
        f_dg = lambda: ops.conv2d_dgrad(d, dy, wc)  # noqa: E731
        b_dg = 2.0 * (x.numel() + m * cout + wk.numel())
        f_wg = lambda: ops.conv2d_wgrad_oihw(d, x, dy, (cout, cin, k, k))  # noqa: E731
        b_wg = 2.0 * (x.numel() + m * cout) + 4.0 * wk.numel()
        tag = ""
        if k == 1 and s == 2:
            # the stage-entry shortcuts, in the form the engine runs them (host/resnet_model.py _conv_bn_folded / _ds_bwd_folded): ONE
            # subsample pass, then dense 1x1 / stride-1 launches over x[:, ::2, ::2] -- forward with the folded BN epilogue, both
            # gradients at the output resolution, the data gradient scatter-added onto the even pixels of the main branch's dx
            x_in = ops.subsample2(x)
            dd = ops.conv_desc(n, d.ho, d.wo, cin, cout, 1, 1, 1, 0, dt)
            dxm = torch.zeros(n, h, h, cin, device=dev, dtype=dt)
            us_sub = timed(lambda: ops.subsample2(x))
            b_sub = 2.0 * 2 * x_in.numel()
            rows.append(f"| ({cin},{cout},{k},{s},{h}) x{cnt} | subsample x[:, ::2, ::2] (once: forward, Gram, both gradients) | subsample2 | {us_sub:.0f} | - | "
                        f"{b_sub / us_sub / 1e3:.0f} | HBM | {b_sub / us_sub / 1e3 / PEAK_GB:.2f} |")
            tot["fwd"] += us_sub * cnt
            f_fwd = lambda: ops.conv2d_fwd_bnact(dd, x_in, wk, st, False, None)  # noqa: E731
            b_fwd = 2.0 * (x_in.numel() + m * cout + wk.numel())
            f_dg = lambda: ops.scatter2_add(ops.conv2d_dgrad(dd, dy, wc), dxm)  # noqa: E731
            b_dg = 2.0 * (m * cout + 4 * x_in.numel() + wk.numel())  # dy, the dense result written + read, dx's even pixels read + written
            f_wg = lambda: ops.conv2d_wgrad_oihw(dd, x_in, dy, (cout, cin, 1, 1))  # noqa: E731
            b_wg = 2.0 * (x_in.numel() + m * cout) + 4.0 * wk.numel()
            tag = ", dense over the subsampled input"
        for op, fn, by in (("fwd" + (" +bnact" if folded else "") + tag, f_fwd, b_fwd), ("dgrad" + (" + scatter-add onto dx" if tag else ""), f_dg, b_dg),
                           ("wgrad" + tag, f_wg, b_wg)):
            us = timed(fn)
            rt = route_of(fn)
            tf, gb = flops / us / 1e6, by / us / 1e3
            ai = flops / by
            bound = "MFMA" if ai >= PEAK_TF * 1e3 / PEAK_GB else "HBM"
            frac = tf / PEAK_TF if bound == "MFMA" else gb / PEAK_GB
            rows.append(f"| ({cin},{cout},{k},{s},{h}) x{cnt} | {op} | {rt} | {us:.0f} | {tf:.0f} | {gb:.0f} | {bound} ({ai:.0f} FLOP/B) | {frac:.2f} |")
            tot[op.split()[0].rstrip(',')] += us * cnt
            if args.alternates:
                for an, setter in alts.items():
                    if an == "default":
                        continue
                    ops.hooks_reset()
                    setter()
                    rt2 = route_of(fn)
                    if rt2 != rt:
                        us2 = timed(fn)
                        rows.append(f"|   alt {an} | {op} | {rt2} | {us2:.0f} | {flops / us2 / 1e6:.0f} | {by / us2 / 1e3:.0f} | | {us2 / us:.2f}x time |")
                ops.hooks_reset()
        del x, dy
        torch.cuda.empty_cache()
    rows.append("")
    rows.append(f"sum over the {sum(x[-1] for x in shapes)} non-stem convolutions of ResNet-{args.resnet} @ {args.image_size}^2 (count-weighted): fwd {tot['fwd']/1e3:.1f} ms, dgrad {tot['dgrad']/1e3:.1f} ms, "
                f"wgrad {tot['wgrad']/1e3:.1f} ms at {n} images")
    text = "\n".join(rows)
    print(text)
    if args.out:
        open(args.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
