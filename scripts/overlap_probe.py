"""Does an MFMA-bound kernel overlap with HBM-bound kernels when they run on two streams -- plain, and with the CUs partitioned
(hipExtStreamCreateWithCUMask)?  A = 3x3 weight gradient (wgrad3x3, matrix-core bound), B = HBM-bound launches of the same block
(BatchNorm-apply over the wide tensor, 1x1 narrow->wide data gradient).  Prints serial vs concurrent wall time.
usage: python scripts/overlap_probe.py [stage]   (stage 3 = 256 ch @14^2, 2 = 128 @28^2, 1 = 64 @56^2)"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from simhand_amd import ops  # noqa: E402

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w, h = {1: (64, 56), 2: (128, 28), 3: (256, 14), 4: (512, 7)}[stage]
N = 2048
dt = torch.bfloat16
hip = C.CDLL("libamdhip64.so")


def masked_stream(pred):
    words = (C.c_uint32 * 8)()
    n = 0
    for i in range(256):
        if pred(i):
            words[i // 32] |= 1 << (i % 32)
            n += 1
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value), n


d3 = ops.conv_desc(N, h, h, w, w, 3, 3, 1, 1, dt)
d1 = ops.conv_desc(N, h, h, 4 * w, w, 1, 1, 1, 0, dt)  # conv1: wide -> narrow; its dgrad writes the wide tensor
a1 = torch.randn(N, h, h, w, device="cuda").to(dt)
dy2 = torch.randn(N, h, h, w, device="cuda").to(dt)
wide = torch.randn(N, h, h, 4 * w, device="cuda").to(dt)
wide2 = torch.empty_like(wide)
w1 = torch.randn(w, 4 * w, 1, 1, device="cuda") * 0.05
w1t = ops.pack_crsk(w1, dt)
w3 = torch.randn(w, w, 3, 3, device="cuda") * 0.05
w3t = ops.pack_crsk(w3, dt)


def A():
    ops.conv2d_wgrad(d3, a1, dy2)


def B_copy():
    wide2.copy_(wide)


def B_dgrad():
    ops.conv2d_dgrad(d1, dy2, w1t)


def B_dgrad3():  # the MFMA-bound data gradient of the 3x3 (control: two matrix-bound kernels should NOT overlap)
    ops.conv2d_dgrad(d3, dy2, w3t)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


def both(sa, sb, fb, nb):
    def run():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        sa.wait_event(ev)
        sb.wait_event(ev)
        with torch.cuda.stream(sa):
            A()
            ea = torch.cuda.Event(); ea.record(sa)
        with torch.cuda.stream(sb):
            for _ in range(nb):
                fb()
            eb = torch.cuda.Event(); eb.record(sb)
        cur.wait_event(ea)
        cur.wait_event(eb)
    return run


tA = timed(A)
print(f"stage {stage}: A wgrad3x3 alone {tA:.0f} us")
for name, fb in (("copy wide", B_copy), ("1x1 dgrad narrow->wide", B_dgrad), ("3x3 dgrad (control)", B_dgrad3)):
    tB = timed(fb)
    nb = max(1, round(tA / tB))
    serial = timed(lambda: (A(), [fb() for _ in range(nb)]))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    conc = timed(both(s1, s2, fb, nb))
    line = f"  B = {name}: alone {tB:.0f} us x{nb}; serial {serial:.0f}; two streams {conc:.0f}"
    for frac in (4, 3, 2):  # A gets 1/frac of the CU rows, B the rest
        sa, na = masked_stream(lambda i: (i // 8) % frac == 0)
        sb, nbb = masked_stream(lambda i: (i // 8) % frac != 0)
        c = timed(both(sa, sb, fb, nb))
        line += f"; masks {na}/{nbb} CUs {c:.0f}"
    print(line)
# B alone on 192 / 128 CUs: do HBM-bound kernels keep their rate on fewer CUs?
for frac in (4, 2):
    sb, nbb = masked_stream(lambda i: (i // 8) % frac != 0)
    with torch.cuda.stream(sb):
        tc, td = timed(B_copy), timed(B_dgrad)
    sa, na = masked_stream(lambda i: (i // 8) % frac == 0)
    with torch.cuda.stream(sa):
        ta = timed(A)
    print(f"  on {nbb} CUs: copy {tc:.0f} us, 1x1 dgrad {td:.0f} us;  A on {na} CUs: {ta:.0f} us")
