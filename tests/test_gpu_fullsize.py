"""Parity AT THE BENCHMARKED SIZE: every distinct convolution shape of ResNet-50 @224^2 (SURVEY Appendix C;
reference: src/models/resnet_model.py:13-58 -> torchvision resnet50) at N = 2048 images in bf16, through the
PRODUCTION dispatch -- no tuning hook is touched, and the kernel route each call takes is asserted against the route
the committed per-layer table (profiles/r0X_layer_table.md) names.  These are the launches bench.py times: the
size-dependent decisions (split-K sized to one resident round, 224-row tiles at 401 408 pixels, persistent block counts,
XCD-aware tile order, the ragged last round) only exist here, the small-size tests never reach them.

Checker: ATen CPU fp32 (conv / conv-backward) on the bf16-rounded operands.  A convolution's output and data gradient
are batch-independent, so forward / dgrad are compared on 16 sampled images (first, middle, and the LAST eight images =
the rows next to the ragged-round boundary M - 256 .. M); the weight gradient is compared in full (all 2048 images,
chunked on the host).  Fused forms that the engine uses at this size (BN + residual + ReLU epilogue, two-segment folded
data gradient with bias and fused sums, masked-residual merge with masked store, Gram launch with the BatchNorm-apply in
its loader, chained conv1) are checked the same way."""
import math
import time

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
N = 2048
DT = torch.bfloat16
SAMPLE = [0, 1, 2, 3, 509, 1023, 1024, 1025] + list(range(N - 8, N))

# (cin, cout, k, stride, hin): routes the production dispatch takes at 2048 images (profiles/r03_layer_table.md)
SHAPES = {
    (64, 64, 1, 1, 56): ("gemm1x1_fwd", "gemm1x1_dgrad", "wgrad_plain"),
    (64, 64, 3, 1, 56): ("c64_fwd", "c64_dgrad", "wgrad3x3"),
    (64, 256, 1, 1, 56): ("gemm1x1_fwd", "gemm1x1_dgrad", "wgrad_plain"),
    (256, 64, 1, 1, 56): ("gemm1x1_fwd", "gemm1x1_dgrad", "wgrad_plain"),
    (256, 128, 1, 1, 56): ("gemm1x1_fwd", "gemm1x1_dgrad", "wgrad_plain"),
    (128, 128, 3, 2, 56): ("igemm128_fwd", "r128_dgrad", "wgrad3x3"),   # round 4: the four parity classes on the LDS-ring kernel
    (128, 512, 1, 1, 28): ("gemm1x1_fwd", "n128_dgrad", "wgrad_plain"),   # round 4: 128 x 128 LDS-DMA tiles, two blocks per CU
    (256, 512, 1, 2, 56): ("igemm256_fwd", "igemm128_dgrad", "wgrad_generic"),
    (512, 128, 1, 1, 28): ("n128_fwd", "gemm1x1_dgrad", "wgrad_plain"),
    (128, 128, 3, 1, 28): ("r128_fwd", "r128_dgrad", "wgrad3x3"),
    (512, 256, 1, 1, 28): ("igemm256_fwd", "gemm1x1_dgrad", "wgrad_plain"),
    (256, 256, 3, 2, 28): ("igemm256_fwd", "igemm256_dgrad", "wgrad3x3"),
    (256, 1024, 1, 1, 14): ("gemm1x1_fwd", "igemm256_dgrad", "wgrad_plain"),
    (512, 1024, 1, 2, 28): ("igemm256_fwd", "igemm128_dgrad", "wgrad_generic"),
    (1024, 256, 1, 1, 14): ("igemm256_fwd", "gemm1x1_dgrad", "wgrad_plain"),
    (256, 256, 3, 1, 14): ("igemm256_fwd", "igemm256_dgrad", "wgrad3x3"),
    (1024, 512, 1, 1, 14): ("igemm256_fwd", "igemm256_dgrad", "wgrad_plain"),
    (512, 512, 3, 2, 14): ("igemm256_fwd+igemm256_tail", "igemm256_dgrad", "wgrad3x3"),
    (512, 2048, 1, 1, 7): ("igemm256_fwd", "igemm256_dgrad+igemm256_tail", "wgrad_plain"),
    (1024, 2048, 1, 2, 14): ("igemm256_fwd", "igemm128_dgrad", "wgrad_generic"),
    (2048, 512, 1, 1, 7): ("igemm256_fwd+igemm256_tail", "igemm256_dgrad", "wgrad_plain"),
    (512, 512, 3, 1, 7): ("igemm256_fwd+igemm256_tail", "igemm256_dgrad+igemm256_tail", "wgrad3x3"),
}
# forward route of the same shapes in the form the engine runs them when their BatchNorm is folded (conv + BN (+res+ReLU) epilogue)
BNACT_ROUTES = {
    (64, 256, 1, 1, 56): "gemm1x1_fwd_bnact", (128, 512, 1, 1, 28): "gemm1x1_fwd_bnact", (256, 1024, 1, 1, 14): "gemm1x1_fwd_bnact",
    (512, 2048, 1, 1, 7): "igemm256_fwd", (256, 512, 1, 2, 56): "igemm256_fwd", (512, 1024, 1, 2, 28): "igemm256_fwd",
    (1024, 2048, 1, 2, 14): "igemm256_fwd",
}
_AUX = ("fwd_bnact", "dgrad_fused_sums", "dgrad_parity", "dgrad_concat", "wgrad_colsum")
IDS = ["x".join(map(str, s)) for s in SHAPES]


def _routes(ops):
    return "+".join(k for k, v in ops.route_counts().items() if v and k not in _AUX)


def _operands(shape, seed=0, need=("x", "w", "dy")):
    from simhand_amd import ops

    cin, cout, k, s, h = shape
    pad = 1 if k == 3 else 0
    d = ops.conv_desc(N, h, h, cin, cout, k, k, s, pad, DT)
    g = torch.Generator(device=DEV).manual_seed(1000 + seed + sum(shape))
    out = {"d": d, "pad": pad}
    if "x" in need:
        out["x"] = torch.randn(N, h, h, cin, device=DEV, generator=g).to(DT)
    if "w" in need:
        out["w"] = (torch.randn(cout, cin, k, k, device=DEV, generator=g) / math.sqrt(cin * k * k)).to(DT).float()
    if "dy" in need:
        out["dy"] = torch.randn(N, d.ho, d.wo, cout, device=DEV, generator=g).to(DT)
    return out


def _nchw(t, idx):
    """Sampled images of an NHWC device tensor as an fp32 NCHW host tensor (channels-last strides)."""
    return t[idx].float().cpu().permute(0, 3, 1, 2)


def _close(got, want, tol, tag):
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    assert err <= tol * scale + 1e-6, f"{tag}: err {err:.3e} of scale {scale:.3e}"


@pytest.mark.parametrize("shape", list(SHAPES), ids=IDS)
def test_fullsize_forward(shape):
    from simhand_amd import ops

    cin, cout, k, s, h = shape
    o = _operands(shape, need=("x", "w"))
    d = o["d"]
    wk = ops.pack_krsc(o["w"], DT)
    ops.hooks_reset()
    ops.route_reset()
    y, part = ops.conv2d_fwd(d, o["x"], wk, want_stats=True)
    torch.cuda.synchronize()
    assert _routes(ops) == SHAPES[shape][0], (_routes(ops), SHAPES[shape][0])
    idx = torch.tensor(SAMPLE, device=DEV)
    want = F.conv2d(_nchw(o["x"], idx), o["w"].cpu(), stride=s, padding=o["pad"])
    _close(_nchw(y, idx), want, 1e-2, "fwd")
    # the fused BatchNorm partial sums over ALL 2048 images (of the fp32 accumulators; y is their bf16 rounding)
    m = N * d.ho * d.wo
    yf = y.view(m, cout).float()
    rms = float(yf.pow(2).mean().sqrt())
    s1, s2 = part[:, 0].double().sum(0) / m, part[:, 1].double().sum(0) / m
    assert (s1 - yf.double().mean(0)).abs().max().item() <= 2e-3 * rms
    want2 = yf.double().pow(2).mean(0)
    assert ((s2 - want2).abs() / want2).max().item() <= 5e-3


@pytest.mark.parametrize("shape", list(BNACT_ROUTES), ids=["x".join(map(str, s)) for s in BNACT_ROUTES])
def test_fullsize_forward_bn_residual_relu_epilogue(shape):
    """The folded-forward form: out = relu(conv(x) * scale + shift + residual) + its ReLU bit mask (identity blocks), and the
    plain BN epilogue of the stride-2 shortcuts."""
    from simhand_amd import ops

    cin, cout, k, s, h = shape
    o = _operands(shape, seed=1, need=("x", "w"))
    d = o["d"]
    wk = ops.pack_krsc(o["w"], DT)
    g = torch.Generator(device=DEV).manual_seed(7)
    st = ops.BNState(cout, DEV)
    st.scale.copy_(torch.rand(cout, device=DEV, generator=g) + 0.5)
    st.shift.copy_(torch.randn(cout, device=DEV, generator=g) * 0.3)
    with_res = s == 1
    res = torch.randn(N, d.ho, d.wo, cout, device=DEV, generator=g).to(DT) if with_res else None
    ops.hooks_reset()
    ops.route_reset()
    if with_res:
        out, mask = ops.conv2d_fwd_bnact(d, o["x"], wk, st, True, res, want_mask=True)
    else:
        out, mask = ops.conv2d_fwd_bnact(d, o["x"], wk, st, False, None), None
    torch.cuda.synchronize()
    assert _routes(ops) == BNACT_ROUTES[shape], _routes(ops)
    idx = torch.tensor(SAMPLE, device=DEV)
    want = F.conv2d(_nchw(o["x"], idx), o["w"].cpu(), stride=s, padding=0)
    want = want * st.scale.cpu().view(1, -1, 1, 1) + st.shift.cpu().view(1, -1, 1, 1)
    if with_res:
        want = (want + _nchw(res, idx)).clamp_min(0)
    _close(_nchw(out, idx), want, 1e-2, "fwd_bnact")
    if with_res:  # the bit mask is exactly (out > 0), for every one of the N * ho * wo * cout outputs
        m = N * d.ho * d.wo
        bits = (out.view(m, cout // 8, 8) > 0).to(torch.uint8)
        packed = (bits << torch.arange(8, device=DEV, dtype=torch.uint8)).sum(-1).to(torch.uint8)
        assert torch.equal(packed, mask)


@pytest.mark.parametrize("shape", list(SHAPES), ids=IDS)
def test_fullsize_dgrad(shape):
    from simhand_amd import ops

    cin, cout, k, s, h = shape
    o = _operands(shape, need=("w", "dy"))
    d = o["d"]
    wc = ops.pack_crsk(o["w"], DT)
    ops.hooks_reset()
    ops.route_reset()
    dx = ops.conv2d_dgrad(d, o["dy"], wc)
    torch.cuda.synchronize()
    assert _routes(ops) == SHAPES[shape][1], (_routes(ops), SHAPES[shape][1])
    idx = torch.tensor(SAMPLE, device=DEV)
    want = torch.nn.grad.conv2d_input((len(SAMPLE), cin, h, h), o["w"].cpu(), _nchw(o["dy"], idx).contiguous(), stride=s, padding=o["pad"])
    _close(_nchw(dx, idx), want, 1e-2, "dgrad")
    assert bool(torch.isfinite(dx.float().sum()))  # no NaN / inf anywhere in the 2048 images


def test_fullsize_ring_kernel_dgrad_with_fused_sums():
    """The 128 -> 128 3x3 ring kernel in the form the step runs it for conv2 of stage 2: data gradient + the BatchNorm-backward sums of
    conv1's unit (ReLU gate recomputed from its raw output), 2048 images: dx bit-equal to the plain launch, sums against a direct
    fp64 reduction of the stored dx over all rows."""
    from simhand_amd import ops

    shape = (128, 128, 3, 1, 28)
    o = _operands(shape, need=("w", "dy"))
    d = o["d"]
    wc = ops.pack_crsk(o["w"], DT)
    g = torch.Generator(device=DEV).manual_seed(77)
    y_prev = torch.randn(N, 28, 28, 128, device=DEV, generator=g).to(DT)
    st = ops.BNState(128, DEV)
    st.scale.copy_(torch.randn(128, device=DEV, generator=g))
    st.shift.copy_(torch.randn(128, device=DEV, generator=g) * 0.3)
    ops.hooks_reset()
    ops.route_reset()
    dx0 = ops.conv2d_dgrad(d, o["dy"], wc)
    dx, part = ops.conv2d_dgrad_fused(d, o["dy"], wc, y_prev, st, None)
    rc = ops.route_counts()
    assert rc["r128_dgrad"] == 2 and torch.equal(dx, dx0), rc
    assert part.shape[0] == (N * 29 * 29 + 255) // 256
    yf = y_prev.view(-1, 128).float()
    gq = torch.where(yf * st.scale + st.shift > 0, dx.view(-1, 128).float(), torch.zeros_like(yf))
    s1, s2 = gq.double().sum(0), (gq.double() * yf.double()).sum(0)
    got1, got2 = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
    assert (got1 - s1).abs().max().item() <= 1e-5 * s1.abs().max().item() + 1e-3
    assert (got2 - s2).abs().max().item() <= 1e-5 * s2.abs().max().item() + 1e-3


def _cpu_wgrad(x, dy, wshape, stride, pad, chunk=128):
    """ATen CPU fp32 convolution-backward (weight only) of bf16-rounded NHWC device operands, accumulated over image chunks."""
    acc = torch.zeros(wshape, dtype=torch.float64)
    w0 = torch.zeros(wshape)
    for i in range(0, x.shape[0], chunk):
        xs = x[i:i + chunk].float().cpu().permute(0, 3, 1, 2)
        gs = dy[i:i + chunk].float().cpu().permute(0, 3, 1, 2)
        _, gw, _ = torch.ops.aten.convolution_backward(gs, xs, w0, None, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1,
                                                        [False, True, False])
        acc += gw.double()
    return acc.float()


@pytest.mark.parametrize("shape", list(SHAPES), ids=IDS)
def test_fullsize_wgrad(shape):
    from simhand_amd import ops

    cin, cout, k, s, h = shape
    o = _operands(shape, need=("x", "dy"))
    d = o["d"]
    ops.hooks_reset()
    ops.route_reset()
    dw = ops.conv2d_wgrad_oihw(d, o["x"], o["dy"], (cout, cin, k, k))
    torch.cuda.synchronize()
    assert _routes(ops) == SHAPES[shape][2], (_routes(ops), SHAPES[shape][2])
    t0 = time.time()
    want = _cpu_wgrad(o["x"], o["dy"], (cout, cin, k, k), s, o["pad"])
    print(f"cpu wgrad reference {shape}: {time.time() - t0:.1f} s")
    _close(dw.cpu(), want, 2e-3, "wgrad")


def test_fullsize_stem_forward_and_wgrad():
    """7x7/2 stem at 2048 x 224^2: the persistent direct kernel from the zero-padded NHWC4 batch (two views handed over as a pair)."""
    from simhand_amd import ops

    g = torch.Generator(device=DEV).manual_seed(3)
    v1 = torch.randn(N // 2, 3, 224, 224, device=DEV, generator=g)
    v2 = torch.randn(N // 2, 3, 224, 224, device=DEV, generator=g)
    w = (torch.randn(64, 3, 7, 7, device=DEV, generator=g) / math.sqrt(147)).to(DT).float()
    xp = ops.stem_pad_input((v1, v2), DT)
    wp = ops.stem_pack_weights(w, DT)
    ops.hooks_reset()
    ops.route_reset()
    y, part = ops.stem_conv_fwd(xp, wp, 224, 224, want_stats=True)
    torch.cuda.synchronize()
    assert ops.route_counts()["stem_fwd"] == 1
    idx = SAMPLE
    xs = torch.stack([(v1[i] if i < N // 2 else v2[i - N // 2]) for i in idx]).to(DT).float().cpu()
    want = F.conv2d(xs, w.cpu(), stride=2, padding=3)
    _close(_nchw(y, torch.tensor(idx, device=DEV)), want, 1e-2, "stem fwd")
    m = N * 112 * 112
    yf = y.view(m, 64)
    s1 = part[:, 0].double().sum(0) / m
    mean = torch.stack([yf[i:i + m // 8].float().sum(0).double() for i in range(0, m, m // 8)]).sum(0) / m
    assert (s1 - mean).abs().max().item() <= 2e-3
    dy = torch.randn(N, 112, 112, 64, device=DEV, generator=g).to(DT)
    ops.route_reset()
    dw = ops.stem_conv_wgrad(xp, dy, 224, 224)
    torch.cuda.synchronize()
    assert ops.route_counts()["wgrad_stem"] == 1
    acc = torch.zeros(64, 3, 7, 7, dtype=torch.float64)
    w0 = torch.zeros(64, 3, 7, 7)
    for i in range(0, N, 128):
        src = v1 if i < N // 2 else v2
        j = i if i < N // 2 else i - N // 2
        xs = src[j:j + 128].to(DT).float().cpu()
        gs = dy[i:i + 128].float().cpu().permute(0, 3, 1, 2)
        _, gw, _ = torch.ops.aten.convolution_backward(gs, xs, w0, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])
        acc += gw.double()
    _close(dw.cpu(), acc.float(), 2e-3, "stem wgrad")


# ---- the stage-entry shortcuts in their PRODUCTION form: dense 1x1 over the subsampled copy ------------------------------
# (cin, cout, hin): shortcut conv 1x1 / stride 2; route of the dense forward with the BN epilogue, of the dense two-segment data gradient,
# of the dense weight gradient (profiles/r03_layer_table.md: "dense over the subsampled input")
SHORTCUTS = {
    (256, 512, 56): ("gemm1x1_fwd_bnact", "igemm256_dgrad", "wgrad_plain"),
    (512, 1024, 28): ("igemm256_fwd", "igemm256_dgrad", "wgrad_plain"),
    (1024, 2048, 14): ("igemm256_fwd", "igemm256_dgrad", "wgrad_plain"),
}


@pytest.mark.parametrize("cin,cout,h", list(SHORTCUTS), ids=["x".join(map(str, s)) for s in SHORTCUTS])
def test_fullsize_dense_shortcut_forward_dgrad_merge_wgrad(cin, cout, h):
    """What host/resnet_model.py runs for a stage-entry block at 2048 images (dense_shortcut / merge_shortcut): subsample2 + a dense
    stride-1 1x1 with the BatchNorm epilogue; backward: the shortcut's two-segment data gradient [g | x_in] computed densely at the
    OUTPUT resolution, merged into the main branch's conv1 data gradient at the even pixels (sh_dgrad_opts.sub_grad) under the masked
    store; the dense weight gradient with dy's column sums.  Each against ATen on the host, from the ORIGINAL (un-subsampled) tensors."""
    from simhand_amd import ops

    g = torch.Generator(device=DEV).manual_seed(cin + h)
    ho = h // 2
    cw = cin // 2                                                        # conv1 of the stage-entry block: cin -> cin / 2, 1x1 / stride 1
    x = torch.randn(N, h, h, cin, device=DEV, generator=g).relu().to(DT)
    w = (torch.randn(cout, cin, 1, 1, device=DEV, generator=g) / math.sqrt(cin)).to(DT).float()
    st = ops.BNState(cout, DEV)
    st.scale.copy_(torch.rand(cout, device=DEV, generator=g) + 0.5)
    st.shift.copy_(torch.randn(cout, device=DEV, generator=g) * 0.3)
    idx = torch.tensor(SAMPLE, device=DEV)
    # ---- forward: subsample + dense 1x1 + BN epilogue == the stride-2 convolution of x ------------------------------------------------------
    ops.hooks_reset()
    ops.route_reset()
    x_in = ops.subsample2(x)
    assert torch.equal(x_in, x[:, ::2, ::2, :])
    dd = ops.conv_desc(N, ho, ho, cin, cout, 1, 1, 1, 0, DT)
    out = ops.conv2d_fwd_bnact(dd, x_in, ops.pack_krsc(w, DT), st, False, None)
    torch.cuda.synchronize()
    assert _routes(ops) == SHORTCUTS[(cin, cout, h)][0], _routes(ops)
    want = F.conv2d(_nchw(x, idx), w.cpu(), stride=2) * st.scale.cpu().view(1, -1, 1, 1) + st.shift.cpu().view(1, -1, 1, 1)
    _close(_nchw(out, idx), want, 1e-2, "dense shortcut forward")
    # ---- backward: dsub = g (A W) + x_in (-W^T B W) + C W densely, then merged into conv1's masked-store data gradient --------------------------
    gy = torch.randn(N, ho, ho, cout, device=DEV, generator=g).to(DT)
    wa = (torch.randn(cin, cout, device=DEV, generator=g) / math.sqrt(cout)).to(DT)
    wm = (torch.randn(cin, cin, device=DEV, generator=g) / math.sqrt(cin)).to(DT)
    bias = torch.randn(cin, device=DEV, generator=g)
    assert ops.conv2d_dgrad_concat_ok(dd, cin)
    ops.route_reset()
    dsub, _ = ops.conv2d_dgrad_ex(dd, gy, wa, bias=bias, x2=x_in, wt2=wm)
    torch.cuda.synchronize()
    assert _routes(ops) == SHORTCUTS[(cin, cout, h)][1], _routes(ops)
    want_sub = (gy[idx].float().cpu().reshape(-1, cout) @ wa.float().cpu().t() + x_in[idx].float().cpu().reshape(-1, cin) @ wm.float().cpu().t()
                + bias.cpu())
    _close(dsub[idx].float().cpu().reshape(-1, cin), want_sub, 1e-2, "dense shortcut data gradient")
    d1 = ops.conv_desc(N, h, h, cin, cw, 1, 1, 1, 0, DT)
    dy1 = torch.randn(N, h, h, cw, device=DEV, generator=g).to(DT)
    w1 = (torch.randn(cw, cin, 1, 1, device=DEV, generator=g) / math.sqrt(cin)).to(DT).float()
    wt1 = ops.pack_crsk(w1, DT)
    m = N * h * h
    pmask = torch.randint(0, 256, (m, cin // 8), device=DEV, generator=g, dtype=torch.uint8)
    ops.route_reset()
    dx, _ = ops.conv2d_dgrad_ex(d1, dy1, wt1, fuse_mode=4, prev_mask=pmask, want_sums=False, sub_grad=dsub)
    torch.cuda.synchronize()
    merged_route = _routes(ops)
    # the same through the two-launch form (plain masked store, then the scatter-add pass): the merge must not change a bit
    dx2, _ = ops.conv2d_dgrad_ex(d1, dy1, wt1, fuse_mode=4, prev_mask=pmask, want_sums=False)
    ops.scatter2_add(dsub, dx2, pmask)
    assert torch.equal(dx, dx2), f"merged sub_grad ({merged_route}) differs from masked store + scatter-add"
    bits = ((pmask.view(N, h * h, cin // 8)[idx].unsqueeze(-1) >> torch.arange(8, device=DEV, dtype=torch.uint8)) & 1).reshape(len(SAMPLE), h, h, cin).float().cpu()
    ref = (dy1[idx].float().cpu().reshape(-1, cw) @ w1.view(cw, cin).cpu()).view(len(SAMPLE), h, h, cin)
    ref[:, ::2, ::2, :] += dsub[idx].float().cpu()
    _close(dx[idx].float().cpu(), ref * bits, 1e-2, "conv1 data gradient + merged shortcut gradient under the masked store")
    # ---- dense weight gradient (+ dy's column sums) against ATen over all 2048 images ---------------------------------------------------------
    ops.route_reset()
    dw, colsum = ops.conv2d_wgrad_colsum(dd, x_in, gy)
    torch.cuda.synchronize()
    rc = ops.route_counts()
    assert rc[SHORTCUTS[(cin, cout, h)][2]] == 1 and rc["wgrad_colsum"] == 1, rc
    want_w = _cpu_wgrad(x_in, gy, (cout, cin, 1, 1), 1, 0)
    _close(dw.cpu().view(cout, cin, 1, 1), want_w, 2e-3, "dense shortcut weight gradient")
    want_cs = gy.view(-1, cout).float().sum(0, dtype=torch.float64)
    assert (colsum.double() - want_cs).abs().max().item() <= 1e-4 * float(want_cs.abs().max()) + 1e-2


# ---- other per-GPU batches BASELINE names: configs[3] ResNet-152 at 512 pairs = 1024 images, configs[4] simclr at 2048
# pairs = 4096 images.  Tile rounds, split-K and the ragged last round are functions of M: every Appendix-C shape again at those sizes, forward
# and data gradient against ATen on sampled images (the last eight included), the route each launch took pinned by tests/golden/
# fullsize_routes.json (generated on MI355X by scripts/dump_fullsize_routes.py from this very dispatch: a change of a size-dependent
# decision shows up as a diff of that file).
def _routes_golden():
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_routes.json")
    return json.load(open(path)) if os.path.exists(path) else {}


def _operands_n(shape, n, need):
    from simhand_amd import ops

    cin, cout, k, s, h = shape
    pad = 1 if k == 3 else 0
    d = ops.conv_desc(n, h, h, cin, cout, k, k, s, pad, DT)
    g = torch.Generator(device=DEV).manual_seed(5000 + n + sum(shape))
    out = {"d": d, "pad": pad}
    if "x" in need:
        out["x"] = torch.randn(n, h, h, cin, device=DEV, generator=g).to(DT)
    out["w"] = (torch.randn(cout, cin, k, k, device=DEV, generator=g) / math.sqrt(cin * k * k)).to(DT).float()
    if "dy" in need:
        out["dy"] = torch.randn(n, d.ho, d.wo, cout, device=DEV, generator=g).to(DT)
    return out


@pytest.mark.parametrize("n", [1024, 4096])
@pytest.mark.parametrize("shape", list(SHAPES), ids=IDS)
def test_other_batches_forward_dgrad_wgrad(shape, n):
    from simhand_amd import ops

    cin, cout, k, s, h = shape
    sample = [0, 1, n // 2 - 1, n // 2] + list(range(n - 8, n))
    idx = torch.tensor(sample, device=DEV)
    golden = _routes_golden().get(f"{n}:" + "x".join(map(str, shape)))
    o = _operands_n(shape, n, ("x", "dy"))
    d = o["d"]
    ops.hooks_reset()
    ops.route_reset()
    y, _ = ops.conv2d_fwd(d, o["x"], ops.pack_krsc(o["w"], DT), want_stats=True)
    torch.cuda.synchronize()
    r_fwd = _routes(ops)
    _close(_nchw(y, idx), F.conv2d(_nchw(o["x"], idx), o["w"].cpu(), stride=s, padding=o["pad"]), 1e-2, f"fwd at {n} images")
    del y
    ops.route_reset()
    dx = ops.conv2d_dgrad(d, o["dy"], ops.pack_crsk(o["w"], DT))
    torch.cuda.synchronize()
    r_dg = _routes(ops)
    want = torch.nn.grad.conv2d_input((len(sample), cin, h, h), o["w"].cpu(), _nchw(o["dy"], idx).contiguous(), stride=s, padding=o["pad"])
    _close(_nchw(dx, idx), want, 1e-2, f"dgrad at {n} images")
    assert bool(torch.isfinite(dx.float().sum()))
    del dx
    ops.route_reset()
    dw = ops.conv2d_wgrad_oihw(d, o["x"], o["dy"], (cout, cin, k, k))
    torch.cuda.synchronize()
    r_wg = _routes(ops)
    _close(dw.cpu(), _cpu_wgrad(o["x"], o["dy"], (cout, cin, k, k), s, o["pad"]), 2e-3, f"wgrad at {n} images")
    print("ROUTES", n, shape, [r_fwd, r_dg, r_wg])
    assert golden is not None, f"no golden route entry for {n} images {shape}: measured {[r_fwd, r_dg, r_wg]} (scripts/dump_fullsize_routes.py)"
    assert [r_fwd, r_dg, r_wg] == golden, ([r_fwd, r_dg, r_wg], golden)


def test_fullsize_step_backward_operands_captured_against_aten():
    """An INDEPENDENT check of the full-size backward: during a real bf16 ResNet-50 step at 1024 pairs
    (production dispatch) the (x, dy) operands the engine hands to four weight-gradient launches are captured together with the
    gradient the step delivers for that layer; ATen on the host recomputes dW from those operands.  Covers a 3x3 of stage 1, the
    stride-2 3x3 of a stage entry, a conv1 whose dy comes out of the dy-source data gradient, and a 3x3 of stage 4."""
    from oracle import step as orc
    from simhand_amd import ops
    from tests.test_gpu_configs import _product

    b = 1024
    wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
    batch = {k: v.to(DEV) for k, v in orc.synthetic_batch(b, size=224, seed=6).items()}
    torch.manual_seed(6)
    om = orc.StepOracle("simhand_w", "50", ["color_jitter", "crop", "random_crop", "resize", "rotate"], **wcfg).train()
    with torch.no_grad():
        for k, p in om.named_parameters():
            if k.endswith("bn3.weight"):
                p.fill_(0.1)
    ops.hooks_reset()
    model = _product("HandCLR_W", "50", wcfg, om, DT, b)
    names = {"encoder.features.4.1.conv2.weight", "encoder.features.5.0.conv2.weight", "encoder.features.6.2.conv1.weight",
             "encoder.features.7.1.conv2.weight"}
    shapes = {tuple(p.shape): k for k, p in model.named_parameters() if k in names}
    assert len(shapes) == 4                                          # the four layers have four distinct weight shapes
    seen = {}
    real = ops.conv2d_wgrad_oihw

    def spy(d, x, dy, shape):
        dw = real(d, x, dy, shape)
        k = shapes.get(tuple(shape))
        # the LAST launch with that weight shape in backward order is the earliest block of the stage: keep the first one of the named block
        if k is not None:
            seen.setdefault(tuple(shape), []).append((d, x, dy, dw))
        return dw

    ops.conv2d_wgrad_oihw = spy
    try:
        out = model.training_step(batch, 0)
        out["loss"].backward()
        torch.cuda.synchronize()
    finally:
        ops.conv2d_wgrad_oihw = real
    grads = {k: p.grad for k, p in model.named_parameters() if k in names}
    checked = 0
    for shape, k in shapes.items():
        # the launch whose result IS this parameter's gradient (several blocks of a stage share the weight shape)
        hit = [rec for rec in seen[shape] if rec[3].data_ptr() == grads[k].data_ptr() or torch.equal(rec[3], grads[k])]
        assert len(hit) == 1, (k, len(seen[shape]), len(hit))
        d, x, dy, dw = hit[0]
        assert x.shape[0] == 2 * b and bool(torch.isfinite(dw).all())
        want = _cpu_wgrad(x, dy, shape, d.stride, d.pad)
        _close(dw.cpu(), want, 3e-3, f"{k}: dW of the step vs ATen on the step's own operands")
        checked += 1
    assert checked == 4


def test_fullsize_step_backward_data_gradients_captured_against_aten():
    """The DATA path of the full-size backward, checked independently: during a real bf16 ResNet-50 step at
    1024 pairs (production dispatch) four data-gradient launches are captured with the operands the engine handed them, and ATen on the
    host recomputes dx on the 16 sampled images: (A) a stage-1 folded two-segment gradient g (A W) + a2 (-W^T B W) + C W, (B) a stage-entry
    conv1 gradient with the stride-2 shortcut's dense gradient merged at the even pixels (`sub_grad`) and stored through the consumer's
    mask, (C) an identity block's conv1 gradient whose dy operand is DERIVED on load (`dy_src`: dy = A gate(da) - B y + C, also checked)
    with the masked-residual merge, (D) a 3x3 / stride-2 gradient."""
    from oracle import step as orc
    from simhand_amd import ops
    from tests.test_gpu_configs import _product

    b = 1024
    wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
    batch = {k: v.to(DEV) for k, v in orc.synthetic_batch(b, size=224, seed=7).items()}
    torch.manual_seed(7)
    om = orc.StepOracle("simhand_w", "50", ["color_jitter", "crop", "random_crop", "resize", "rotate"], **wcfg).train()
    with torch.no_grad():
        for k, p in om.named_parameters():
            if k.endswith("bn3.weight"):
                p.fill_(0.1)
    ops.hooks_reset()
    model = _product("HandCLR_W", "50", wcfg, om, DT, b)
    idx = torch.tensor(SAMPLE, device=DEV)
    cap = {}
    real_ex, real_fused = ops.conv2d_dgrad_ex, ops.conv2d_dgrad_fused

    def pick(t):
        return None if t is None else t[idx].float().cpu()

    def spy_ex(d, dy, wt, dx=None, accumulate=False, res_grad=None, res_mask=None, bias=None, fuse_mode=None, prev_y=None, prev_st=None,
               prev_mask=None, want_sums=True, x2=None, wt2=None, dy_src=None, fp8=None, sub_grad=None):
        out = real_ex(d, dy, wt, dx=dx, accumulate=accumulate, res_grad=res_grad, res_mask=res_mask, bias=bias, fuse_mode=fuse_mode,
                      prev_y=prev_y, prev_st=prev_st, prev_mask=prev_mask, want_sums=want_sums, x2=x2, wt2=wt2, dy_src=dy_src, fp8=fp8,
                      sub_grad=sub_grad)
        kind = None
        if x2 is not None and d.cin == 64 and fuse_mode == 2 and not accumulate:
            kind = "A"
        elif sub_grad is not None:
            kind = "B"
        elif dy_src is not None and res_grad is not None and sub_grad is None and not accumulate:
            kind = "C"
        if kind is not None and kind not in cap:
            rec = dict(d=d, dx=pick(out[0]), wt=wt.float().cpu(), bias=None if bias is None else bias.float().cpu(), fuse_mode=fuse_mode,
                       dy=pick(dy), x2=pick(x2), wt2=None if wt2 is None else wt2.float().cpu(), res_grad=pick(res_grad),
                       res_mask=None if res_mask is None else res_mask.view(d.n, -1)[idx].cpu(),
                       prev_mask=None if prev_mask is None else prev_mask.view(d.n, -1)[idx].cpu(),
                       sub=None if sub_grad is None else sub_grad[idx].float().cpu())
            if dy_src is not None:
                da, ysrc, st_, coefs, relu_, dy_out = dy_src
                rec.update(da=pick(da), y=pick(ysrc), scale=st_.scale.cpu(), shift=st_.shift.cpu(), coefs=[c.cpu() for c in coefs],
                           relu=bool(relu_), dy_out=pick(dy_out))
            cap[kind] = rec
        return out

    def spy_fused(d, dy, wt, prev_y, prev_st, prev_mask, dx=None, accumulate=False, res_grad=None, res_mask=None):
        out = real_fused(d, dy, wt, prev_y, prev_st, prev_mask, dx=dx, accumulate=accumulate, res_grad=res_grad, res_mask=res_mask)
        if d.stride == 2 and d.r == 3 and "D" not in cap and not accumulate and res_grad is None:
            cap["D"] = dict(d=d, dx=pick(out[0]), dy=pick(dy), wt=wt.float().cpu())
        return out

    real_plain = ops.conv2d_dgrad

    def spy_plain(d, dy, wt, dx=None, accumulate=False):  # the stride-2 3x3 layers take the plain entry (no fused sums at stride 2)
        out = real_plain(d, dy, wt, dx=dx, accumulate=accumulate)
        if d.stride == 2 and d.r == 3 and "D" not in cap and not accumulate:
            cap["D"] = dict(d=d, dx=pick(out), dy=pick(dy), wt=wt.float().cpu())
        return out

    ops.conv2d_dgrad_ex, ops.conv2d_dgrad_fused, ops.conv2d_dgrad = spy_ex, spy_fused, spy_plain
    try:
        out = model.training_step(batch, 0)
        out["loss"].backward()
        torch.cuda.synchronize()
    finally:
        ops.conv2d_dgrad_ex, ops.conv2d_dgrad_fused, ops.conv2d_dgrad = real_ex, real_fused, real_plain
    assert sorted(cap) == ["A", "B", "C", "D"], sorted(cap)

    def unmask(mb, c):
        """uint8 [imgs][pixels * c / 8] -> {0, 1} fp32 [imgs][pixels][c] (bit e of byte j = channel 8 j + e)."""
        m8 = mb.view(mb.shape[0], -1, c // 8).to(torch.int32)
        return torch.stack([(m8 >> e) & 1 for e in range(8)], dim=-1).reshape(mb.shape[0], -1, c).float()

    # (A) two-segment folded gradient, bias in fp32, plain store
    r = cap["A"]
    d = r["d"]
    ns = len(SAMPLE)
    want = r["dy"].reshape(-1, d.cout) @ r["wt"].t() + r["x2"].reshape(-1, d.cin) @ r["wt2"].t() + r["bias"]
    _close(r["dx"].reshape(-1, d.cin), want, 1e-2, "A: stage-1 two-segment folded data gradient of the step")
    # (B) stage-entry conv1: dy . W + shortcut's dense gradient at the even pixels, stored through the consumer's mask
    r = cap["B"]
    d = r["d"]
    dyb = r["dy_out"] if "dy_out" in r else r["dy"]
    want = (dyb.reshape(-1, d.cout) @ r["wt"].t()).view(ns, d.h, d.w, d.cin)
    want[:, ::2, ::2, :] += r["sub"]
    if r["fuse_mode"] == 4:
        want = want * unmask(r["prev_mask"], d.cin).view(ns, d.h, d.w, d.cin)
    _close(r["dx"], want, 1e-2, "B: stage-entry conv1 data gradient with the merged shortcut gradient")
    assert float(r["sub"].abs().max()) > 0 and float(r["dx"].abs().max()) > 0
    # (C) identity block conv1: dy derived on load, masked-residual merge, masked store
    r = cap["C"]
    d = r["d"]
    ca, cb, cc = r["coefs"]
    gate = ((r["y"] * r["scale"] + r["shift"]) > 0).float() if r["relu"] else 1.0
    dy_want = ca * (r["da"] * gate) - cb * r["y"] + cc
    _close(r["dy_out"], dy_want, 1e-2, "C: dy derived on load (BatchNorm-backward apply)")
    want = (r["dy_out"].reshape(-1, d.cout) @ r["wt"].t()).view(ns, -1, d.cin)
    want = want + r["res_grad"].reshape(ns, -1, d.cin) * unmask(r["res_mask"], d.cin)
    if r["fuse_mode"] == 4:
        want = want * unmask(r["prev_mask"], d.cin)
    _close(r["dx"].reshape(ns, -1, d.cin), want, 1e-2, "C: identity-block conv1 data gradient (dy_src + masked residual)")
    # (D) 3x3 / stride 2 (CRSK weights [cin][r][s][cout] -> OIHW)
    r = cap["D"]
    d = r["d"]
    w_oihw = r["wt"].view(d.cin, 3, 3, d.cout).permute(3, 0, 1, 2).contiguous()
    want = torch.nn.grad.conv2d_input((ns, d.cin, d.h, d.w), w_oihw, r["dy"].permute(0, 3, 1, 2).contiguous(), stride=2, padding=1)
    _close(r["dx"].permute(0, 3, 1, 2), want, 1e-2, "D: 3x3 / stride-2 data gradient of the step")


FOLD_SHAPES = [(64, 256, 56), (128, 512, 28), (256, 1024, 14), (512, 2048, 7)]  # (w, 4w, H) of conv3 + bn3 per stage


@pytest.mark.parametrize("cw,cc,h", FOLD_SHAPES)
def test_fullsize_folded_dgrad_two_segments_bias_and_fused_sums(cw, cc, h):
    """The folded bn3 backward's data gradient as the engine issues it at 2048 images: da2 = g (A W) + a2 (-W^T B W) + C W in ONE
    launch over the concatenated reduction [g | a2] (K = 4w + w), fp32 bias, and the BatchNorm-backward sums of the unit below
    (sum g2, sum g2 * y2 with g2 = da2 gated by bn2's recomputed ReLU mask) in its epilogue."""
    from simhand_amd import ops

    g = torch.Generator(device=DEV).manual_seed(cw + h)
    d = ops.conv_desc(N, h, h, cw, cc, 1, 1, 1, 0, DT)
    assert ops.conv2d_dgrad_concat_ok(d, cw)
    gy = torch.randn(N, h, h, cc, device=DEV, generator=g).to(DT)
    a2 = torch.randn(N, h, h, cw, device=DEV, generator=g).relu().to(DT)
    wa = (torch.randn(cw, cc, device=DEV, generator=g) / math.sqrt(cc)).to(DT)
    wm = (torch.randn(cw, cw, device=DEV, generator=g) / math.sqrt(cw)).to(DT)
    bias = torch.randn(cw, device=DEV, generator=g)
    y2 = torch.randn(N, h, h, cw, device=DEV, generator=g).to(DT)
    st = ops.BNState(cw, DEV)
    st.scale.copy_(torch.rand(cw, device=DEV, generator=g) + 0.5)
    st.shift.copy_(torch.randn(cw, device=DEV, generator=g) * 0.3)
    ops.hooks_reset()
    ops.route_reset()
    dx, part = ops.conv2d_dgrad_ex(d, gy, wa, bias=bias, x2=a2, wt2=wm, fuse_mode=2, prev_y=y2, prev_st=st)
    torch.cuda.synchronize()
    rc = ops.route_counts()
    assert rc["dgrad_concat"] == 1 and rc["dgrad_fused_sums"] == 1, rc
    assert rc["n128_dgrad"] == (1 if cw == 128 else 0), rc   # round 4: the 128-channel layer on gemm_n128_kernel
    idx = torch.tensor(SAMPLE, device=DEV)
    want = (gy[idx].float().cpu().reshape(-1, cc) @ wa.float().cpu().t() + a2[idx].float().cpu().reshape(-1, cw) @ wm.float().cpu().t()
            + bias.cpu())
    _close(dx[idx].float().cpu().reshape(-1, cw), want, 1e-2, "two-segment dgrad")
    # fused sums against a direct fp64 reduction of the stored dx over all rows (gate recomputed from y2 * scale + shift > 0)
    m = N * h * h
    s1 = torch.zeros(cw, dtype=torch.float64, device=DEV)
    s2 = torch.zeros(cw, dtype=torch.float64, device=DEV)
    dxv, yv = dx.view(m, cw), y2.view(m, cw)
    for r0 in range(0, m, 1 << 20):
        yy = yv[r0:r0 + (1 << 20)].float()
        gg = dxv[r0:r0 + (1 << 20)].float() * ((yy * st.scale + st.shift) > 0)
        s1 += gg.double().sum(0)
        s2 += (gg * yy).double().sum(0)
    got1, got2 = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
    # the epilogue sums the fp32 accumulators, the check their bf16 roundings: 2^-9 relative per addend, random sign
    tol1 = 1e-3 * float(dxv.float().abs().mean()) * m ** 0.5 * 4 + 1e-3 * float(s1.abs().max())
    assert (got1 - s1).abs().max().item() <= tol1, ((got1 - s1).abs().max().item(), tol1)
    assert (got2 - s2).abs().max().item() <= tol1 * 2 + 1e-3 * float(s2.abs().max())


@pytest.mark.parametrize("cw,cc,h", FOLD_SHAPES)
def test_fullsize_conv1_dgrad_residual_merge_masked_store(cw, cc, h):
    """conv1's data gradient of an identity block as the engine issues it: dx = (dgrad(dy) + dz * bit(own mask)) * bit(mask of the
    block below) -- the masked-residual merge with the masked store (fuse_mode 4), and the dy-source form on top of it."""
    from simhand_amd import ops

    g = torch.Generator(device=DEV).manual_seed(cw + h + 1)
    d = ops.conv_desc(N, h, h, cc, cw, 1, 1, 1, 0, DT)   # conv1: 4w -> w
    m = N * h * h
    dy = torch.randn(N, h, h, cw, device=DEV, generator=g).to(DT)
    w = (torch.randn(cw, cc, 1, 1, device=DEV, generator=g) / math.sqrt(cc)).to(DT).float()
    wt = ops.pack_crsk(w, DT)
    rg = torch.randn(N, h, h, cc, device=DEV, generator=g).to(DT)
    rmask = torch.randint(0, 256, (m, cc // 8), device=DEV, generator=g, dtype=torch.uint8)
    pmask = torch.randint(0, 256, (m, cc // 8), device=DEV, generator=g, dtype=torch.uint8)
    ops.hooks_reset()
    ops.route_reset()
    dx, _ = ops.conv2d_dgrad_ex(d, dy, wt, res_grad=rg, res_mask=rmask, fuse_mode=4, prev_mask=pmask, want_sums=False)
    torch.cuda.synchronize()
    idx = torch.tensor(SAMPLE, device=DEV)

    def bits(mask, rows):
        return ((mask.view(N, h * h, cc // 8)[rows].unsqueeze(-1) >> torch.arange(8, device=DEV, dtype=torch.uint8)) & 1).reshape(-1, cc).float().cpu()

    ref = dy[idx].float().cpu().reshape(-1, cw) @ w.view(cw, cc).cpu()
    want = (ref + rg[idx].float().cpu().reshape(-1, cc) * bits(rmask, idx)) * bits(pmask, idx)
    _close(dx[idx].float().cpu().reshape(-1, cc), want, 1e-2, "masked merge")
    if ops.conv2d_dgrad_dysrc_ok(d):  # dy derived on load: dy = A (da [bn(y) > 0]) - B y + C
        y1 = torch.randn(N, h, h, cw, device=DEV, generator=g).to(DT)
        da = torch.randn(N, h, h, cw, device=DEV, generator=g).to(DT)
        st = ops.BNState(cw, DEV)
        st.scale.copy_(torch.rand(cw, device=DEV, generator=g) + 0.5)
        st.shift.copy_(torch.randn(cw, device=DEV, generator=g) * 0.3)
        coefs = tuple((torch.randn(cw, device=DEV, generator=g) * sc).contiguous() for sc in (1.0, 0.1, 0.05))
        dyo = torch.empty_like(y1)
        ops.route_reset()
        dx2, _ = ops.conv2d_dgrad_ex(d, None, wt, res_grad=rg, res_mask=rmask, fuse_mode=4, prev_mask=pmask, want_sums=False,
                                     dy_src=(da, y1, st, coefs, True, dyo))
        torch.cuda.synchronize()
        assert ops.route_counts()["dgrad_dysrc"] == 1
        yy = y1.float()
        want_dy = coefs[0] * (da.float() * ((yy * st.scale + st.shift) > 0)) - coefs[1] * yy + coefs[2]
        err = (dyo.float() - want_dy).abs().max().item()
        assert err <= 1e-2 * want_dy.abs().max().item(), err
        ref2, _ = ops.conv2d_dgrad_ex(d, dyo, wt, res_grad=rg, res_mask=rmask, fuse_mode=4, prev_mask=pmask, want_sums=False)
        assert torch.equal(dx2, ref2)


@pytest.mark.parametrize("c,h", [(64, 56), (128, 28), (256, 14), (512, 7)])
def test_fullsize_bn_apply_gram_launch(c, h):
    """a = relu(y * scale + shift), a^T a and sum a from ONE launch (bn2's apply riding on the fold's Gram launch) at 2048 images:
    split-K = exactly one round of resident blocks."""
    from simhand_amd import ops

    g = torch.Generator(device=DEV).manual_seed(c)
    y = (torch.randn(N, h, h, c, device=DEV, generator=g) * 1.2 + 0.1).to(DT)
    st = ops.BNState(c, DEV)
    st.scale.copy_(torch.rand(c, device=DEV, generator=g) + 0.5)
    st.shift.copy_(torch.randn(c, device=DEV, generator=g) * 0.3)
    ops.hooks_reset()
    ops.route_reset()
    a, s2, t2 = ops.bn_apply_gram(y, st, True)
    torch.cuda.synchronize()
    assert ops.route_counts()["bn_apply_gram"] == 1
    m = N * h * h
    want_a = ops.bn_apply(y.view(m, c), st, m, c, True)          # the stand-alone pass: same fp32 expression, same rounding
    assert torch.equal(a.view(m, c), want_a)
    ref_a = (y.view(m, c).float() * st.scale + st.shift).clamp_min(0)  # plain torch (mul + add; the kernels use one fma): one bf16 ulp
    assert ((a.view(m, c).float() - ref_a).abs() <= ref_a.abs() * 2.0 ** -7 + 1e-6).all()
    ws2 = torch.zeros(c, c, dtype=torch.float64, device=DEV)
    wt2 = torch.zeros(c, dtype=torch.float64, device=DEV)
    for r0 in range(0, m, 1 << 19):  # fp64 Gram matrix of the bf16 activation, chunked (checker: torch on the device)
        blk = want_a[r0:r0 + (1 << 19)].double()
        ws2 += blk.t() @ blk
        wt2 += blk.sum(0)
    _close(s2.double(), ws2, 1e-4, "a^T a")
    _close(t2.double(), wt2, 1e-4, "sum a")


@pytest.mark.parametrize("c,h,route", [(64, 56, "c64_fwd"), (128, 28, "r128_fwd")])
def test_fullsize_bn_on_load_forward(c, h, route):
    """Round 4: bn1 + ReLU applied inside conv2's LDS ring (simhand_conv2d_fwd_bnin) at 2048 images -- the persistent block ranges / the
    6 728 tiles with their halos, the by-product written exactly once per pixel.  Activation against torch's expression (one bf16 ulp) and
    bit for bit against the stand-alone pass; the convolution on 16 sampled images against ATen CPU; y and the BatchNorm sums bit for bit
    against the plain kernel on that activation."""
    from simhand_amd import ops

    o = _operands((c, c, 3, 1, h), need=("x", "w"))
    d, y_in = o["d"], o["x"]
    wk = ops.pack_krsc(o["w"], DT)
    g = torch.Generator(device=DEV).manual_seed(c + h)
    st = ops.BNState(c, DEV)
    st.scale.copy_(torch.randn(c, device=DEV, generator=g) * 0.8)
    st.shift.copy_(torch.randn(c, device=DEV, generator=g) * 0.3)
    st.scale[1] = 0.0
    st.shift[1] = 0.5   # a pad position read after the activation would show up here
    ops.hooks_reset()
    ops.route_reset()
    a, y, part = ops.conv2d_fwd_bnin(d, y_in, st, wk, want_stats=True)
    torch.cuda.synchronize()
    rc = ops.route_counts()
    assert rc["fwd_bnin"] == 1 and rc[route] == 1, rc
    m = N * h * h
    want_a = ops.bn_apply(y_in.view(m, c), st, m, c, True)
    assert torch.equal(a.view(m, c), want_a)
    ref_a = (y_in.view(m, c).float() * st.scale + st.shift).clamp_min(0)
    assert ((a.view(m, c).float() - ref_a).abs() <= ref_a.abs() * 2.0 ** -7 + 1e-6).all()
    idx = torch.tensor(SAMPLE, device=DEV)
    want = F.conv2d(_nchw(a, idx), o["w"].cpu(), stride=1, padding=1)
    _close(_nchw(y, idx), want, 1e-2, "fwd through the ring rewrite")
    y2, part2 = ops.conv2d_fwd(d, a, wk, want_stats=True)
    assert torch.equal(y, y2) and torch.equal(part, part2)


def test_fullsize_chained_conv1_stage1():
    """conv3 + bn3 + residual + ReLU of a stage-1 block with the next block's conv1 chained on, at 2048 x 56^2."""
    from simhand_amd import ops

    cin, cout, h = 64, 256, 56
    g = torch.Generator(device=DEV).manual_seed(21)
    d = ops.conv_desc(N, h, h, cin, cout, 1, 1, 1, 0, DT)
    assert ops.conv2d_fwd_chain_ok(d)
    x = torch.randn(N, h, h, cin, device=DEV, generator=g).relu().to(DT)
    w = (torch.randn(cout, cin, 1, 1, device=DEV, generator=g) / math.sqrt(cin)).to(DT).float()
    w1 = (torch.randn(cin, cout, 1, 1, device=DEV, generator=g) / math.sqrt(cout)).to(DT).float()
    res = torch.randn(N, h, h, cout, device=DEV, generator=g).to(DT)
    st = ops.BNState(cout, DEV)
    st.scale.copy_(torch.rand(cout, device=DEV, generator=g) + 0.5)
    st.shift.copy_(torch.randn(cout, device=DEV, generator=g) * 0.3)
    wk, w1k = ops.pack_krsc(w, DT), ops.pack_krsc(w1, DT)
    ops.hooks_reset()
    ops.route_reset()
    out, mask, cy, cpart = ops.conv2d_fwd_bnact_chain(d, x, wk, st, res, w1k)
    torch.cuda.synchronize()
    assert ops.route_counts()["fwd_chain"] == 1
    want_out, want_mask = ops.conv2d_fwd_bnact(d, x, wk, st, True, res, want_mask=True)
    assert torch.equal(out, want_out) and torch.equal(mask, want_mask)
    d1 = ops.conv_desc(N, h, h, cout, cin, 1, 1, 1, 0, DT)
    want_cy, want_part = ops.conv2d_fwd(d1, out, w1k, want_stats=True)
    assert torch.equal(cy, want_cy)
    m = N * h * h
    for col in (0, 1):
        a, b = cpart[:, col].double().sum(0), want_part[:, col].double().sum(0)
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item() + 1e-3
    idx = torch.tensor(SAMPLE, device=DEV)
    ref = F.conv2d(_nchw(out, idx), w1.cpu())
    _close(_nchw(cy, idx), ref, 1e-2, "chained conv1")


def test_fullsize_step_bf16_against_fp32_mode_same_weights():
    """End-to-end at the benchmarked batch (ResNet-50 handclr_w, 1024 pairs of 224^2, production dispatch): the bf16 step's loss and
    embeddings against the library's own exact-fp32 mode on the SAME weights and batch.  The fp32 mode is pinned to the oracle at
    1e-3 by the small-size tests; a wrong tile anywhere in the 2048-image launches shows up here as a broken embedding row.
    Weights: seeded init with gamma3 = 0.1 (the regime of tests/test_gpu_configs.py, where bf16 round-off does not amplify).
    Bands from the small-size tests (test_config1_...: loss 2.9e-4, mean z cosine 0.99957 against the fp32 oracle)."""
    from oracle import step as orc
    from simhand_amd import ops
    from tests.test_gpu_configs import _product

    b = 1024
    wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
    batch = {k: v.to(DEV) for k, v in orc.synthetic_batch(b, size=224, seed=5).items()}
    torch.manual_seed(5)
    om = orc.StepOracle("simhand_w", "50", ["color_jitter", "crop", "random_crop", "resize", "rotate"], **wcfg).train()
    with torch.no_grad():
        for k, p in om.named_parameters():
            if k.endswith("bn3.weight"):
                p.fill_(0.1)
    res = {}
    ops.hooks_reset()
    for dtype in (torch.bfloat16, torch.float32):
        model = _product("HandCLR_W", "50", wcfg, om, dtype, b)
        ops.route_reset()
        with torch.no_grad():  # train-mode forward (batch statistics), nothing saved: the fp32 mode fits next to the bf16 run
            loss = float(model.training_step(batch, 0)["loss"])
            z1, z2 = model.get_transformed_projections(batch)
        res[dtype] = (loss, torch.cat((z1, z2)).float(), ops.route_counts())
        del model
        torch.cuda.empty_cache()
    (lb, zb, rb), (lf, zf, _) = res[torch.bfloat16], res[torch.float32]
    cos = F.cosine_similarity(zb.double(), zf.double(), dim=1)
    print({"loss_bf16": lb, "loss_fp32": lf, "z_cos_mean": float(cos.mean()), "z_cos_min": float(cos.min())})
    print("ROUTES", {k: v for k, v in rb.items() if v})
    for r in ("igemm256_fwd", "igemm256_tail", "c64_fwd", "gemm1x1_fwd", "gemm1x1_fwd_bnact", "bn_apply_gram", "stem_fwd"):
        assert rb[r] > 0, r
    assert abs(lb - lf) <= 2e-3 * abs(lf), (lb, lf)
    assert float(cos.mean()) >= 0.999 and float(cos.min()) >= 0.99, (float(cos.mean()), float(cos.min()))
