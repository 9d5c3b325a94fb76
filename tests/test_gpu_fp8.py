"""GPU: the fp8 slice (BASELINE configs[4]: ResNet-50 simclr fp8).  The reference has no fp8 path (fp16 autocast,
src/experiments/main.py:158-159), so there is nothing to be bit-compatible with -- "parity: n/a".  What is checked:
  * the e4m3 quantiser against torch's float8_e4m3fn cast (bit-exact codes) and the scale bookkeeping (current / delayed);
  * the scaled-MFMA convolution against an fp32 convolution of the SAME dequantised operands (the matrix instruction's
    products of e4m3 values are exact in fp32: only the summation order differs);
  * the SimCLR ResNet-50 step with fp8 forward operands against the bf16 step: loss band, embedding cosine, and a
    multi-step run (optimizer in the loop) that stays finite and tracks the bf16 trajectory."""
import pytest
import torch
import torch.nn.functional as F

from oracle import step as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
AUG = ["color_jitter", "crop", "random_crop", "resize", "rotate"]


def _deq(q: torch.Tensor) -> torch.Tensor:
    return q.cpu().view(torch.float8_e4m3fn).float()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_quantiser_matches_torch_float8_cast_and_scale_policy(dtype):
    from simhand_amd import ops

    g = torch.Generator().manual_seed(3)
    x = (torch.randn(4, 33, 16, generator=g) * 3.0).to(dtype)
    x[0, 0, 0] = 70.0  # the amax
    xd = x.to(DEV)
    sc = ops.FP8Scaler(DEV, delayed=False, margin_bits=0)
    q = sc.quantize(xd)
    scale = float(sc.state[0])
    assert abs(scale - 448.0 / 70.0) <= 1e-6 * scale and abs(float(sc.state[1]) * scale - 1.0) < 1e-6
    want = (x.float() * scale).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q.cpu(), want)
    assert float(_deq(q).abs().max()) == 448.0
    # delayed scaling: call 1 calibrates, call 2 still uses call 1's amax and records its own, call 3 sees max(ring)
    sd = ops.FP8Scaler(DEV, delayed=True, margin_bits=1)
    sd.quantize(xd)
    s1 = float(sd.state[0])
    assert abs(s1 - 448.0 / 140.0) <= 1e-6 * s1
    q2 = sd.quantize((xd.float() * 4).to(dtype))  # 4x larger tensor, old scale: saturates instead of overflowing
    assert torch.isfinite(_deq(q2)).all() and float(_deq(q2).abs().max()) == 448.0
    s2 = float(sd.state[0])
    assert abs(s2 - 448.0 / 560.0) <= 1e-5 * s2  # ring max = 280 now
    sd.quantize(xd)
    assert abs(float(sd.state[0]) - s2) <= 1e-6 * s2  # the ring remembers the large tensor


@pytest.mark.parametrize("shape", [(2, 14, 14, 128, 128, 3, 1, 1), (3, 9, 9, 256, 128, 3, 2, 1), (2, 7, 7, 512, 256, 1, 1, 0),
                                   (1, 5, 5, 128, 256, 3, 1, 1), (5, 6, 6, 1024, 128, 1, 1, 0)])
def test_fp8_conv_forward_against_fp32_conv_of_the_dequantised_operands(shape):
    _fp8_conv_case(shape, big=False)


@pytest.mark.parametrize("shape", [(3, 14, 14, 256, 256, 3, 1, 1), (2, 9, 9, 512, 256, 3, 2, 1), (5, 7, 7, 1024, 512, 1, 1, 0), (4, 14, 14, 128, 256, 3, 1, 1),
                                   (32, 14, 14, 256, 256, 3, 1, 1)])  # last: 6272 pixels = 28 tiles of 224 rows
def test_fp8_conv_forward_on_the_256x256_lds_dma_kernel(shape):
    """The e4m3 variant of the 256 x 256 LDS-DMA tile kernel (one K = 128 scaled MFMA per tile pair and k-step), forced at test sizes."""
    _fp8_conv_case(shape, big=True)


def test_fp8_conv_forward_fullsize_256ch_3x3_at_2048_images():
    """The layer class the fp8 configuration exists for, at the benchmarked size and through the default dispatch: 3x3 256 -> 256 @ 14^2,
    2048 images; checked on 16 sampled images against an fp32 convolution of the dequantised operands, BatchNorm sums against the output."""
    from simhand_amd import ops

    n, h, c = 2048, 14, 256
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, h, h, c, device=DEV, generator=g).to(torch.bfloat16)
    wt = torch.randn(c, c, 3, 3, device=DEV, generator=g) * (2.0 / (c * 9)) ** 0.5
    d = ops.conv_desc(n, h, h, c, c, 3, 3, 1, 1, torch.bfloat16)
    sx, sw = ops.FP8Scaler(DEV, delayed=True), ops.FP8Scaler(DEV, delayed=False)
    xq, wq = sx.quantize(x), sw.pack_weights(wt)
    ops.hooks_reset()
    ops.route_reset()
    y, part = ops.conv2d_fwd_fp8(d, xq, wq, sx, sw)
    torch.cuda.synchronize()
    rc = ops.route_counts()
    assert rc["fp8_fwd"] == 1 and rc["igemm256_fwd"] == 1, rc
    idx = [0, 1, 2, 3, 509, 1023, 1024, 1025] + list(range(n - 8, n))
    xdq = _deq(xq[idx]).permute(0, 3, 1, 2) * float(sx.state[1])
    wdq = _deq(wq).view(c, 3, 3, c).permute(0, 3, 1, 2) * float(sw.state[1])
    want = F.conv2d(xdq, wdq, padding=1).permute(0, 2, 3, 1)
    err = (y[idx].float().cpu() - want).abs().max() / want.abs().max()
    assert err <= 1e-2, err
    m = n * h * h
    yf = y.view(m, c).float()
    s1, s2 = part[:, 0].double().sum(0) / m, part[:, 1].double().sum(0) / m
    assert (s1 - yf.double().mean(0)).abs().max().item() <= 2e-3 * float(yf.pow(2).mean().sqrt())
    assert ((s2 - yf.double().pow(2).mean(0)).abs() / yf.double().pow(2).mean(0)).max().item() <= 5e-3


def test_delayed_scale_state_keeps_the_descale_of_the_codes_that_exist():
    """state[0] = the scale the NEXT quantisation uses, state[1] = 1 / (the scale the LAST codes were made with): a consumer launched
    behind the delayed update (every convolution is) must de-scale with the latter (round-5 fix: it used to read the next scale's)."""
    from simhand_amd import ops

    g = torch.Generator().manual_seed(4)
    x = torch.randn(64, 32, generator=g).to(torch.bfloat16).to(DEV)
    sd = ops.FP8Scaler(DEV, delayed=True, margin_bits=1)
    sd.quantize(x)                       # calibration call: update runs BEFORE the pass, codes made with state[0]
    s1 = float(sd.state[0])
    assert abs(float(sd.state[1]) * s1 - 1.0) < 1e-6
    big = (x.float() * 8).to(torch.bfloat16)
    q2 = sd.quantize(big)                # made with s1; the update behind it moves state[0] (ring max is 8x now)
    s2 = float(sd.state[0])
    assert abs(s2 - s1 / 8) <= 1e-5 * s1
    assert abs(float(sd.state[1]) * s1 - 1.0) < 1e-6, (float(sd.state[1]), 1 / s1, 1 / s2)   # NOT 1 / s2
    want_codes = (big.float().cpu() * s1).clamp(-448, 448).to(torch.float8_e4m3fn)
    assert torch.equal(q2.cpu(), want_codes.view(torch.uint8))                      # the codes were made with s1 ...
    back = _deq(q2) * float(sd.state[1])                                            # ... and state[1] de-scales them
    assert torch.allclose(back, want_codes.float() / s1, rtol=1e-6, atol=0)


WGRAD_F8 = [(3, 14, 14, 256, 256), (5, 7, 7, 512, 512), (2, 28, 28, 256, 256), (7, 9, 13, 512, 256), (1, 31, 126, 256, 256), (70, 7, 7, 256, 512)]


@pytest.mark.parametrize("shape", WGRAD_F8)
def test_fp8_weight_gradient_against_fp32_of_the_dequantised_operands(shape):
    """simhand_conv2d_wgrad_fp8: dW = dy^T x over e4m3 codes on the scaled K = 128 MFMA (reduction over pixels, ds_read_b64_tr_b8 operands,
    x rows in an LDS ring, all nine taps per block).  Products of e4m3 values are exact in fp32: against the fp32 weight gradient of the
    SAME dequantised operands only the summation order differs."""
    from simhand_amd import ops

    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, h, w, cin, generator=g).relu().to(torch.bfloat16).to(DEV)
    dy = (torch.randn(n, h, w, cout, generator=g) * 0.03).to(torch.bfloat16).to(DEV)
    d = ops.conv_desc(n, h, w, cin, cout, 3, 3, 1, 1, torch.bfloat16)
    assert ops.conv2d_wgrad_fp8_pays(d)
    sx, sdy = ops.FP8Scaler(DEV, delayed=True), ops.FP8Scaler(DEV, delayed=True)
    xq, dyq = sx.quantize(x), sdy.quantize(dy)
    ops.route_reset()
    dw = ops.conv2d_wgrad_fp8(d, xq, dyq, sx.state, sdy.state)
    torch.cuda.synchronize()
    assert ops.route_counts()["fp8_wgrad"] == 1
    xdq = (_deq(xq) * float(sx.state[1])).permute(0, 3, 1, 2).contiguous()
    dydq = (_deq(dyq) * float(sdy.state[1])).permute(0, 3, 1, 2).contiguous()
    want = torch.nn.grad.conv2d_weight(xdq, (cout, cin, 3, 3), dydq, padding=1)
    err = (dw.cpu() - want).abs().max() / want.abs().max()
    assert err <= 2e-4, err
    # and it is the weight gradient of the bf16 operands up to e4m3's 3-bit significand (sanity of the scales): ~2^-4 relative per operand
    ref = torch.nn.grad.conv2d_weight(x.float().cpu().permute(0, 3, 1, 2).contiguous(), (cout, cin, 3, 3), dy.float().cpu().permute(0, 3, 1, 2).contiguous(), padding=1)
    cos = F.cosine_similarity(dw.cpu().flatten(), ref.flatten(), dim=0).item()
    assert cos > 0.99, cos


@pytest.mark.parametrize("n", [2048, 4096])
def test_fp8_weight_gradient_fullsize(n):
    """At the benchmarked image counts (2048 = configs[1]'s, 4096 = BASELINE configs[4]'s per-GPU batch): 3x3 256 -> 256 @ 14^2; the e4m3
    kernel against the bf16 all-taps kernel on the dequantised operands (both reduce 401 408 / 802 816 pixels in fp32)."""
    from simhand_amd import ops

    h, c = 14, 256
    g = torch.Generator(device=DEV).manual_seed(n)
    x = torch.randn(n, h, h, c, device=DEV, generator=g).relu().to(torch.bfloat16)
    dy = (torch.randn(n, h, h, c, device=DEV, generator=g) * 0.03).to(torch.bfloat16)
    d = ops.conv_desc(n, h, h, c, c, 3, 3, 1, 1, torch.bfloat16)
    sx, sdy = ops.FP8Scaler(DEV, delayed=True), ops.FP8Scaler(DEV, delayed=True)
    xq, dyq = sx.quantize(x), sdy.quantize(dy)
    dw = ops.conv2d_wgrad_fp8(d, xq, dyq, sx.state, sdy.state)
    # e4m3 VALUES are exactly representable in bf16: the bf16 all-taps kernel on the un-scaled code values computes the same products
    xb = xq.view(torch.float8_e4m3fn).to(torch.bfloat16)
    dyb = dyq.view(torch.float8_e4m3fn).to(torch.bfloat16)
    assert torch.equal(xb.float(), xq.view(torch.float8_e4m3fn).float())
    want = ops.conv2d_wgrad_oihw(d, xb, dyb, (c, c, 3, 3)) * (sx.state[1] * sdy.state[1])
    torch.cuda.synchronize()
    err = (dw - want).abs().max() / want.abs().max()
    assert float(err) <= 2e-4, float(err)


def _fp8_conv_case(shape, big):
    from simhand_amd import _lib, ops

    if big:
        _lib.load().simhand_test_igemm256_enable(2)
    n, h, w, cin, cout, k, s, p = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16).to(DEV)
    wt = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(DEV)
    d = ops.conv_desc(n, h, w, cin, cout, k, k, s, p, torch.bfloat16)
    assert ops.conv2d_fwd_fp8_supported(d)
    sx, sw = ops.FP8Scaler(DEV, delayed=True), ops.FP8Scaler(DEV, delayed=False)
    xq, wq = sx.quantize(x), sw.pack_weights(wt)
    ops.route_reset()
    y, part = ops.conv2d_fwd_fp8(d, xq, wq, sx, sw)
    assert ops.route_counts()["fp8_fwd"] == 1 and ops.route_counts()["igemm256_fwd"] == (1 if big else 0)
    xdq = _deq(xq).permute(0, 3, 1, 2) * float(sx.state[1])
    wdq = _deq(wq).view(cout, k, k, cin).permute(0, 3, 1, 2) * float(sw.state[1])
    want = F.conv2d(xdq, wdq, stride=s, padding=p).permute(0, 2, 3, 1)
    err = (y.float().cpu() - want).abs().max() / want.abs().max()
    assert err <= 1e-2, err  # output rounded to bf16
    m = n * d.ho * d.wo
    sums = part.sum(dim=0).cpu()
    flat = want.reshape(m, cout)
    assert (sums[0] - flat.sum(0)).abs().max() <= 1e-3 * flat.abs().sum(0).max()
    assert (sums[1] - (flat * flat).sum(0)).abs().max() <= 1e-3 * (flat * flat).sum(0).max()
    # and against the unquantised bf16 convolution: e4m3 keeps 3 mantissa bits -> a few percent
    ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wt.cpu(), stride=s, padding=p).permute(0, 2, 3, 1)
    rel = (y.float().cpu() - ref).norm() / ref.norm()
    assert rel <= 6e-2, rel


@pytest.mark.parametrize("shape", [(3, 14, 14, 256, 256, 1), (2, 18, 18, 256, 512, 2), (5, 7, 7, 512, 512, 1), (33, 14, 14, 256, 256, 1)])
@pytest.mark.parametrize("fused_sums", [False, True])
def test_fp8_data_gradient_on_the_256x256_kernel(shape, fused_sums):
    """conv2d_dgrad_ex with e4m3 operands (3x3, >= 256 channels; stride 1 and the stride-2 parity classes) against the fp32 data gradient
    of the SAME dequantised operands; with the previous unit's BatchNorm-backward sums fused, the partials against a direct reduction of
    the stored dx."""
    from simhand_amd import _lib, ops

    n, h, w, cin, cout, stride = shape
    _lib.load().simhand_test_igemm256_enable(2)
    g = torch.Generator().manual_seed(sum(shape))
    d = ops.conv_desc(n, h, w, cin, cout, 3, 3, stride, 1, torch.bfloat16)
    assert ops.conv2d_dgrad_fp8_pays(d)
    dy = (torch.randn(n, d.ho, d.wo, cout, generator=g) * 0.3).to(torch.bfloat16).to(DEV)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(DEV)
    sdy, sw = ops.FP8Scaler(DEV, delayed=True), ops.FP8Scaler(DEV, delayed=False)
    dyq, wq = sdy.quantize(dy), ops.fp8_pack_crsk(sw, wt)
    kw = {}
    if fused_sums:
        y_prev = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16).to(DEV)
        st = ops.BNState(cin, DEV)
        st.scale.copy_(torch.rand(cin, generator=g) + 0.5)
        st.shift.copy_(torch.randn(cin, generator=g) * 0.3)
        kw = dict(fuse_mode=2, prev_y=y_prev, prev_st=st)
    ops.route_reset()
    dx, part = ops.conv2d_dgrad_ex(d, dy, ops.pack_crsk(wt, torch.bfloat16), fp8=(dyq, wq, sdy, sw), **kw)
    rc = ops.route_counts()
    assert rc["fp8_dgrad"] == 1 and rc["igemm256_dgrad"] == 1, rc
    dydq = (_deq(dyq) * float(sdy.state[1])).permute(0, 3, 1, 2).contiguous()
    wdq = _deq(wq).view(cin, 3, 3, cout).permute(3, 0, 1, 2).contiguous() * float(sw.state[1])  # back to OIHW
    want = torch.nn.grad.conv2d_input((n, cin, h, w), wdq, dydq, stride=stride, padding=1).permute(0, 2, 3, 1)
    err = (dx.float().cpu() - want).abs().max() / want.abs().max()
    assert err <= 1e-2, err
    if fused_sums:
        gate = (y_prev.float() * st.scale + st.shift) > 0
        gdx = dx.float() * gate
        s1, s2 = gdx.double().sum((0, 1, 2)), (gdx * y_prev.float()).double().sum((0, 1, 2))
        got1, got2 = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
        assert (got1 - s1).abs().max().item() <= 2e-3 * s1.abs().max().item() + 1e-3
        assert (got2 - s2).abs().max().item() <= 2e-3 * s2.abs().max().item() + 1e-3
    else:
        assert part is None


@pytest.mark.parametrize("shape", [(256, 256, 14, 1), (512, 512, 7, 1), (256, 256, 28, 2)])
def test_fp8_forward_and_data_gradient_at_configs4_batch(shape):
    """The e4m3 instantiations of the 256 x 256 kernel at BASELINE configs[4]'s OWN per-GPU size (2048 pairs = 4096 images) through the default
    dispatch -- the size-dependent tile plan (224-row tiles, rounds, tail) differs from the 2048-image one: forward and data gradient of the
    3x3 layers of the fp8 set on 16 sampled images against fp32 convolutions of the dequantised operands."""
    from simhand_amd import ops

    cin, cout, h, stride = shape
    n = 4096
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = torch.randn(n, h, h, cin, device=DEV, generator=g).relu().to(torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) * (2.0 / (cin * 9)) ** 0.5
    d = ops.conv_desc(n, h, h, cin, cout, 3, 3, stride, 1, torch.bfloat16)
    assert ops.conv2d_fwd_fp8_pays(d) and ops.conv2d_dgrad_fp8_pays(d)
    sx, sw, sdy, swt = (ops.FP8Scaler(DEV, delayed=True), ops.FP8Scaler(DEV, delayed=False), ops.FP8Scaler(DEV, delayed=True),
                        ops.FP8Scaler(DEV, delayed=False))
    xq, wq = sx.quantize(x), sw.pack_weights(wt)
    ops.hooks_reset()
    ops.route_reset()
    y, _ = ops.conv2d_fwd_fp8(d, xq, wq, sx, sw)
    torch.cuda.synchronize()
    assert ops.route_counts()["fp8_fwd"] == 1 and ops.route_counts()["igemm256_fwd"] == 1
    idx = [0, 1, 2, 3, 1021, 2047, 2048, 2049] + list(range(n - 8, n))
    xdq = _deq(xq[idx]).permute(0, 3, 1, 2) * float(sx.state[1])
    wdq = _deq(wq).view(cout, 3, 3, cin).permute(0, 3, 1, 2) * float(sw.state[1])
    want = F.conv2d(xdq, wdq, stride=stride, padding=1).permute(0, 2, 3, 1)
    err = (y[idx].float().cpu() - want).abs().max() / want.abs().max()
    assert err <= 1e-2, err
    dy = (torch.randn(n, d.ho, d.wo, cout, device=DEV, generator=g) * 0.05).to(torch.bfloat16)
    dyq, wtq = sdy.quantize(dy), ops.fp8_pack_crsk(swt, wt)
    ops.route_reset()
    dx, _ = ops.conv2d_dgrad_ex(d, dy, ops.pack_crsk(wt, torch.bfloat16), fp8=(dyq, wtq, sdy, swt))
    torch.cuda.synchronize()
    assert ops.route_counts()["fp8_dgrad"] == 1 and ops.route_counts()["igemm256_dgrad"] == 1
    dydq = (_deq(dyq[idx]) * float(sdy.state[1])).permute(0, 3, 1, 2).contiguous()
    wtdq = _deq(wtq).view(cin, 3, 3, cout).permute(3, 0, 1, 2).contiguous() * float(swt.state[1])
    want = torch.nn.grad.conv2d_input((len(idx), cin, h, h), wtdq, dydq, stride=stride, padding=1).permute(0, 2, 3, 1)
    err = (dx[idx].float().cpu() - want).abs().max() / want.abs().max()
    assert err <= 1e-2, err
    assert bool(torch.isfinite(dx.float().sum()))


def test_bn_backward_apply_with_fused_e4m3_emission_equals_the_two_pass_form():
    from simhand_amd import ops

    g = torch.Generator().manual_seed(13)
    c, m = 256, 3 * 14 * 14
    gamma = (torch.rand(c, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(c, generator=g) * 0.2).to(DEV)
    fused, plain = ops.FP8Scaler(DEV, delayed=True), ops.FP8Scaler(DEV, delayed=True)
    for step in range(3):
        y = (torch.randn(m, c, generator=g) * 1.3).to(torch.bfloat16).to(DEV)
        da = (torch.randn(m, c, generator=g) * (0.1 + step)).to(torch.bfloat16).to(DEV)
        part = ops.bn_partial_stats(y, m, c)
        st = ops.bn_finalize(part, m, c, gamma, beta, torch.zeros(c, device=DEV), torch.ones(c, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV))
        dy1, _, dg1, db1, q1 = ops.bn_backward(da, None, y, st, gamma, m, c, True, False, mask_from_y=True, fp8_scaler=fused)
        dy2, _, dg2, db2 = ops.bn_backward(da, None, y, st, gamma, m, c, True, False, mask_from_y=True)
        q2 = plain.quantize(dy2)
        assert torch.equal(dy1, dy2) and torch.equal(q1, q2) and torch.equal(dg1, dg2) and torch.equal(db1, db2), step
        assert torch.equal(fused.state, plain.state), step


def test_bn_apply_with_fused_e4m3_emission_equals_the_two_pass_form():
    """simhand_bn_apply_fp8: a (bf16) and q (e4m3 codes) from one pass over y == simhand_bn_apply followed by simhand_fp8_quantize, bit for
    bit, and the delayed-scaling state evolves identically (same amax enters the ring)."""
    from simhand_amd import ops

    g = torch.Generator().manual_seed(9)
    c, m = 256, 3 * 14 * 14
    st = ops.BNState(c, DEV)
    st.scale.copy_(torch.rand(c, generator=g) + 0.5)
    st.shift.copy_(torch.randn(c, generator=g) * 0.3)
    fused, plain = ops.FP8Scaler(DEV, delayed=True), ops.FP8Scaler(DEV, delayed=True)
    for step in range(4):  # call 0 calibrates (two-pass form in both), calls 1.. run the fused kernel; growing tensors move the ring
        y = (torch.randn(3, 14, 14, c, generator=g) * (1.0 + step)).to(torch.bfloat16).to(DEV)
        a1, q1 = fused.bn_apply_quantize(y, st, True)
        a2 = ops.bn_apply(y.view(m, c), st, m, c, True).view_as(y)
        q2 = plain.quantize(a2)
        assert torch.equal(a1, a2) and torch.equal(q1, q2), step
        assert torch.equal(fused.state, plain.state), step


def test_simclr_rn50_step_fp8_tracks_bf16_and_the_oracle_over_several_steps():
    """BASELINE configs[4] arithmetic (ResNet-50 simclr, e4m3 forward operands on the layers where fp8 pays): the step against the fp32 CPU
    ORACLE (loss, embeddings -- an independent reference, not the library's own bf16 run) and against the bf16 step over several
    optimizer steps.  The 256 x 256 kernel is forced, as the production dispatch chooses it at the benchmarked 2048 images."""
    from simhand_amd import _lib, ops
    from tests.test_gpu_configs import _oracle, _product

    b, img = 8, 224
    om = _oracle("simclr", "50", {}, 31, 0.1)
    cpu_batch = orc.synthetic_batch(b, size=img, seed=31)
    batch = {k: v.to(DEV) for k, v in cpu_batch.items()}
    with torch.no_grad():
        lo = float(om.contrastive_step(cpu_batch))
        z_o = om.last["z"].detach().clone()
    _lib.load().simhand_test_igemm256_enable(2)

    class _T:
        max_epochs, world_size = 100, 1

    runs = {}
    for mode in ("bf16", "fp8"):
        model = _product("SimCLR", "50", {}, om, torch.bfloat16, b)
        model.set_compute_dtype(torch.bfloat16, fp8=(mode == "fp8"))
        model.trainer = _T()
        model.setup("fit")
        (opt,), _ = model.configure_optimizers()
        for grp in opt.param_groups:
            grp["lr"] = 1e-3
        ops.route_reset()
        losses, z0 = [], None
        for i in range(5):
            opt.zero_grad(set_to_none=True)
            out = model.training_step(batch, i)
            out["loss"].backward()
            if i == 0:
                with torch.no_grad():
                    z0 = torch.cat(model.get_transformed_projections(batch)).float().cpu()
                grads_ok = all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
                assert grads_ok
                g0 = {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
            if i == 2:  # a step on the DELAYED path (every site has history): the 3x3 weight gradients of stages 3 and 4, norms kept
                n2 = {k: float(p.grad.detach().double().norm()) for k, p in model.named_parameters()
                      if p.grad is not None and ".conv2.weight" in k and (".6." in k or ".7." in k)}
            opt.step()
            losses.append(float(out["loss"]))
        rc = ops.route_counts()
        runs[mode] = (losses, z0, rc["fp8_fwd"], rc["fp8_dgrad"], g0, rc["fp8_wgrad"], n2)
    lb, zb, nb, _, gb, nwb, n2b = runs["bf16"]
    lf, zf, nf, ndg, gf, nwf, n2f = runs["fp8"]
    assert ndg >= 5 * 9, ndg  # and their data gradients
    # the e4m3 weight gradient is on by default: the seven stride-1 3x3 layers with >= 256 channels, every step (the cosine
    # below is scale-invariant and would not see a wrong x_state / dy_state wiring or the route silently not taken)
    assert nwb == 0 and nwf == 5 * 7, (nwb, nwf)
    assert len(n2f) == 9 and set(n2f) == set(n2b)
    ratios = {k: n2f[k] / n2b[k] for k in n2f}
    print("fp8 / bf16 norm of the stage-3/4 conv2 weight gradients at step 2 (delayed scaling):", {k: round(v, 3) for k, v in ratios.items()})
    assert all(0.9 <= r <= 1.1 for r in ratios.values()), ratios
    # first step's parameter gradients (same weights in both runs) of the fp8 run against the bf16 run's: e4m3 operands in nine 3x3
    # forwards and data gradients, ReLU kinks on top -- a sanity band (a wrong scale or layout gives cosines near 0), not parity
    cosg = sorted(float(F.cosine_similarity(gf[k].flatten().double(), gb[k].flatten().double(), dim=0)) for k in gf if gb[k].abs().max() > 1e-7)
    print("fp8 vs bf16 gradient cosines: median", cosg[len(cosg) // 2], "p10", cosg[len(cosg) // 10])
    assert cosg[len(cosg) // 2] >= 0.9, cosg[len(cosg) // 2]
    assert nb == 0 and nf >= 5 * 9, (nb, nf)  # the nine 3x3 layers with >= 256 channels (stages 3 and 4), every step
    cos_o = F.cosine_similarity(zf.double(), z_o.double(), dim=1)
    print("fp8 vs oracle: loss", lf[0], lo, "z cosine mean / min", float(cos_o.mean()), float(cos_o.min()))
    assert abs(lf[0] - lo) <= 1e-2 * abs(lo), (lf[0], lo)
    assert float(cos_o.mean()) >= 0.99 and float(cos_o.min()) >= 0.95, (float(cos_o.mean()), float(cos_o.min()))
    print("bf16", lb, "fp8", lf)
    assert all(l == l and abs(l) < 1e4 for l in lf)
    assert abs(lf[0] - lb[0]) <= 1e-2 * abs(lb[0]), (lf[0], lb[0])
    cos = F.cosine_similarity(zb.double(), zf.double(), dim=1)
    assert float(cos.mean()) >= 0.99 and float(cos.min()) >= 0.95, (float(cos.mean()), float(cos.min()))
    for a, c in zip(lf, lb):
        assert abs(a - c) <= 3e-2 * abs(c), (lf, lb)
    assert abs(lf[-1] - lf[0]) > 1e-4  # the optimizer moved the fp8 run too


def test_fp8_weight_gradient_descale_is_the_one_of_the_forwards_codes():
    """the e4m3 weight gradient reads the activation codes of ITS forward; their de-scale must be the one they were made with,
    not whatever the site's live scaler holds at backward time.  Forward (grad) -> a second forward of a 3x larger batch under no_grad
    (moves every activation site's ring and scale) -> backward of the first: every parameter gradient must equal, bit for bit, the run
    without the intermediate forward (the kernels are deterministic; the dy sites see the same history in both runs)."""
    from simhand_amd import _lib, ops
    from tests.test_gpu_configs import _oracle, _product

    b, img = 4, 224
    om = _oracle("simclr", "50", {}, 37, 0.1)
    batch = {k: v.to(DEV) for k, v in orc.synthetic_batch(b, size=img, seed=37).items()}
    loud = {k: (v * 3.0 if k.startswith("transformed_image") else v) for k, v in batch.items()}
    _lib.load().simhand_test_igemm256_enable(2)
    try:
        got = {}
        for disturb in (False, True):
            model = _product("SimCLR", "50", {}, om, torch.bfloat16, b)
            model.set_compute_dtype(torch.bfloat16, fp8=True)
            # two warm steps: every delayed site has history, so the measured step takes the fused (delayed) quantisation paths
            for i in range(2):
                model.zero_grad(set_to_none=True)
                model.training_step(batch, i)["loss"].backward()
            model.zero_grad(set_to_none=True)
            ops.route_reset()
            out = model.training_step(batch, 2)
            if disturb:
                with torch.no_grad():
                    model.training_step(loud, 3)
            out["loss"].backward()
            assert ops.route_counts()["fp8_wgrad"] >= 7
            got[disturb] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        bad = [k for k in got[False] if not torch.equal(got[False][k], got[True][k])]
        assert not bad, bad[:5]
    finally:
        ops.hooks_reset()
