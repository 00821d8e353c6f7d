"""GPU parity of the MFMA convolution path (fwd / dgrad / wgrad, BN+LeakyReLU) against a plain
PyTorch fp32 reference of the same op evaluated on the same bf16-rounded operands."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402


def dev():
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).bfloat16().float()


def nhwc(x, ld=None):
    """NCHW fp32 cpu -> NHWC bf16 gpu with pixel pitch ld."""
    n, c, h, w = x.shape
    ld = ld or c
    out = torch.zeros(n, h, w, ld, dtype=torch.bfloat16, device=dev())
    out[..., :c] = x.permute(0, 2, 3, 1).to(dev()).bfloat16()
    return out


CASES = [
    # n, h, w, cin, cout, k, s
    (2, 16, 16, 64, 128, 3, 1),
    (2, 16, 16, 128, 256, 3, 1),
    (1, 20, 20, 512, 1024, 3, 1),
    (2, 16, 16, 32, 64, 3, 2),
    (2, 16, 16, 32, 64, 3, 1),
    (2, 16, 16, 64, 32, 1, 1),
    (3, 13, 13, 128, 64, 1, 1),
    (2, 26, 26, 64, 128, 3, 2),
    (1, 13, 13, 256, 512, 3, 2),
    (2, 9, 11, 128, 256, 3, 1),
    (1, 8, 8, 768, 256, 1, 1),
    (1, 8, 8, 384, 128, 1, 1),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case):
    from object_detectors_amd import ops
    n, h, w, cin, cout, k, s = case
    x = rnd((n, cin, h, w), 1)
    wt = rnd((cout, cin, k, k), 2, (2.0 / (cin * k * k)) ** 0.5)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, stride=s, padding=(k - 1) // 2)
    gy = rnd(tuple(y_ref.shape), 3)
    y_ref.backward(gy)
    shape = ops.conv_shape(n, h, w, cin, cout, k, s)
    wt_d = wt.to(dev())
    wf, wd = ops.pack_weights(shape, wt_d)
    xd = nhwc(x)
    y = torch.empty(n, shape.ho, shape.wo, cout, dtype=torch.bfloat16, device=dev())
    rows = ops.conv_stats_rows(shape)
    cp = ops.cout_pad_of(cout)
    stats = torch.zeros(rows + 64, 2, cp, device=dev())
    ops.conv_fwd(shape, xd, wf, y, stats=stats)
    got = y.float().permute(0, 3, 1, 2).cpu()
    tol = 2e-2 * y_ref.abs().max().item()
    assert (got - y_ref.detach()).abs().max().item() < tol
    # BN statistics are those of the STORED (bf16-rounded) tensor
    s1 = stats[:rows, 0, :cout].sum(0).cpu()
    s2 = stats[:rows, 1, :cout].sum(0).cpu()
    np.testing.assert_allclose(s1, got.sum((0, 2, 3)), rtol=1e-4, atol=1e-4 * got.abs().sum((0, 2, 3)).max().item())
    np.testing.assert_allclose(s2, (got ** 2).sum((0, 2, 3)), rtol=1e-4)
    # dgrad
    gyd = nhwc(gy)
    dx = torch.zeros(n, h, w, cin, dtype=torch.bfloat16, device=dev())
    ops.conv_dgrad(shape, gyd, wd, dx)
    gdx = dx.float().permute(0, 3, 1, 2).cpu()
    assert (gdx - xr.grad).abs().max().item() < 2e-2 * xr.grad.abs().max().item()
    # dgrad with residual add
    res = rnd((n, cin, h, w), 4)
    dx2 = torch.zeros_like(dx)
    res_d = nhwc(res)
    ops.conv_dgrad(shape, gyd, wd, dx2, residual=res_d, residual_ld=cin)
    assert (dx2.float().permute(0, 3, 1, 2).cpu() - (xr.grad + res)).abs().max().item() < 2e-2 * (xr.grad + res).abs().max().item()
    # wgrad: fp32 [cout][k*k][cin]
    dw = torch.zeros(cout, k * k * cin, device=dev())
    ops.conv_wgrad(shape, xd, gyd, dw)
    gdw = dw.view(cout, k, k, cin).permute(0, 3, 1, 2).cpu()
    assert (gdw - wr.grad).abs().max().item() < 1e-2 * wr.grad.abs().max().item()


def test_head_conv_f32_bias_255():
    from object_detectors_amd import ops
    n, h, w, cin, cout = 2, 10, 10, 256, 255
    x = rnd((n, cin, h, w), 5)
    wt = rnd((cout, cin, 1, 1), 6, 0.05)
    bias = rnd((cout,), 7, 0.1)
    y_ref = F.conv2d(x, wt, bias)
    shape = ops.conv_shape(n, h, w, cin, cout, 1, 1, out_ld=256)
    wf, wd = ops.pack_weights(shape, wt.to(dev()), want_dgrad=False)
    y = torch.zeros(n, h, w, 256, dtype=torch.float32, device=dev())
    bias_d, x_d = bias.to(dev()), nhwc(x)
    ops.conv_fwd(shape, x_d, wf, y, bias=bias_d, out_f32=True)
    got = y[..., :255].permute(0, 3, 1, 2).cpu()
    assert (got - y_ref).abs().max().item() < 1e-2 * y_ref.abs().max().item()
    assert (y[..., 255] == 0).all()
    # backward of the head conv: dy bf16 with pitch 256 (pad column zero), dgrad K = 256 (padded)
    gy = rnd((n, cout, h, w), 8)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    F.conv2d(xr, wr, br).backward(gy)
    gyd = nhwc(gy, ld=256)
    shape_b = ops.conv_shape(n, h, w, cin, 256, 1, 1, out_ld=256)   # treat the pad column as a zero channel
    wt_pad = torch.zeros(256, cin, 1, 1)
    wt_pad[:255] = wt
    _, wd = ops.pack_weights(shape_b, wt_pad.to(dev()))
    dx = torch.zeros(n, h, w, cin, dtype=torch.bfloat16, device=dev())
    ops.conv_dgrad(shape_b, gyd, wd, dx)
    assert (dx.float().permute(0, 3, 1, 2).cpu() - xr.grad).abs().max().item() < 2e-2 * xr.grad.abs().max().item()
    dw = torch.zeros(255, cin, device=dev())
    db = torch.zeros(255, device=dev())
    ops.conv_wgrad(shape, x_d, gyd, dw, dbias=db)
    assert (dw.cpu() - wr.grad.view(255, cin)).abs().max().item() < 1e-2 * wr.grad.abs().max().item()
    assert (db.cpu() - br.grad).abs().max().item() < 1e-2 * br.grad.abs().max().item()


@pytest.mark.parametrize("c,pixels,res", [(32, 1000, False), (64, 4096, True), (256, 777, True), (1024, 300, False),
                                          (24, 500, False), (96, 1500, True), (328, 600, False), (2056, 130, True)])      # channel counts that are not 8 * 2^k
def test_bn_lrelu_fwd_bwd(c, pixels, res):
    from object_detectors_amd._lib import check, lib, ptr, stream_ptr
    z = rnd((pixels, c), 11, 2.0) + 0.3
    z = z.bfloat16().float()
    gamma = rnd((c,), 12, 0.3) + 1.0
    beta = rnd((c,), 13, 0.2)
    r = rnd((pixels, c), 14) if res else None
    g = rnd((pixels, c), 15)
    # torch reference (fp32): BN(train) -> LeakyReLU(0.1) -> +res
    zr = z.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    zz = zr.t().reshape(1, c, pixels, 1)
    rm, rv = torch.zeros(c), torch.ones(c)
    y = F.leaky_relu(F.batch_norm(zz, rm, rv, gr, br, True, 0.1, 1e-5), 0.1)
    out_ref = y.reshape(c, pixels).t() + (r if res else 0)
    out_ref.backward(g)
    d = dev()
    zd = z.to(d).bfloat16()
    rows = 3
    part = torch.zeros(rows, 2, c, device=d)
    part[0, 0] = z.sum(0).to(d)
    part[1, 1] = (z ** 2).sum(0).to(d)
    ss = torch.empty(4 * c, device=d)
    rmd, rvd = torch.zeros(c, device=d), torch.ones(c, device=d)
    gam_d, bet_d = gamma.to(d), beta.to(d)   # keep alive: ptr() of a temporary would dangle
    check(lib().mi355det_bn_finalize(ptr(part), rows, c, c, pixels, ptr(gam_d), ptr(bet_d), 1e-5, 0.1, ptr(rmd), ptr(rvd),
                                     ptr(ss), stream_ptr()))
    np.testing.assert_allclose(rmd.cpu(), rm, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(rvd.cpu(), rv, rtol=1e-3, atol=1e-4)
    out = torch.empty(pixels, c, dtype=torch.bfloat16, device=d)
    rd = r.to(d).bfloat16() if res else None
    check(lib().mi355det_bn_act_fwd(ptr(zd), c, ptr(ss), c, pixels, 0.1, ptr(rd), c, ptr(out), c, stream_ptr()))
    assert (out.float().cpu() - out_ref.detach()).abs().max().item() < 2e-2 * out_ref.abs().max().item()
    gd = g.to(d).bfloat16()
    sums = torch.zeros(2 * c, device=d)
    check(lib().mi355det_bn_act_bwd_reduce(ptr(gd), c, None, 0, ptr(zd), c, ptr(ss), c, pixels, 0.1, ptr(sums), stream_ptr()))
    dz = torch.empty(pixels, c, dtype=torch.bfloat16, device=d)
    dg, db = torch.zeros(c, device=d), torch.zeros(c, device=d)
    check(lib().mi355det_bn_act_bwd_apply(ptr(gd), c, None, 0, ptr(zd), c, ptr(ss), ptr(sums), None, c, pixels, 0.1, ptr(dz), c, ptr(dg),
                                          ptr(db), stream_ptr()))
    assert (dz.float().cpu() - zr.grad).abs().max().item() < 2e-2 * zr.grad.abs().max().item()
    np.testing.assert_allclose(dg.cpu(), gr.grad, rtol=2e-2, atol=2e-2 * gr.grad.abs().max().item())
    np.testing.assert_allclose(db.cpu(), br.grad, rtol=2e-2, atol=2e-2 * br.grad.abs().max().item())


@pytest.mark.parametrize("c,pixels", [(8, 70), (32, 4097), (64, 20000), (72, 3333), (256, 9000), (1024, 12800), (512, 200000)])
def test_bn_bwd_reduce_fixed_order_form(c, pixels):
    """mi355det_bn_act_bwd_reduce_det (partial rows + a fold launch, what the engines use): equal to an fp64 evaluation to fp32 accuracy,
    close to the atomic form, BIT-identical from launch to launch with one (uninitialised) workspace reused, and it WRITES `sums` (no
    zeroing).  Channel counts: powers of two and not (72: a partly empty last slab); pixel counts from one workgroup to the cap."""
    from object_detectors_amd._lib import check, lib, ptr, stream_ptr
    d = dev()
    L = lib()
    torch.manual_seed(c + pixels)
    z = torch.randn(pixels, c, device=d).bfloat16()
    g = (torch.randn(pixels, c, device=d) * 0.1).bfloat16()
    g2 = (torch.randn(pixels, c, device=d) * 0.1).bfloat16()
    scale, shift = torch.rand(c, device=d) + 0.5, torch.randn(c, device=d) * 0.3
    mean, invstd = torch.randn(c, device=d) * 0.1, torch.rand(c, device=d) + 0.5
    ss = torch.cat([scale, shift, mean, invstd]).contiguous()
    nbytes = L.mi355det_bn_act_bwd_reduce_workspace(c, pixels)
    assert nbytes >= 64
    ws = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device=d)      # NaN bit patterns: every row the fold reads must have been written
    for second in (None, g2):
        gg = g.double() + (second.double() if second is not None else 0)
        if second is not None:
            gg = (g.float() + second.float()).double()
        zz = z.double()
        y = zz * scale.double() + shift.double()
        dy = torch.where(y > 0, gg, gg * 0.1)
        want = torch.cat([dy.sum(0), (dy * ((zz - mean.double()) * invstd.double())).sum(0)])
        outs = []
        for rep in range(3):
            sums = torch.full((2 * c,), float("nan"), device=d)          # written, not accumulated
            check(L.mi355det_bn_act_bwd_reduce_det(ptr(g), c, ptr(second) if second is not None else None, c if second is not None else 0, ptr(z), c,
                                                    ptr(ss), c, pixels, 0.1, ptr(sums), ptr(ws), nbytes, stream_ptr()), "reduce_det")
            outs.append(sums.clone())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
        tol = 2e-5 * float((dy.abs().sum(0)).max()) + 1e-6
        assert float((outs[0].double() - want).abs().max()) < tol
        atom = torch.zeros(2 * c, device=d)
        check(L.mi355det_bn_act_bwd_reduce(ptr(g), c, ptr(second) if second is not None else None, c if second is not None else 0, ptr(z), c,
                                            ptr(ss), c, pixels, 0.1, ptr(atom), stream_ptr()), "reduce")
        assert float((atom.double() - want).abs().max()) < tol
    # too small / missing workspace: EINVAL, nothing launched
    sums = torch.zeros(2 * c, device=d)
    with pytest.raises(ValueError):
        check(L.mi355det_bn_act_bwd_reduce_det(ptr(g), c, None, 0, ptr(z), c, ptr(ss), c, pixels, 0.1, ptr(sums), ptr(ws), 32, stream_ptr()), "reduce_det")


def test_upsample_and_layout():
    from object_detectors_amd._lib import check, lib, ptr, stream_ptr
    d = dev()
    x = rnd((2, 16, 5, 7), 21)
    xd = nhwc(x)
    out = torch.zeros(2, 10, 14, 24, dtype=torch.bfloat16, device=d)   # write into channels [0,16) of a 24-wide buffer
    check(lib().mi355det_upsample2x_fwd(ptr(xd), 16, 2, 5, 7, 16, ptr(out), 24, stream_ptr()))
    ref = F.interpolate(x, scale_factor=2, mode="nearest")
    assert torch.equal(out[..., :16].float().permute(0, 3, 1, 2).cpu(), ref)
    assert (out[..., 16:] == 0).all()
    g = rnd((2, 16, 10, 14), 22)
    gd = nhwc(g)
    gx = torch.zeros(2, 5, 7, 16, dtype=torch.bfloat16, device=d)
    check(lib().mi355det_upsample2x_bwd(ptr(gd), 16, 2, 5, 7, 16, ptr(gx), 16, stream_ptr()))
    xr = x.clone().requires_grad_(True)
    F.interpolate(xr, scale_factor=2, mode="nearest").backward(g)
    assert (gx.float().permute(0, 3, 1, 2).cpu() - xr.grad).abs().max().item() < 3e-2 * xr.grad.abs().max().item()
    # stem im2col == unfold
    img = rnd((2, 3, 9, 8), 23)
    col = torch.zeros(2 * 9 * 8, 32, dtype=torch.bfloat16, device=d)
    img_d = img.to(d)
    check(lib().mi355det_stem_im2col(ptr(img_d), ptr(col), 2, 9, 8, stream_ptr()))
    unf = F.unfold(img, 3, padding=1).view(2, 3, 9, 72).permute(0, 3, 2, 1).reshape(2 * 72, 27)   # k = tap*3 + c
    assert torch.equal(col[:, :27].float().cpu(), unf)
    assert (col[:, 27:] == 0).all()
    # rows wider than one 128-pixel segment, last segment partial
    img = rnd((1, 3, 5, 300), 25).bfloat16().float()
    col = torch.full((5 * 300 + 8, 32), 9.0, dtype=torch.bfloat16, device=d)
    img_d = img.to(d)
    check(lib().mi355det_stem_im2col(ptr(img_d), ptr(col), 1, 5, 300, stream_ptr()))
    unf = F.unfold(img, 3, padding=1).view(1, 3, 9, 1500).permute(0, 3, 2, 1).reshape(1500, 27)
    assert torch.equal(col[:1500, :27].float().cpu(), unf)
    assert (col[:1500, 27:] == 0).all() and (col[1500:] == 9.0).all()
    # layout converters
    t = rnd((2, 5, 4, 6), 24)
    o = torch.zeros(2, 4, 6, 8, dtype=torch.bfloat16, device=d)
    t_d = t.to(d)
    check(lib().mi355det_nchw_f32_to_nhwc(ptr(t_d), 2, 5, 4, 6, ptr(o), 1, 8, stream_ptr()))
    assert torch.equal(o[..., :5].float().permute(0, 3, 1, 2).cpu(), t)
    back = torch.empty(2, 5, 4, 6, device=d)
    check(lib().mi355det_nhwc_to_nchw_f32(ptr(o), 1, 8, 2, 5, 4, 6, ptr(back), stream_ptr()))
    assert torch.equal(back.cpu(), t)


def test_batched_pack_equals_single_pack():
    """mi355det_pack_weights_batched (one launch, tiled transpose for the dgrad packs) == mi355det_pack_weights per layer."""
    import ctypes as C
    from object_detectors_amd import _lib, ops
    from object_detectors_amd._lib import check, lib
    cases = [(64, 128, 3, 1), (32, 64, 3, 2), (128, 64, 1, 1), (256, 512, 1, 2), (64, 32, 1, 1), (96, 256, 3, 1), (512, 1024, 3, 2)]
    L = lib()
    items = (_lib.PackItem * len(cases))()
    keep, want = [], []
    for i, (cin, cout, k, s) in enumerate(cases):
        shp = ops.conv_shape(1, 8, 8, cin, cout, k, s)
        w = (torch.randn((cout, k, k, cin), generator=torch.Generator().manual_seed(i)) * 0.1).to(dev())
        wf0, wd0 = ops.pack_weights(shp, w, ohwi=True)
        want.append((wf0, wd0))
        wf, wd = torch.full_like(wf0, 7.0), torch.full_like(wd0, 7.0)
        keep += [shp, w, wf, wd]
        items[i].w, items[i].w_fwd, items[i].w_dgrad = w.data_ptr(), wf.data_ptr(), wd.data_ptr()
        items[i].shape, items[i].cout_pad, items[i].w_is_ohwi = shp, ops.cout_pad_of(cout), 1
    ne, nb = C.c_int32(0), C.c_int32(0)
    nbytes = L.mi355det_pack_table_bytes(items, len(cases), C.byref(ne), C.byref(nb))
    host = torch.empty(nbytes, dtype=torch.uint8)
    check(L.mi355det_pack_table_build(items, len(cases), C.c_void_p(host.data_ptr()), nbytes), "pack_table_build")
    tab = host.to(dev())
    check(L.mi355det_pack_weights_batched(C.c_void_p(tab.data_ptr()), ne.value, nb.value, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pack")
    torch.cuda.synchronize()
    for i, (wf0, wd0) in enumerate(want):
        wf, wd = keep[4 * i + 2], keep[4 * i + 3]
        assert torch.equal(wf, wf0), cases[i]
        n = lib().mi355det_dgrad_pack_elems(C.byref(keep[4 * i])) - 64          # the 64 trailing elements are slack
        assert torch.equal(wd[:n], wd0[:n]), cases[i]


def test_partial_stats_rows_identical_across_tile_configs():
    """Every tile configuration must fill the same partial-statistics rows (one per 128 lattice pixels): the plan autotuner runs
    all of them on one buffer, and bn_finalize sums a fixed number of rows."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import lib
    n, h, w, cin, cout = 3, 24, 24, 128, 256           # M = 1728 = 13.5 rows of 128: the last 256-pixel tile is half empty
    x = rnd((n, cin, h, w), 1)
    wt = rnd((cout, cin, 3, 3), 2, (2.0 / (cin * 9)) ** 0.5)
    shape = ops.conv_shape(n, h, w, cin, cout, 3, 1)
    wf, _ = ops.pack_weights(shape, wt.to(dev()), want_dgrad=False)
    xd = nhwc(x)
    rows = ops.conv_stats_rows(shape)
    assert rows == (n * h * w + 127) // 128
    stats = torch.full((rows + 64, 2, ops.cout_pad_of(cout)), 123.0, device=dev())
    y = torch.zeros((n, h, w, cout), device=dev(), dtype=torch.bfloat16)
    try:
        for cfg in (1, 2, 3, 4, 5, 6, 3, 1, 6):
            lib().mi355det_debug_set(0, cfg)
            ops.conv_fwd(shape, xd, wf, y, stats=stats)
            torch.cuda.synchronize()
            yf = y.float().reshape(-1, cout)
            s1, s2 = stats[:rows, 0, :cout].sum(0), stats[:rows, 1, :cout].sum(0)
            torch.testing.assert_close(s1, yf.sum(0), rtol=1e-3, atol=1e-2, msg=f"cfg {cfg} sum")
            torch.testing.assert_close(s2, (yf * yf).sum(0), rtol=1e-3, atol=1e-2, msg=f"cfg {cfg} sumsq")
    finally:
        lib().mi355det_debug_set(0, 0)


@pytest.mark.parametrize("case", [(2, 24, 24, 128, 256, 3, 1), (2, 16, 16, 64, 128, 3, 2), (3, 13, 13, 256, 128, 1, 1), (2, 26, 26, 32, 64, 3, 2),
                                  (1, 40, 40, 256, 512, 3, 1)])
@pytest.mark.parametrize("with_res", [False, True])
def test_dgrad_with_fused_bn_backward_sums(case, with_res):
    """mi355det_conv_dgrad_bn: dx identical to the plain dgrad, and its per-channel sums equal bn_act_bwd_reduce on that dx."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import check, lib, ptr, stream_ptr
    n, h, w, cin, cout, k, s = case
    shape = ops.conv_shape(n, h, w, cin, cout, k, s)
    wt = rnd((cout, cin, k, k), 2, (2.0 / (cin * k * k)) ** 0.5)
    _, wd = ops.pack_weights(shape, wt.to(dev()))
    dy = nhwc(rnd((n, cout, shape.ho, shape.wo), 3))
    z = nhwc(rnd((n, cin, h, w), 4))                      # pre-BN output of the layer that produced this conv's input
    res = nhwc(rnd((n, cin, h, w), 5)) if with_res else None
    g = torch.Generator().manual_seed(6)
    ss = torch.cat([torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3, torch.randn(cin, generator=g) * 0.2,
                    torch.rand(cin, generator=g) + 0.5]).to(dev())
    dx0 = torch.zeros((n, h, w, cin), device=dev(), dtype=torch.bfloat16)
    ops.conv_dgrad(shape, dy, wd, dx0, residual=res, residual_ld=cin)
    want = torch.zeros(2 * cin, device=dev())
    check(lib().mi355det_bn_act_bwd_reduce(ptr(dx0), cin, None, 0, ptr(z), cin, ptr(ss), cin, n * h * w, 0.1, ptr(want), stream_ptr()), "reduce")
    for cfg in (0, 1, 3, 6):
        lib().mi355det_debug_set(0, cfg)
        try:
            dx = torch.full_like(dx0, 9.0)
            sums = ops.conv_dgrad_bn(shape, dy, wd, dx, z, ss, 0.1, residual=res, residual_ld=cin)
            torch.cuda.synchronize()
        finally:
            lib().mi355det_debug_set(0, 0)
        assert torch.equal(dx, dx0), cfg
        scale = float(want.abs().max())
        assert float((sums - want).abs().max()) <= 2e-3 * scale + 1e-3, (cfg, float((sums - want).abs().max()), scale)


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 128), (2, 9, 11, 128, 256), (3, 13, 13, 64, 128), (1, 20, 20, 512, 1024), (2, 40, 40, 256, 512),
                                  (1, 5, 3, 192, 128), (2, 1, 1, 256, 256), (2, 2, 2, 256, 128), (16, 1, 1, 256, 256), (3, 2, 1, 64, 128),
                                  (5, 1, 2, 128, 128)])
def test_dx_reuse_kernel_matches_default(case):
    """Tile configuration 15 (3x3 stride-1 kernel with shared pixel tiles) against PyTorch fp32 and the default configuration:
    forward (+ BN partial statistics) and data gradient (+ residual), image edges and tile tails included."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import lib
    n, h, w, cin, cout = case
    x = rnd((n, cin, h, w), 1)
    wt = rnd((cout, cin, 3, 3), 2, (2.0 / (cin * 9)) ** 0.5)
    xr, wr = x.clone().requires_grad_(True), wt.clone()
    y_ref = F.conv2d(xr, wr, padding=1)
    gy = rnd(tuple(y_ref.shape), 3)
    y_ref.backward(gy)
    shape = ops.conv_shape(n, h, w, cin, cout, 3, 1)
    wf, wd = ops.pack_weights(shape, wt.to(dev()))
    xd, gyd = nhwc(x), nhwc(gy)
    res = nhwc(rnd((n, cin, h, w), 4))
    rows = ops.conv_stats_rows(shape)
    outs = {}
    try:
        for cfg in (1, 15, 16, 17, 18, 19, 26, 27, 28, 35):
            lib().mi355det_debug_set(0, cfg)
            y = torch.full((n, h, w, cout), 5.0, device=dev(), dtype=torch.bfloat16)
            stats = torch.zeros((rows + 64, 2, ops.cout_pad_of(cout)), device=dev())
            ops.conv_fwd(shape, xd, wf, y, stats=stats)
            dx = torch.full((n, h, w, cin), 5.0, device=dev(), dtype=torch.bfloat16)
            ops.conv_dgrad(shape, gyd, wd, dx)
            dxr = torch.full((n, h, w, cin), 5.0, device=dev(), dtype=torch.bfloat16)
            ops.conv_dgrad(shape, gyd, wd, dxr, residual=res, residual_ld=cin)
            torch.cuda.synchronize()
            outs[cfg] = (y.float().cpu(), stats[:rows].sum(0).cpu(), dx.float().cpu(), dxr.float().cpu())
    finally:
        lib().mi355det_debug_set(0, 0)
    yr = y_ref.detach().permute(0, 2, 3, 1)
    gr = xr.grad.permute(0, 2, 3, 1)
    y1, st1, dx1, dxr1 = outs[1]
    for cfg in (15, 16, 17, 18, 19, 26, 27, 28, 35):          # 17 / 18 / 28 (256-wide) fall back to the default tile when cout % 256 != 0
        y15, st15, dx15, dxr15 = outs[cfg]
        assert float((y15 - yr).abs().max()) <= 2e-2 * float(yr.abs().max()), cfg
        assert float((dx15 - gr).abs().max()) <= 2e-2 * float(gr.abs().max()), cfg
        # against the default tile configuration: same products, different summation order only
        assert float((y15 - y1).abs().max()) <= 1e-2 * float(y1.abs().max()), cfg
        assert float((dx15 - dx1).abs().max()) <= 1e-2 * float(dx1.abs().max()), cfg
        assert float((dxr15 - dxr1).abs().max()) <= 1e-2 * float(dxr1.abs().max()), cfg
        torch.testing.assert_close(st15, st1, rtol=2e-2, atol=2e-2 * float(st1.abs().max()))


@pytest.mark.parametrize("case", [(2, 24, 24, 64, 64), (2, 16, 20, 64, 32), (2, 12, 12, 32, 64), (1, 9, 7, 64, 64)])
def test_narrow_output_shared_pixel_tile_kernels(case):
    """256x64 / 256x32 shared-pixel-tile kernels (ids 30 / 29: the first Darknet layers and their data gradients; 31 = the 256x64 form with a
    32-deep k-step for 32 input channels) against the plain narrow tiles: forward + statistics where cout is 32/64, data gradient where
    cin is 32/64."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import lib
    n, h, w, cin, cout = case
    x = rnd((n, cin, h, w), 1)
    wt = rnd((cout, cin, 3, 3), 2, (2.0 / (cin * 9)) ** 0.5)
    gy = rnd((n, cout, h, w), 3)
    shape = ops.conv_shape(n, h, w, cin, cout, 3, 1)
    wf, wd = ops.pack_weights(shape, wt.to(dev()))
    xd, gyd = nhwc(x), nhwc(gy)
    rows = ops.conv_stats_rows(shape)
    outs = {}
    try:
        for cfg in (0, 29, 30, 31):
            lib().mi355det_debug_set(0, cfg)
            y = torch.full((n, h, w, cout), 5.0, device=dev(), dtype=torch.bfloat16)
            stats = torch.zeros((rows + 64, 2, ops.cout_pad_of(cout)), device=dev())
            dx = torch.full((n, h, w, cin), 5.0, device=dev(), dtype=torch.bfloat16)
            if cin % 64 == 0 or (cin == 32 and cout == 64):      # 31 applies to the 32 -> 64 forward only; elsewhere it falls back to the plain tile
                ops.conv_fwd(shape, xd, wf, y, stats=stats)
            if cout % 64 == 0:              # the dgrad's reduction dimension
                ops.conv_dgrad(shape, gyd, wd, dx)
            torch.cuda.synchronize()
            outs[cfg] = (y.float().cpu(), stats[:rows].sum(0).cpu(), dx.float().cpu())
    finally:
        lib().mi355det_debug_set(0, 0)
    y0, s0, d0 = outs[0]
    for cfg in (29, 30, 31):
        y1, s1, d1 = outs[cfg]
        assert float((y1 - y0).abs().max()) <= 1e-2 * float(y0.abs().max()) + 1e-6, cfg
        assert float((d1 - d0).abs().max()) <= 1e-2 * float(d0.abs().max()) + 1e-6, cfg
        torch.testing.assert_close(s1, s0, rtol=2e-2, atol=2e-2 * float(s0.abs().max()) + 1e-6)
    ref = F.conv2d(x, wt, padding=1).permute(0, 2, 3, 1)
    if cin % 64 == 0 or (cin == 32 and cout == 64):
        assert float((y0 - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
        assert float((outs[31][0] - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("case", [(3, 20, 12, 64, 128, 3, 1, 64, 128), (2, 24, 16, 32, 64, 3, 2, 48, 64), (2, 16, 16, 128, 72, 1, 1, 128, 80),
                                  (5, 40, 40, 128, 256, 3, 1, 128, 256), (1, 4, 4, 256, 128, 3, 1, 256, 128),
                                  (3, 13, 13, 64, 128, 3, 1, 64, 128), (2, 25, 25, 64, 64, 3, 2, 64, 64), (3, 7, 7, 128, 128, 3, 1, 128, 128),
                                  (2, 50, 50, 64, 128, 1, 1, 64, 128), (1, 9, 5, 32, 32, 3, 1, 40, 32),
                                  (2, 32, 32, 64, 128, 3, 1, 64, 136), (2, 64, 64, 32, 64, 3, 2, 32, 64), (1, 7, 48, 128, 128, 3, 1, 128, 128)])
def test_wgrad_scalar_bookkeeping_form_is_bit_identical(case):
    """Maps at least 4 wide take the wgrad form with wave-uniform pixel bookkeeping (buffer-descriptor LDS-DMA; pieces that
    straddle a row end when the width is not a multiple of 4);
    it must add exactly the same products in the same order as the general per-lane form, for every split count."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import lib
    n, h, w, cin, cout, k, s, ldx, lddy = case
    shape = ops.conv_shape(n, h, w, cin, cout, k, s, in_ld=ldx, out_ld=lddy)
    x = nhwc(rnd((n, cin, h, w), 11), ldx)
    gy = nhwc(rnd((n, cout, shape.ho, shape.wo), 12), lddy)
    x[..., cin:] = 7.0      # pitch padding must never be read as data
    gy[..., cout:] = 7.0
    outs = []
    try:
        for general in (1, 0):
            lib().mi355det_debug_set(1, general)
            dw = torch.zeros(cout, k * k * cin, device=dev())
            db = torch.zeros(cout, device=dev())
            ops.conv_wgrad(shape, x, gy, dw, dbias=db)
            outs.append((dw.cpu(), db.cpu()))
    finally:
        lib().mi355det_debug_set(1, 0)
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-4)
    wz = torch.zeros(cout, cin, k, k, requires_grad=True)
    y = torch.nn.functional.conv2d(x[..., :cin].float().permute(0, 3, 1, 2).cpu(), wz, stride=s, padding=(k - 1) // 2)
    y.backward(gy[..., :cout].float().permute(0, 3, 1, 2).cpu())
    got = outs[1][0].view(cout, k, k, cin).permute(0, 3, 1, 2)
    assert (got - wz.grad).abs().max().item() < 1e-2 * wz.grad.abs().max().item()


@pytest.mark.parametrize("case", [(2, 40, 40, 128, 256, 3, 1, 128, 256, (1, 3, 10)),      # NP = 1152: the fifth 256-wide n' tile is half empty
                                  (3, 20, 20, 256, 512, 3, 1, 256, 512, (1, 2, 5)),         # map width 20: the two 4-pixel pieces of a wave lie in different rows
                                  (2, 24, 16, 64, 256, 3, 2, 72, 264, (1, 2)),              # stride 2, Cin = 64 (four taps per 256 n' columns), pitch padding
                                  (3, 20, 12, 512, 256, 1, 1, 512, 256, (1, 4)),            # 1x1; M = 720: the last k-step has 16 live pixels
                                  (1, 12, 12, 256, 256, 3, 1, 256, 256, (1,)),              # one workgroup per tile walks all the pixels (direct += epilogue)
                                  (2, 16, 32, 32, 256, 3, 1, 32, 256, (1, 2)),              # Cin = 32: NP = 288, taps change inside a 64-column half
                                  (2, 16, 16, 64, 324, 3, 1, 64, 336, (1, 2)),              # partial last co tile, Cout % 8 = 4 (the 10 836-channel cls_logits)
                                  (2, 50, 50, 64, 256, 3, 1, 64, 256, (1, 3, 7)),           # map width 50: pieces cross row ends (the per-lane select form)
                                  (3, 13, 13, 128, 256, 3, 1, 128, 256, (1, 2)),            # 507 pixels: not a multiple of 4, the last piece is partial
                                  (2, 26, 26, 64, 256, 3, 2, 64, 256, (1, 2)),              # stride 2 onto a 13 x 13 map
                                  (1, 25, 25, 256, 512, 1, 1, 256, 512, (1, 2))])           # 1x1 on a 25 x 25 map
def test_wgrad_phase_staggered_kernel_is_bit_identical(case):
    """wgrad8_kernel (256 x 256 x 64, the igemm8 schedule on transposed operands) adds the same products in the same order as the 128 x 128
    kernel for the same split count: bit-identical dW for every split count, and equal to autograd within the storage rounding."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import lib
    n, h, w, cin, cout, k, s, ldx, lddy, split_counts = case
    shape = ops.conv_shape(n, h, w, cin, cout, k, s, in_ld=ldx, out_ld=lddy)
    x = nhwc(rnd((n, cin, h, w), 31), ldx)
    gy = nhwc(rnd((n, cout, shape.ho, shape.wo), 32), lddy)
    x[..., cin:] = 7.0
    gy[..., cout:] = 7.0
    ref = None
    try:
        for sp in split_counts:
            outs = []
            for form in (0, 65536):
                lib().mi355det_debug_set(7, sp + form)
                dw = torch.full((cout, k * k * cin), 0.25, device=dev())      # the kernel ADDS into dW
                ops.conv_wgrad(shape, x, gy, dw)
                outs.append(dw.cpu())
            assert torch.equal(outs[0], outs[1]), sp
            ref = outs[1]
    finally:
        lib().mi355det_debug_set(7, 0)
    wz = torch.zeros(cout, cin, k, k, requires_grad=True)
    y = torch.nn.functional.conv2d(x[..., :cin].float().permute(0, 3, 1, 2).cpu(), wz, stride=s, padding=(k - 1) // 2)
    y.backward(gy[..., :cout].float().permute(0, 3, 1, 2).cpu())
    got = (ref - 0.25).view(cout, k, k, cin).permute(0, 3, 1, 2)
    assert (got - wz.grad).abs().max().item() < 1e-2 * wz.grad.abs().max().item()


@pytest.mark.parametrize("case", [(2, 8, 128, 32, 64, 32, 64, False), (1, 6, 64, 32, 64, 40, 64, True), (2, 4, 64, 64, 64, 64, 64, True),
                                  (1, 10, 256, 64, 64, 64, 72, False), (3, 4, 128, 32, 64, 32, 80, True), (2, 4, 64, 64, 128, 64, 128, True)])
def test_stride2_dgrad_single_launch_matches_class_launches(case):
    """3x3 stride-2 data gradient of the few-channel layers: the one-launch kernel over shared dy tiles (dgrad_s2_kernels.hip) against
    the four parity-class launches (same products, different summation order) and against autograd."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import lib
    n, h, w, cin, cout, ldx, lddy, with_res = case
    shape = ops.conv_shape(n, h, w, cin, cout, 3, 2, in_ld=ldx, out_ld=lddy)
    wt = rnd((cout, cin, 3, 3), 21, (2.0 / (cin * 9)) ** 0.5)
    gy = rnd((n, cout, h // 2, w // 2), 22)
    res = rnd((n, cin, h, w), 23) if with_res else None
    wf, wd = ops.pack_weights(shape, wt.to(dev()))
    gyd = nhwc(gy, lddy)
    gyd[..., cout:] = 5.0
    res_d = nhwc(res, ldx) if with_res else None
    outs = []
    try:
        for off in (1, 0):
            lib().mi355det_debug_set(2, off)
            dx = torch.full((n, h, w, ldx), 3.0, dtype=torch.bfloat16, device=dev())
            ops.conv_dgrad(shape, gyd, wd, dx, residual=res_d, residual_ld=ldx if with_res else 0)
            outs.append(dx.float().cpu())
    finally:
        lib().mi355det_debug_set(2, 0)
    assert torch.equal(outs[0][..., cin:], outs[1][..., cin:])          # pitch padding untouched by both
    xr = torch.zeros(n, cin, h, w, requires_grad=True)
    F.conv2d(xr, wt.bfloat16().float(), stride=2, padding=1).backward(gy.bfloat16().float())
    ref = xr.grad + (res.bfloat16().float() if with_res else 0)
    tol = 2e-2 * ref.abs().max().item()
    for o in outs:
        assert (o[..., :cin].permute(0, 3, 1, 2) - ref).abs().max().item() < tol
    assert (outs[0] - outs[1]).abs().max().item() < 1e-2 * ref.abs().max().item()


@pytest.mark.parametrize("rows,c,c_pad", [(1600, 256, 256), (300, 96, 128), (257, 32, 32), (5000, 1024, 1024)])
def test_bn_finalize_many_rows(rows, c, c_pad):
    """rows > 256: two-stage reduction through the 64 spare rows; repeated launches on fresh data match a float64 reduction of the
    same partial rows."""
    from object_detectors_amd._lib import check, lib, ptr, stream_ptr
    d = dev()
    count = rows * 128
    gamma = (rnd((c,), 31, 0.3) + 1.0).to(d)
    beta = rnd((c,), 32, 0.2).to(d)
    for rep in range(3):
        part = torch.zeros(rows + 64, 2, c_pad)
        part[:rows, 0, :c] = rnd((rows, c), 40 + rep, 30.0) + 5.0
        part[:rows, 1, :c] = rnd((rows, c), 50 + rep, 20.0).abs() * 40 + 900.0
        pd = part.to(d)
        ss = torch.zeros(4 * c, device=d)
        rm, rv = torch.zeros(c, device=d), torch.ones(c, device=d)
        check(lib().mi355det_bn_finalize(ptr(pd), rows, c, c_pad, count, ptr(gamma), ptr(beta), 1e-5, 0.1, ptr(rm), ptr(rv), ptr(ss), stream_ptr()))
        s1 = part[:rows, 0, :c].double().sum(0)
        s2 = part[:rows, 1, :c].double().sum(0)
        mean = s1 / count
        var = (s2 / count - mean * mean).clamp_min(0)
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        got = ss.cpu().double()
        np.testing.assert_allclose(got[2 * c:3 * c], mean, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(got[3 * c:], invstd, rtol=1e-4)
        np.testing.assert_allclose(got[:c], gamma.cpu().double() * invstd, rtol=1e-4)
        np.testing.assert_allclose(rm.cpu().double(), 0.1 * mean, rtol=1e-4, atol=1e-6)


def test_stride2_dgrad_single_launch_full_size():
    """BASELINE size of the layer the single-launch kernel exists for (32 -> 64, 3x3 / 2, 640 px, batch 32): identical to the four
    class launches up to the summation order, on every pixel of the 839 MB gradient."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import lib
    n, h, w, cin, cout = 32, 640, 640, 32, 64
    shape = ops.conv_shape(n, h, w, cin, cout, 3, 2)
    g = torch.Generator(device="cpu").manual_seed(5)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5).to(dev())
    wf, wd = ops.pack_weights(shape, wt)
    gy = torch.randn(n, h // 2, w // 2, cout, device=dev(), generator=torch.Generator(device=dev()).manual_seed(6)).bfloat16()
    outs = []
    try:
        for off in (1, 0):
            lib().mi355det_debug_set(2, off)
            dx = torch.empty(n, h, w, cin, dtype=torch.bfloat16, device=dev())
            ops.conv_dgrad(shape, gy, wd, dx)
            outs.append(dx)
    finally:
        lib().mi355det_debug_set(2, 0)
    diff = (outs[0].float() - outs[1].float()).abs().max().item()
    scale = outs[0].float().abs().max().item()
    assert scale > 0 and diff <= 1e-2 * scale
    # one bf16 ulp at most wherever the two summation orders round differently
    rel = ((outs[0].float() - outs[1].float()).abs() / (outs[0].float().abs() + 1e-3)).max().item()
    assert rel < 2e-2


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 256, 3, 1), (2, 9, 11, 128, 256, 3, 1), (1, 20, 20, 512, 1024, 3, 1), (2, 40, 40, 256, 512, 3, 1),
                                  (3, 13, 13, 512, 256, 1, 1), (1, 26, 26, 256, 512, 3, 2), (2, 1, 1, 256, 256, 3, 1), (5, 2, 1, 64, 256, 3, 1),
                                  (1, 8, 8, 768, 256, 1, 1), (4, 20, 20, 512, 512, 3, 1), (32, 20, 20, 512, 1024, 3, 1), (8, 80, 80, 128, 256, 3, 1),
                                  # more than 8 channel tiles: the blocked tile order (8 x 4 blocks per XCD round; 10 and 11 tiles: narrow last group)
                                  (3, 24, 28, 64, 2560, 3, 1), (1, 50, 50, 64, 2816, 1, 1)])
def test_phase_staggered_kernel_matches_default(case):
    """Tile configuration 40 (igemm8_kernels.hip: 256x256x64, four phases per k-step, SIMD partners one barrier apart) against PyTorch fp32
    and the default configuration: forward (+ BN partial statistics), data gradient (+ residual); image edges, tile tails, one- and many-step K."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import lib
    n, h, w, cin, cout, k, s = case
    x = rnd((n, cin, h, w), 11)
    wt = rnd((cout, cin, k, k), 12, (2.0 / (cin * k * k)) ** 0.5)
    xr, wr = x.clone().requires_grad_(True), wt.clone()
    y_ref = F.conv2d(xr, wr, padding=k // 2, stride=s)
    gy = rnd(tuple(y_ref.shape), 13)
    y_ref.backward(gy)
    shape = ops.conv_shape(n, h, w, cin, cout, k, s)
    wf, wd = ops.pack_weights(shape, wt.to(dev()))
    xd, gyd = nhwc(x), nhwc(gy)
    res = nhwc(rnd((n, cin, h, w), 14))
    rows = ops.conv_stats_rows(shape)
    outs = {}
    try:
        for cfg in (1, 40):
            lib().mi355det_debug_set(0, cfg)
            y = torch.full((n, shape.ho, shape.wo, cout), 5.0, device=dev(), dtype=torch.bfloat16)
            stats = torch.zeros((rows + 64, 2, ops.cout_pad_of(cout)), device=dev())
            ops.conv_fwd(shape, xd, wf, y, stats=stats)
            dx = torch.full((n, h, w, cin), 5.0, device=dev(), dtype=torch.bfloat16)
            ops.conv_dgrad(shape, gyd, wd, dx)
            dxr = torch.full((n, h, w, cin), 5.0, device=dev(), dtype=torch.bfloat16)
            ops.conv_dgrad(shape, gyd, wd, dxr, residual=res, residual_ld=cin)
            torch.cuda.synchronize()
            outs[cfg] = (y.float().cpu(), stats[:rows].sum(0).cpu(), dx.float().cpu(), dxr.float().cpu())
    finally:
        lib().mi355det_debug_set(0, 0)
    yr = y_ref.detach().permute(0, 2, 3, 1)
    gr = xr.grad.permute(0, 2, 3, 1)
    y1, st1, dx1, dxr1 = outs[1]
    for cfg in (40,):
        y8, st8, dx8, dxr8 = outs[cfg]
        assert float((y8 - yr).abs().max()) <= 2e-2 * float(yr.abs().max()), cfg
        assert float((dx8 - gr).abs().max()) <= 2e-2 * float(gr.abs().max()), cfg
        assert float((y8 - y1).abs().max()) <= 1e-2 * float(y1.abs().max()), cfg
        assert float((dx8 - dx1).abs().max()) <= 1e-2 * float(dx1.abs().max()), cfg
        assert float((dxr8 - dxr1).abs().max()) <= 1e-2 * float(dxr1.abs().max()), cfg
        torch.testing.assert_close(st8, st1, rtol=2e-2, atol=2e-2 * float(st1.abs().max()))


@pytest.mark.parametrize("case", [(2, 26, 26, 64, 128), (1, 16, 48, 128, 256), (3, 12, 20, 64, 64), (2, 40, 40, 128, 128)])
@pytest.mark.parametrize("with_res", [False, True])
def test_stride2_dgrad_class_concatenated_form(case, with_res):
    """3x3 / stride-2 data gradient as two class-concatenated GEMMs (N = 2*Cin, y stride 2 / x stride 1 epilogue on the [n, h, w/2, 2*Cin]
    view) against the four class launches and against PyTorch fp32, with and without the residual; both the single-call pack and the
    batched pack table produce the concatenated weights."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import lib
    n, h, w, cin, cout = case
    x = rnd((n, cin, h, w), 31)
    wt = rnd((cout, cin, 3, 3), 32, (2.0 / (cin * 9)) ** 0.5)
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wt, stride=2, padding=1)
    gy = rnd(tuple(y_ref.shape), 33)
    y_ref.backward(gy)
    res = rnd((n, cin, h, w), 34)
    want = xr.grad + (res if with_res else 0)
    shape = ops.conv_shape(n, h, w, cin, cout, 3, 2)
    wf, wd = ops.pack_weights(shape, wt.to(dev()))
    gyd = nhwc(gy)
    resd = nhwc(res) if with_res else None
    outs = {}
    try:
        for form in (0, 1):
            lib().mi355det_debug_set(5, form)
            dx = torch.full((n, h, w, cin), 7.0, device=dev(), dtype=torch.bfloat16)
            ops.conv_dgrad(shape, gyd, wd, dx, residual=resd, residual_ld=cin if with_res else 0)
            torch.cuda.synchronize()
            outs[form] = dx.float().cpu().permute(0, 3, 1, 2)
    finally:
        lib().mi355det_debug_set(5, -1)
    scale = float(want.abs().max())
    assert float((outs[0] - want).abs().max()) <= 2e-2 * scale
    assert float((outs[1] - want).abs().max()) <= 2e-2 * scale
    assert float((outs[1] - outs[0]).abs().max()) <= 1e-2 * scale


@pytest.mark.parametrize("case", [(2, 7, 7, 256, 1024, 3), (1, 13, 13, 128, 2176, 3), (3, 5, 9, 256, 8192, 1)])
@pytest.mark.parametrize("with_res", [False, True])
def test_dgrad_split_k_small_maps(case, with_res):
    """Data gradient with few output pixels and a deep reduction (the 1204-class RetinaNet head on the small pyramid levels,
    retinanet.py:75-105): mi355det_conv_dgrad_ws splits the channel axis over workgroups (fp32 partial tiles, fixed-order sum, one rounding)
    - against the plain launch and against PyTorch fp32; the workspace query is 0 where the form does not apply, and conv_dgrad_ws is then
    conv_dgrad; a short workspace is refused."""
    import ctypes as C
    from object_detectors_amd import ops
    from object_detectors_amd._lib import check, lib, ptr, stream_ptr
    n, h, w, cin, cout, k = case
    x = rnd((n, cin, h, w), 41)
    wt = rnd((cout, cin, k, k), 42, (2.0 / (cin * k * k)) ** 0.5)
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wt, stride=1, padding=k // 2)
    gy = rnd(tuple(y_ref.shape), 43)
    y_ref.backward(gy)
    res = rnd((n, cin, h, w), 44)
    want = xr.grad + (res if with_res else 0)
    shape = ops.conv_shape(n, h, w, cin, cout, k, 1)
    wf, wd = ops.pack_weights(shape, wt.to(dev()))
    gyd = nhwc(gy)
    resd = nhwc(res) if with_res else None
    L = lib()
    need = L.mi355det_conv_dgrad_workspace(C.byref(shape))
    assert need > 0 and need % (n * h * w * cin * 4) == 0          # a whole number of fp32 partial tensors
    ws = torch.empty(need, device=dev(), dtype=torch.uint8)
    dx_s = torch.full((n, h, w, cin), 7.0, device=dev(), dtype=torch.bfloat16)
    check(L.mi355det_conv_dgrad_ws(C.byref(shape), ptr(gyd), ptr(wd), ptr(dx_s), ptr(resd), cin if with_res else 0, ptr(ws), ws.numel(), stream_ptr()),
          "conv_dgrad_ws")
    dx_p = torch.full((n, h, w, cin), 7.0, device=dev(), dtype=torch.bfloat16)
    ops.conv_dgrad(shape, gyd, wd, dx_p, residual=resd, residual_ld=cin if with_res else 0)
    torch.cuda.synchronize()
    a, b = dx_s.float().cpu().permute(0, 3, 1, 2), dx_p.float().cpu().permute(0, 3, 1, 2)
    scale = float(want.abs().max())
    assert float((a - want).abs().max()) <= 2e-2 * scale
    assert float((b - want).abs().max()) <= 2e-2 * scale
    assert float((a - b).abs().max()) <= 1e-2 * scale
    # deterministic: a second run is bit-identical
    dx_2 = torch.empty_like(dx_s)
    check(L.mi355det_conv_dgrad_ws(C.byref(shape), ptr(gyd), ptr(wd), ptr(dx_2), ptr(resd), cin if with_res else 0, ptr(ws), ws.numel(), stream_ptr()),
          "conv_dgrad_ws")
    torch.cuda.synchronize()
    assert torch.equal(dx_2, dx_s)
    with pytest.raises(ValueError):
        check(L.mi355det_conv_dgrad_ws(C.byref(shape), ptr(gyd), ptr(wd), ptr(dx_2), None, 0, ptr(ws), 1024, stream_ptr()), "conv_dgrad_ws")
    # a big map is not split: no workspace, and the _ws entry falls through to the plain launch
    big = ops.conv_shape(2, 64, 64, 128, 256, 3, 1)
    assert L.mi355det_conv_dgrad_workspace(C.byref(big)) == 0


@pytest.mark.parametrize("case", [(2, 25, 25, 64, 256, 3), (1, 50, 38, 256, 64, 1), (3, 13, 13, 256, 256, 3), (2, 40, 40, 512, 512, 3), (8, 20, 20, 128, 1024, 1)])
@pytest.mark.parametrize("relu,with_scale", [(1, True), (1, False), (0, True)])
def test_dgrad_with_relu_affine_backward_in_the_epilogue(case, relu, with_scale):
    """mi355det_conv_dgrad_mask (the FrozenBN / ReLU backward of the producing layer folded into the data gradient's epilogue) stores exactly
    what mi355det_conv_dgrad followed by mi355det_relu_affine_bwd stores; every tile configuration the autotuner may pick is covered through
    mi355det_debug_set(0, cfg) (1 = 128x128, 3 = 256x256, 15 = shared pixel tiles, 40 = phase-staggered)."""
    import ctypes as C
    from object_detectors_amd import ops
    from object_detectors_amd._lib import check, lib, ptr, stream_ptr
    n, h, w, cin, cout, k = case
    wt = rnd((cout, cin, k, k), 52, (2.0 / (cin * k * k)) ** 0.5)
    shape = ops.conv_shape(n, h, w, cin, cout, k, 1)
    wf, wd = ops.pack_weights(shape, wt.to(dev()))
    gyd = nhwc(rnd((n, cout, h, w), 53))
    act = nhwc(rnd((n, cin, h, w), 54))                                # the producing layer's stored activation (sign = ReLU mask)
    scale = (1.0 + 0.5 * rnd((cin,), 55)).to(dev()) if with_scale else None
    L = lib()
    g = torch.empty((n, h, w, cin), device=dev(), dtype=torch.bfloat16)
    ops.conv_dgrad(shape, gyd, wd, g)
    want = ops.relu_affine_bwd(g, act, scale=scale, relu=bool(relu))
    cfgs = [0, 1] + ([3, 40] if cin % 256 == 0 and cout % 64 == 0 else []) + ([15] if k == 3 and cout % 64 == 0 and cin % 128 == 0 else [])
    try:
        for cfg in cfgs:
            L.mi355det_debug_set(0, cfg)
            ops.conv_dgrad(shape, gyd, wd, g)                              # the same tile configuration for both sides
            want = ops.relu_affine_bwd(g, act, scale=scale, relu=bool(relu))
            got = torch.full((n, h, w, cin), 7.0, device=dev(), dtype=torch.bfloat16)
            check(L.mi355det_conv_dgrad_mask(C.byref(shape), ptr(gyd), ptr(wd), ptr(got), ptr(act), cin, ptr(scale), relu, stream_ptr()), "conv_dgrad_mask")
            torch.cuda.synchronize()
            assert torch.equal(got.float(), want.float()), cfg              # value-equal (a masked entry is +0 here, -0 * scale there)
    finally:
        L.mi355det_debug_set(0, 0)
    s2 = ops.conv_shape(n, h - h % 2, w - w % 2, cin, cout, 3, 2)
    assert L.mi355det_conv_dgrad_mask(C.byref(s2), ptr(gyd), ptr(wd), ptr(got), ptr(act), cin, None, 1, stream_ptr()) == -1      # stride 2 refused


@pytest.mark.parametrize("case", [(3, 24, 24, 128, 256, 3), (2, 40, 40, 256, 512, 3), (1, 20, 20, 512, 1024, 3), (5, 13, 11, 256, 256, 1), (1, 9, 7, 64, 256, 3),
                                  (2, 52, 52, 64, 256, 3)])
def test_phase_staggered_kernel_on_shorter_pixel_tiles(case):
    """Tile configurations 44 / 45 (igemm8_kernel on 224 / 208-pixel tiles: VERDICT r3 item 1a, tile quantisation) against
    configuration 40 (the same loop on 256 pixels): the convolution sums the same products in the same k order, so the stored tensors are
    BIT-identical - forward (+ BN partial statistics, whose rows are laid out differently but must add up to the same per-channel sums),
    data gradient with and without the residual, the FrozenBN-affine epilogue of inference plans.  Pixel counts that are no multiple of
    any tile height (tail tiles, the tile that ends inside a 128-pixel statistics row) and one that spans several images per tile."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import lib
    n, h, w, cin, cout, k = case
    x = rnd((n, cin, h, w), 1)
    wt = rnd((cout, cin, k, k), 2, (2.0 / (cin * k * k)) ** 0.5)
    gy = rnd((n, cout, h, w), 3)
    shape = ops.conv_shape(n, h, w, cin, cout, k, 1)
    wf, wd = ops.pack_weights(shape, wt.to(dev()))
    xd, gyd = nhwc(x), nhwc(gy)
    res = nhwc(rnd((n, cin, h, w), 4))
    rows = ops.conv_stats_rows(shape)
    dgrad_ok = cin % 256 == 0                     # the data gradient's output channels must fill a 256-wide tile
    aff_s, aff_b = (1.0 + 0.3 * rnd((cout,), 6)).to(dev()), (0.2 * rnd((cout,), 7)).to(dev())
    resy = nhwc(rnd((n, cout, h, w), 8))
    outs = {}
    try:
        for cfg in (40, 44, 45, 44, 40):
            lib().mi355det_debug_set(0, cfg)
            y = torch.full((n, h, w, cout), 5.0, device=dev(), dtype=torch.bfloat16)
            stats = torch.full((rows + 64, 2, ops.cout_pad_of(cout)), 77.0, device=dev())       # stale values: every row must be rewritten
            ops.conv_fwd(shape, xd, wf, y, stats=stats)
            dx = torch.full((n, h, w, cin), 5.0, device=dev(), dtype=torch.bfloat16)
            dxr = torch.full((n, h, w, cin), 5.0, device=dev(), dtype=torch.bfloat16)
            if dgrad_ok:
                ops.conv_dgrad(shape, gyd, wd, dx)
                ops.conv_dgrad(shape, gyd, wd, dxr, residual=res, residual_ld=cin)
            ya = torch.full((n, h, w, cout), 5.0, device=dev(), dtype=torch.bfloat16)
            ops.conv_fwd_ex(shape, xd, wf, ya, scale=aff_s, shift=aff_b, residual=resy, residual_ld=cout, leaky_slope=0.1)
            torch.cuda.synchronize()
            outs.setdefault(cfg, []).append((y.clone(), stats[:rows].double().sum(0).cpu(), dx.clone(), dxr.clone(), ya.clone()))
    finally:
        lib().mi355det_debug_set(0, 0)
    y0, st0, dx0, dxr0, ya0 = outs[40][0]
    yr = F.conv2d(x, wt, padding=k // 2).permute(0, 2, 3, 1)
    assert float((y0.float().cpu() - yr).abs().max()) <= 2e-2 * float(yr.abs().max())
    yf = y0.float().reshape(-1, cout).double().cpu()
    for cfg, runs in outs.items():
        for y, st, dx, dxr, ya in runs:
            assert torch.equal(y, y0) and torch.equal(ya, ya0), cfg
            assert torch.equal(dx, dx0) and torch.equal(dxr, dxr0), cfg
            torch.testing.assert_close(st[0, :cout], yf.sum(0), rtol=1e-4, atol=1e-3, msg=f"cfg {cfg} sum")
            torch.testing.assert_close(st[1, :cout], (yf * yf).sum(0), rtol=1e-4, atol=1e-3, msg=f"cfg {cfg} sumsq")
