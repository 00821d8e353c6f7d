"""Darknet stem recompute kernels (csrc/stem_kernels.hip; darknet.py:41-43,74-76 conv1 -> bn1 -> LeakyReLU(0.1)) against a plain PyTorch fp32
evaluation of the same ops on the same bf16-rounded operands: forward statistics, activation, BatchNorm-backward sums, and the weight /
gamma / beta gradients that the fused apply + weight-gradient kernel produces without ever storing z or dz."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

SLOPE, EPS = 0.1, 1e-5


def dev():
    return torch.device("cuda:0")


def _vp(t):
    return C.c_void_p(t.data_ptr())


def _inputs(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn((n, 3, h, w), generator=g).bfloat16().float()
    wt = (torch.randn((32, 3, 3, 3), generator=g) * (2.0 / 27) ** 0.5).bfloat16().float()
    gamma = 1.0 + 0.2 * torch.randn(32, generator=g)
    beta = 0.1 * torch.randn(32, generator=g)
    da = (torch.randn((n, 32, h, w), generator=g) * 0.05).bfloat16().float()
    return img, wt, gamma, beta, da


def _pack(wt):
    wp = torch.zeros((32, 32), dtype=torch.bfloat16, device=dev())
    wp[:, :27] = wt.permute(0, 2, 3, 1).reshape(32, 27).to(dev()).bfloat16()          # k = (kh*3+kw)*3 + c
    return wp


def _reference(img, wt, gamma, beta, da):
    x = img.double()
    w = wt.double().requires_grad_(True)
    ga, be = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    z = F.conv2d(x, w, padding=1)
    mean = z.mean((0, 2, 3))
    var = z.var((0, 2, 3), unbiased=False)
    invstd = (var + EPS).rsqrt()
    y = (z - mean[None, :, None, None]) * (invstd * ga)[None, :, None, None] + be[None, :, None, None]
    a = F.leaky_relu(y, SLOPE)
    (a * da.double()).sum().backward()
    return dict(z=z.detach(), a=a.detach(), mean=mean.detach(), invstd=invstd.detach(), dw=w.grad, dgamma=ga.grad, dbeta=be.grad)


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 8, 32), (3, 40, 96), (2, 128, 64)])
def test_stem_forward_and_backward_match_fp32_reference(shape):
    from object_detectors_amd._lib import check, lib
    L = lib()
    n, h, w = shape
    img, wt, gamma, beta, da = _inputs(n, h, w, 7 + h)
    ref = _reference(img, wt, gamma, beta, da)
    imgd, wp = img.to(dev()), _pack(wt)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = L.mi355det_stem_rows(n, h, w)
    assert 0 < rows <= n * (h // 8) * (w // 32)
    count = n * h * w
    # ---- forward statistics
    part = torch.zeros((rows + 64, 2, 32), device=dev())
    check(L.mi355det_stem_fwd_stats(_vp(imgd), _vp(wp), _vp(part), n, h, w, st), "stem_fwd_stats")
    s = part[:rows].double().sum(0).cpu()
    torch.testing.assert_close(s[0], ref["z"].sum((0, 2, 3)), rtol=1e-4, atol=1e-3 * count ** 0.5)
    torch.testing.assert_close(s[1], (ref["z"] ** 2).sum((0, 2, 3)), rtol=1e-4, atol=1e-6 * count)
    gd, bd = gamma.to(dev()), beta.to(dev())
    rm, rv = torch.zeros(32, device=dev()), torch.ones(32, device=dev())
    ss = torch.zeros(128, device=dev())
    check(L.mi355det_bn_finalize(_vp(part), rows, 32, 32, count, _vp(gd), _vp(bd), EPS, 0.1, _vp(rm), _vp(rv), _vp(ss), st), "bn_finalize")
    torch.testing.assert_close(ss[64:96].cpu().double(), ref["mean"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ss[96:128].cpu().double(), ref["invstd"], rtol=1e-4, atol=1e-5)
    # ---- activation
    a = torch.full((n, h, w, 32), 7.0, dtype=torch.bfloat16, device=dev())
    check(L.mi355det_stem_fwd_apply(_vp(imgd), _vp(wp), _vp(ss), SLOPE, _vp(a), 32, n, h, w, st), "stem_fwd_apply")
    got = a.float().cpu().permute(0, 3, 1, 2).double()
    assert float((got - ref["a"]).abs().max()) <= 1e-2 * float(ref["a"].abs().max())      # bf16 output rounding
    # ---- backward sums
    dad = da.permute(0, 2, 3, 1).contiguous().to(dev()).bfloat16()
    part2 = torch.zeros((rows + 64, 2, 32), device=dev())
    check(L.mi355det_stem_bwd_reduce(_vp(imgd), _vp(wp), _vp(ss), SLOPE, _vp(dad), 32, _vp(part2), n, h, w, st), "stem_bwd_reduce")
    sums = torch.zeros(64, device=dev())
    check(L.mi355det_bn_bwd_sum_partials(_vp(part2), rows, 32, 32, _vp(sums), st), "bn_bwd_sum_partials")
    sc = sums.cpu().double()
    tol = 2e-3 * float(ref["dgamma"].abs().max() + ref["dbeta"].abs().max())
    torch.testing.assert_close(sc[:32], ref["dbeta"], rtol=2e-3, atol=tol)
    torch.testing.assert_close(sc[32:], ref["dgamma"], rtol=2e-3, atol=tol)
    # ---- dz -> weight gradient (dz is rounded to bf16 for the MFMA like every stored dz of the engine), gamma / beta gradients
    slab = torch.zeros((rows, 1024), device=dev())
    dw = torch.zeros((32, 32), device=dev())
    dg, db = torch.zeros(32, device=dev()), torch.zeros(32, device=dev())
    check(L.mi355det_stem_bwd_apply_wgrad(_vp(imgd), _vp(wp), _vp(ss), _vp(sums), SLOPE, _vp(dad), 32, _vp(slab), _vp(dw), _vp(dg), _vp(db),
                                          n, h, w, st), "stem_bwd_apply_wgrad")
    torch.cuda.synchronize()
    want = ref["dw"].permute(0, 2, 3, 1).reshape(32, 27)
    gotw = dw.cpu().double()
    assert float(gotw[:, 27:].abs().max()) == 0.0                                            # padded k columns stay zero
    assert float((gotw[:, :27] - want).abs().max()) <= 1.5e-2 * float(want.abs().max())
    cos = float((gotw[:, :27] * want).sum() / (gotw[:, :27].norm() * want.norm()))
    assert cos > 0.9999
    torch.testing.assert_close(dg.cpu().double(), sc[32:], rtol=0, atol=0)
    torch.testing.assert_close(db.cpu().double(), sc[:32], rtol=0, atol=0)
    # ---- fixed-order reductions: a second run is bit-identical
    dw2 = torch.zeros((32, 32), device=dev())
    part3 = torch.zeros_like(part2)
    check(L.mi355det_stem_bwd_reduce(_vp(imgd), _vp(wp), _vp(ss), SLOPE, _vp(dad), 32, _vp(part3), n, h, w, st), "stem_bwd_reduce")
    check(L.mi355det_stem_bwd_apply_wgrad(_vp(imgd), _vp(wp), _vp(ss), _vp(sums), SLOPE, _vp(dad), 32, _vp(slab), _vp(dw2), None, None,
                                          n, h, w, st), "stem_bwd_apply_wgrad")
    torch.cuda.synchronize()
    assert torch.equal(part3[:rows], part2[:rows]) and torch.equal(dw2, dw)
    # ---- the single-pass form (stem_bwd_fused + stem_bwd_finish): A = dy^T [im2col | 1] and the Gram matrix on MFMA, sums and dW from them
    slab4 = torch.zeros((rows, 2048), device=dev())
    ag, sums4 = torch.zeros(2048, device=dev()), torch.zeros(64, device=dev())
    dw4, dg4, db4 = torch.zeros((32, 32), device=dev()), torch.zeros(32, device=dev()), torch.zeros(32, device=dev())
    check(L.mi355det_stem_bwd_fused(_vp(imgd), _vp(wp), _vp(ss), SLOPE, _vp(dad), 32, _vp(slab4), _vp(ag), _vp(sums4), n, h, w, st), "stem_bwd_fused")
    check(L.mi355det_stem_bwd_finish(_vp(wp), _vp(ss), _vp(ag), _vp(sums4), count, _vp(dw4), _vp(dg4), _vp(db4), st), "stem_bwd_finish")
    torch.cuda.synchronize()
    s4 = sums4.cpu().double()
    torch.testing.assert_close(s4[:32], ref["dbeta"], rtol=2e-3, atol=tol)
    torch.testing.assert_close(s4[32:], ref["dgamma"], rtol=2e-3, atol=tol)
    torch.testing.assert_close(dg4.cpu().double(), s4[32:], rtol=0, atol=0)
    torch.testing.assert_close(db4.cpu().double(), s4[:32], rtol=0, atol=0)
    # the Gram matrix: sum of im2col columns in row 27, pixel count in [27][27], symmetric
    G = ag[1024:].reshape(32, 32).cpu().double()
    col = F.unfold(img.double(), 3, padding=1).reshape(n, 3, 9, h * w).permute(0, 3, 2, 1).reshape(-1, 27)          # k = (kh*3+kw)*3 + c
    torch.testing.assert_close(G[27, :27], col.sum(0), rtol=1e-4, atol=1e-3 * count ** 0.5)
    assert abs(float(G[27, 27]) - count) <= 1e-6 * count and float((G - G.T).abs().max()) <= 1e-6 * float(G.abs().max())
    torch.testing.assert_close(G[:27, :27], col.T @ col, rtol=1e-4, atol=1e-5 * count)
    got4 = dw4.cpu().double()
    assert float(got4[:, 27:].abs().max()) == 0.0
    assert float((got4[:, :27] - want).abs().max()) <= 1.5e-2 * float(want.abs().max())
    assert float((got4[:, :27] * want).sum() / (got4[:, :27].norm() * want.norm())) > 0.9999
    assert float((got4 - gotw).abs().max()) <= 1e-2 * float(want.abs().max())                 # the two forms agree (different rounding points)
    dw5 = torch.zeros((32, 32), device=dev())
    slab5, ag5, sums5 = torch.zeros_like(slab4), torch.zeros_like(ag), torch.zeros_like(sums4)
    check(L.mi355det_stem_bwd_fused(_vp(imgd), _vp(wp), _vp(ss), SLOPE, _vp(dad), 32, _vp(slab5), _vp(ag5), _vp(sums5), n, h, w, st), "stem_bwd_fused")
    check(L.mi355det_stem_bwd_finish(_vp(wp), _vp(ss), _vp(ag5), _vp(sums5), count, _vp(dw5), None, None, st), "stem_bwd_finish")
    torch.cuda.synchronize()
    assert torch.equal(ag5, ag) and torch.equal(sums5, sums4) and torch.equal(dw5, dw4)      # fixed order: bit-reproducible


def test_stem_rejects_unsupported_sizes():
    from object_detectors_amd._lib import lib
    L = lib()
    assert L.mi355det_stem_rows(1, 20, 32) == 0 and L.mi355det_stem_rows(1, 8, 48) == 0
    x = torch.zeros((1, 3, 20, 32), device=dev())
    wp = torch.zeros((32, 32), dtype=torch.bfloat16, device=dev())
    part = torch.zeros((80, 2, 32), device=dev())
    assert L.mi355det_stem_fwd_stats(_vp(x), _vp(wp), _vp(part), 1, 20, 32, None) == -1


@pytest.mark.parametrize("shape", [(2, 64, 64), (1, 16, 32), (3, 48, 96), (2, 128, 64)])
def test_fused_stem_and_first_downsampling_conv(shape):
    """csrc/stem_l1_kernels.hip: a0 = lrelu(bn1(conv1(img))) and z1 = layer1.ds_conv(a0) (32 -> 64, 3x3, stride 2) from ONE kernel against
    PyTorch fp32 on the same bf16-rounded operands: the side-output activation equals stem_fwd_apply's bit for bit, z1 and its statistics
    match the convolution of that (bf16) activation."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import check, lib
    L = lib()
    n, h, w = shape
    img, wt, gamma, beta, _da = _inputs(n, h, w, 21 + h)
    g = torch.Generator().manual_seed(99)
    w1 = (torch.randn((64, 32, 3, 3), generator=g) * (2.0 / 288) ** 0.5).bfloat16().float()
    imgd, wp = img.to(dev()), _pack(wt)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = L.mi355det_stem_rows(n, h, w)
    part = torch.zeros((rows + 64, 2, 32), device=dev())
    check(L.mi355det_stem_fwd_stats(_vp(imgd), _vp(wp), _vp(part), n, h, w, st), "stem_fwd_stats")
    gd, bd = gamma.to(dev()), beta.to(dev())
    ss = torch.zeros(128, device=dev())
    check(L.mi355det_bn_finalize(_vp(part), rows, 32, 32, n * h * w, _vp(gd), _vp(bd), EPS, 0.1, None, None, _vp(ss), st), "bn_finalize")
    a_ref = torch.full((n, h, w, 32), 7.0, dtype=torch.bfloat16, device=dev())
    check(L.mi355det_stem_fwd_apply(_vp(imgd), _vp(wp), _vp(ss), SLOPE, _vp(a_ref), 32, n, h, w, st), "stem_fwd_apply")
    shp1 = ops.conv_shape(n, h, w, 32, 64, 3, 2)
    wf1, _wd1 = ops.pack_weights(shp1, w1.to(dev()))
    rows1 = L.mi355det_stem_l1_rows(n, h, w)
    assert rows1 > 0
    a0 = torch.full((n, h, w, 32), 5.0, dtype=torch.bfloat16, device=dev())
    z1 = torch.full((n, h // 2, w // 2, 64), 5.0, dtype=torch.bfloat16, device=dev())
    stats = torch.zeros((rows1 + 64, 2, 64), device=dev())
    check(L.mi355det_stem_l1_fwd(_vp(imgd), _vp(wp), _vp(ss), SLOPE, _vp(wf1), _vp(a0), 32, _vp(z1), 64, _vp(stats), n, h, w, st), "stem_l1_fwd")
    torch.cuda.synchronize()
    assert torch.equal(a0, a_ref)                                            # same arithmetic, same rounding
    zr = F.conv2d(a_ref.float().cpu().permute(0, 3, 1, 2), w1, stride=2, padding=1)
    got = z1.float().cpu().permute(0, 3, 1, 2)
    assert float((got - zr).abs().max()) <= 1e-2 * float(zr.abs().max())
    s = stats[:rows1].double().sum(0).cpu()
    zq = z1.double().cpu()
    torch.testing.assert_close(s[0], zq.sum((0, 1, 2)), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(s[1], (zq ** 2).sum((0, 1, 2)), rtol=1e-5, atol=1e-3)
    # without the side output
    z1b = torch.zeros_like(z1)
    check(L.mi355det_stem_l1_fwd(_vp(imgd), _vp(wp), _vp(ss), SLOPE, _vp(wf1), None, 0, _vp(z1b), 64, _vp(stats), n, h, w, st), "stem_l1_fwd")
    torch.cuda.synchronize()
    assert torch.equal(z1b, z1)
    # inference form: layer 1's folded BN + LeakyReLU in the epilogue, the stem activation never stored
    g2 = torch.Generator().manual_seed(7)
    sc1, sh1 = (1.0 + 0.3 * torch.randn(64, generator=g2)), 0.2 * torch.randn(64, generator=g2)
    ss1 = torch.cat([sc1, sh1]).to(dev())
    a1 = torch.full((n, h // 2, w // 2, 64), 5.0, dtype=torch.bfloat16, device=dev())
    check(L.mi355det_stem_l1_fwd_eval(_vp(imgd), _vp(wp), _vp(ss), SLOPE, _vp(wf1), _vp(ss1), _vp(a1), 64, n, h, w, st), "stem_l1_fwd_eval")
    torch.cuda.synchronize()
    want1 = F.leaky_relu(zr * sc1[None, :, None, None] + sh1[None, :, None, None], SLOPE)
    got1 = a1.float().cpu().permute(0, 3, 1, 2)
    assert float((got1 - want1).abs().max()) <= 1.5e-2 * float(want1.abs().max())
