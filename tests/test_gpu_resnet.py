"""GPU parity of the ResNet-FPN / RetinaNet kernels (conv with fused affine epilogue, 1x1 stride-2 data gradient, 7x7 stem
as im2col + GEMM, max-pool, ReLU/FrozenBN backward, FPN nearest upsample-add, batched RetinaNet loss) against plain
PyTorch fp32 evaluations of the same ops on the same bf16-rounded operands and against oracle/tv_oracle.py."""
import numpy as np
import pytest

from oracle import detrand
from oracle import tv_oracle as tv

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

from tests.test_gpu_conv import dev, nhwc, rnd  # noqa: E402


def to_nchw(y):
    return y.float().permute(0, 3, 1, 2).cpu()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


AFF_CASES = [
    # n, h, w, cin, cout, k, s, scale, shift, res, relu
    (2, 16, 16, 64, 256, 1, 1, True, True, False, True),      # bottleneck conv1 + FrozenBN + ReLU
    (2, 16, 16, 64, 64, 3, 2, True, True, False, True),       # bottleneck conv2 (stride on the 3x3, resnet v1.5)
    (2, 8, 8, 64, 256, 1, 1, True, True, True, True),         # conv3 + FrozenBN + identity + ReLU
    (2, 16, 16, 256, 512, 1, 2, True, True, False, False),    # downsample 1x1 stride 2 + FrozenBN (no ReLU)
    (1, 25, 25, 256, 256, 3, 2, False, True, False, False),   # LastLevelP6 3x3 stride 2 on an odd map, bias only
    (1, 13, 13, 256, 256, 3, 1, False, True, False, True),    # head tower conv + bias + ReLU
    (1, 7, 7, 2048, 256, 1, 1, False, True, False, False),    # FPN inner block
]


@pytest.mark.parametrize("case", AFF_CASES)
def test_conv_affine_epilogue_and_grads(case):
    from object_detectors_amd import ops
    n, h, w, cin, cout, k, s, use_sc, use_sh, use_res, relu = case
    x = rnd((n, cin, h, w), 1)
    wt = rnd((cout, cin, k, k), 2, (2.0 / (cin * k * k)) ** 0.5)
    sc = (torch.rand(cout, generator=torch.Generator().manual_seed(3)) + 0.5) if use_sc else None
    sh = torch.randn(cout, generator=torch.Generator().manual_seed(4)) * 0.3 if use_sh else None
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    z = F.conv2d(xr, wr, stride=s, padding=(k - 1) // 2)
    res = rnd(tuple(z.shape), 5) if use_res else None
    y = z
    if sc is not None:
        y = y * sc.view(1, -1, 1, 1)
    if sh is not None:
        y = y + sh.view(1, -1, 1, 1)
    if res is not None:
        y = y.bfloat16().float() + res            # the kernel adds the identity to the bf16-rounded branch value
    if relu:
        y = torch.relu(y)
    shape = ops.conv_shape(n, h, w, cin, cout, k, s)
    wf, wd = ops.pack_weights(shape, wt.to(dev()))
    xd = nhwc(x)
    yd = torch.zeros((n, shape.ho, shape.wo, cout), device=dev(), dtype=torch.bfloat16)
    scd = sc.to(dev()) if sc is not None else None
    shd = sh.to(dev()) if sh is not None else None
    resd = nhwc(res) if res is not None else None
    ops.conv_fwd_ex(shape, xd, wf, yd, scale=scd, shift=shd, residual=resd, residual_ld=cout, relu=relu)
    torch.cuda.synchronize()
    tol = 2e-2 * float(y.detach().abs().max())
    assert float((to_nchw(yd) - y.detach()).abs().max()) <= tol

    # backward of the raw convolution: dgrad (incl. the empty parity classes of a 1x1 stride-2 conv) and wgrad
    gy = rnd(tuple(z.shape), 6)
    z.backward(gy)
    gyd = nhwc(gy)
    dx = torch.full((n, h, w, cin), 7.0, device=dev(), dtype=torch.bfloat16)     # every element must be overwritten
    ops.conv_dgrad(shape, gyd, wd, dx)
    dw = torch.zeros((cout, k, k, cin), device=dev())
    ops.conv_wgrad(shape, xd, gyd, dw)
    torch.cuda.synchronize()
    assert float((to_nchw(dx) - xr.grad).abs().max()) <= 2e-2 * float(xr.grad.abs().max()) + 1e-6
    assert float((dw.permute(0, 3, 1, 2).cpu() - wr.grad).abs().max()) <= 2e-2 * float(wr.grad.abs().max())
    # dgrad with a residual (gradient accumulation of the skip branch)
    r2 = rnd((n, cin, h, w), 8)
    ops.conv_dgrad(shape, gyd, wd, dx, residual=nhwc(r2), residual_ld=cin)
    torch.cuda.synchronize()
    assert float((to_nchw(dx) - (xr.grad + r2)).abs().max()) <= 2e-2 * float((xr.grad + r2).abs().max())


def test_head_conv_into_level_concatenated_tensor():
    """cls_logits conv (retinanet.py:95,163-170): fp32 [N, sum HWA, K] written in place by each level's convolution."""
    from object_detectors_amd import ops
    n, cin, A, K = 2, 256, 9, 7
    cout = A * K
    levels = [(8, 8), (4, 4)]
    tot = sum(h * w * A for h, w in levels)
    wt = rnd((cout, cin, 3, 3), 2, 0.02)
    bias = torch.randn(cout, generator=torch.Generator().manual_seed(3))
    out = torch.full((n, tot, K), -99.0, device=dev())
    refs, off = [], 0
    for li, (h, w) in enumerate(levels):
        x = rnd((n, cin, h, w), 10 + li)
        y = F.conv2d(x, wt, bias, padding=1)
        refs.append(y.view(n, A, K, h, w).permute(0, 3, 4, 1, 2).reshape(n, -1, K))
        shape = ops.conv_shape(n, h, w, cin, cout, 3, 1, out_ld=cout)
        wf, _ = ops.pack_weights(shape, wt.to(dev()), want_dgrad=False)
        ops.conv_fwd_ex(shape, nhwc(x), wf, out[:, off:], shift=bias.to(dev()), out_f32=True, out_image_stride=tot * K)
        off += h * w * A
    torch.cuda.synchronize()
    ref = torch.cat(refs, 1)
    assert float((out.cpu() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


def test_stem_7x7_as_im2col_gemm():
    from object_detectors_amd import ops
    n, h, w = 2, 64, 96
    g = torch.Generator().manual_seed(0)
    img = torch.rand((n, 3, h, w), generator=g)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    wt = rnd((64, 3, 7, 7), 2, 0.1)
    xn = ((img - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1))
    ref = F.conv2d(xn.bfloat16().float(), wt, stride=2, padding=3)
    col = ops.im2col_nchw(img.to(dev()), 7, 2, 3, 160, mean=mean.to(dev()), inv_std=(1.0 / std).to(dev()))
    assert col.shape == (n, 32, 48, 160)
    wm = torch.zeros((64, 160), device=dev())
    wm[:, :147] = wt.permute(0, 2, 3, 1).reshape(64, 147).to(dev())       # k = (kh*7+kw)*3 + c
    shape = ops.conv_shape(n, 32, 48, 160, 64, 1, 1)
    wf, _ = ops.pack_weights(shape, wm.view(64, 1, 1, 160), want_dgrad=False, ohwi=True)
    y = torch.zeros((n, 32, 48, 64), device=dev(), dtype=torch.bfloat16)
    ops.conv_fwd_ex(shape, col, wf, y, relu=True)
    torch.cuda.synchronize()
    ref = torch.relu(ref)
    assert float((to_nchw(y) - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    # max-pool 3x3/2 pad 1 on the result (resnet.py:176), odd and even sizes
    for hh, ww in ((32, 48), (25, 31)):
        a = y[:, :hh, :ww].contiguous()
        mp = ops.maxpool3x3s2(a)
        refp = F.max_pool2d(a.float().permute(0, 3, 1, 2), 3, 2, 1)
        assert torch.equal(mp.float().permute(0, 3, 1, 2), refp)


def test_relu_affine_bwd():
    from object_detectors_amd import ops
    n, h, w, c = 2, 9, 7, 64
    a = torch.relu(rnd((n, h, w, c), 1)).to(dev()).bfloat16()
    g1, g2 = rnd((n, h, w, c), 2).to(dev()).bfloat16(), rnd((n, h, w, c), 3).to(dev()).bfloat16()
    sc = (torch.rand(c, generator=torch.Generator().manual_seed(4)) + 0.5).to(dev())
    dz, gm = ops.relu_affine_bwd(g1, a, scale=sc, g2=g2, want_gm=True)
    ref_gm = ((g1.float() + g2.float()).bfloat16().float()) * (a.float() > 0)
    assert torch.equal(gm.float(), ref_gm.bfloat16().float())
    torch.testing.assert_close(dz.float(), (ref_gm * sc).bfloat16().float(), rtol=1e-2, atol=1e-6)
    dz2 = ops.relu_affine_bwd(g1, a, relu=False)
    assert torch.equal(dz2, g1)


@pytest.mark.parametrize("src,dst", [((13, 13), (25, 25)), ((25, 25), (50, 50)), ((7, 10), (13, 19))])
def test_fpn_nearest_upsample_add(src, dst):
    from object_detectors_amd import ops
    n, c = 2, 64
    x = rnd((n, c, *src), 1)
    lat = rnd((n, c, *dst), 2)
    xr = x.clone().requires_grad_(True)
    ref = lat + F.interpolate(xr, size=dst, mode="nearest")
    out = ops.upsample_nearest_add(nhwc(x), nhwc(lat), dst)
    torch.cuda.synchronize()
    assert float((to_nchw(out) - ref.detach()).abs().max()) <= 1e-2 * float(ref.abs().max())
    g = rnd((n, c, *dst), 3)
    ref.backward(g)
    acc = rnd((n, c, *src), 4)
    gx = ops.upsample_nearest_bwd(nhwc(g), src, accumulate=nhwc(acc))
    torch.cuda.synchronize()
    want = xr.grad + acc
    assert float((to_nchw(gx) - want).abs().max()) <= 1e-2 * float(want.abs().max())


def test_retina_loss_batched_vs_oracle(golden):
    """mi355det_retina_loss (whole batch, 3 launches) == RetinaNetHead.compute_loss restated in oracle/tv_oracle.py."""
    from object_detectors_amd import ops
    from object_detectors_amd.tvision._utils import Matcher
    from tests.test_oracle_tv import build_anchors
    g = golden("g5_7_tvision")
    anchors = build_anchors(g, "retina_small")
    N, K, b = anchors.shape[0], 91, 3
    gts = []
    for i in range(b):
        m = [4, 1, 6][i]
        side = detrand.uniform(50 + i, (m, 2), 16, 90)
        tl = detrand.uniform(60 + i, (m, 2), 0, 1) * (np.array([160, 128], np.float32) - side)
        gts.append((np.concatenate([tl, tl + side], 1).astype(np.float32), detrand.randint(70 + i, (m,), 1, K)))
    logits = detrand.uniform(80, (b, N, K), -6, 2)
    reg = detrand.uniform(81, (b, N, 4), -1, 1)
    tfidf = detrand.uniform(82, (K,), 0.5, 2.0)
    mt = Matcher(0.5, 0.4, True)
    matched = torch.stack([mt.match_boxes(T(bx), T(anchors)) for bx, _ in gts])
    gt_boxes = T(np.concatenate([bx for bx, _ in gts]))
    gt_labels = T(np.concatenate([lb for _, lb in gts]).astype(np.int64))
    offs = T(np.cumsum([0] + [len(lb) for _, lb in gts]).astype(np.int32))
    for tf in (None, tfidf):
        cl, rl, mis, (gc, gr) = tv.retinanet_loss(logits, reg, anchors, gts, tfidf=tf)
        losses, nfg, glog, greg = ops.retina_loss(T(logits), T(reg), T(anchors), matched, gt_boxes, gt_labels, offs,
                                                  class_scale=None if tf is None else T(tf))
        torch.cuda.synchronize()
        assert np.array_equal(nfg.cpu().numpy(), np.array([(m >= 0).sum() for m in mis], np.float32))
        np.testing.assert_allclose(losses.cpu().numpy(), [cl, rl], rtol=2e-4)
        np.testing.assert_allclose(glog.cpu().numpy(), gc, rtol=2e-3, atol=1e-7)
        np.testing.assert_allclose(greg.cpu().numpy(), gr, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("K", [91, 36, 5, 3])
def test_retina_loss_level_gradients_equal_the_cast_of_the_fp32_gradient(golden, K):
    """mi355det_retina_loss_lv writes the class gradient as bf16 straight into the per-level NHWC buffers of the cls_logits backward
    (channel a*K + c of a pixel): bit-identical to casting the fp32 gradient of mi355det_retina_loss row by row (what the engine did
    before), same losses, padding channels untouched.  K = 91 (rows straddle the 4-element vector groups) and K = 36 (K % 4 == 0)."""
    from object_detectors_amd import ops
    from object_detectors_amd.tvision._utils import Matcher
    from tests.test_oracle_tv import build_anchors
    g = golden("g5_7_tvision")
    anchors = build_anchors(g, "retina_small")
    N, b, A = anchors.shape[0], 3, 9
    assert N % A == 0
    pix = N // A
    parts = [pix - pix // 3 - pix // 7, pix // 3, pix // 7]            # three "levels" of h = 1
    gts = []
    for i in range(b):
        m = [4, 1, 6][i]
        side = detrand.uniform(150 + i, (m, 2), 16, 90)
        tl = detrand.uniform(160 + i, (m, 2), 0, 1) * (np.array([160, 128], np.float32) - side)
        gts.append((np.concatenate([tl, tl + side], 1).astype(np.float32), detrand.randint(170 + i, (m,), 1, K)))
    logits = T(detrand.uniform(180, (b, N, K), -6, 2))
    reg = T(detrand.uniform(181, (b, N, 4), -1, 1))
    mt = Matcher(0.5, 0.4, True)
    matched = torch.stack([mt.match_boxes(T(bx), T(anchors)) for bx, _ in gts])
    gt_boxes = T(np.concatenate([bx for bx, _ in gts]))
    gt_labels = T(np.concatenate([lb for _, lb in gts]).astype(np.int64))
    offs = T(np.cumsum([0] + [len(lb) for _, lb in gts]).astype(np.int32))
    losses, nfg, glog, greg = ops.retina_loss(logits, reg, T(anchors), matched, gt_boxes, gt_labels, offs)
    ld = ops.pad_to(A * K, 64)
    levels = [torch.full((b, 1, p, ld), 3.0, device=logits.device, dtype=torch.bfloat16) for p in parts]
    losses2, nfg2, none, greg2 = ops.retina_loss(logits, reg, T(anchors), matched, gt_boxes, gt_labels, offs, cls_levels=levels, anchors_per_pixel=A)
    torch.cuda.synchronize()
    assert none is None
    assert torch.equal(nfg, nfg2) and torch.equal(greg, greg2)
    np.testing.assert_allclose(losses2.cpu().numpy(), losses.cpu().numpy(), rtol=1e-5)      # the loss word is an atomic sum: order may differ
    row0 = 0
    for p, buf in zip(parts, levels):
        want = glog[:, row0:row0 + p * A, :].reshape(b, 1, p, A * K).to(torch.bfloat16)
        assert torch.equal(buf[..., :A * K], want)
        assert bool((buf[..., A * K:] == 3.0).all())                                         # padding channels are not written
        row0 += p * A
    with pytest.raises(ValueError):
        ops.retina_loss(logits, reg, T(anchors), matched, gt_boxes, gt_labels, offs, cls_levels=levels[:2], anchors_per_pixel=A)   # rows do not add up


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 32, 32), (3, 128, 64)])
@pytest.mark.parametrize("normalize", [False, True])
def test_resnet_stem_direct_convolution(shape, normalize):
    """mi355det_resnet_stem_fwd (7x7 / 2 / pad 3, 3 -> 64, FrozenBN affine, ReLU in one kernel, no im2col matrix) against PyTorch fp32 on
    the same bf16-rounded operands, with the ImageNet normalisation folded in (padding is zero AFTER normalisation, transform.py:120-124,
    224-240) or not; image borders on all four sides; sizes that are not multiples of 32 are refused."""
    import ctypes as C
    from object_detectors_amd._lib import check, lib, ptr, stream_ptr
    n, h, w = shape
    g = torch.Generator().manual_seed(5 + h)
    img = torch.rand((n, 3, h, w), generator=g)
    wt = (torch.randn((64, 3, 7, 7), generator=g) * (2.0 / 147) ** 0.5).bfloat16().float()
    scale, shift = 1.0 + 0.3 * torch.randn(64, generator=g), 0.2 * torch.randn(64, generator=g)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    x = ((img - mean[None, :, None, None]) / std[None, :, None, None]) if normalize else img
    x = x.bfloat16().float()                                              # the kernel stages the (normalised) image as bf16
    want = torch.relu(F.conv2d(x, wt, stride=2, padding=3) * scale[None, :, None, None] + shift[None, :, None, None])
    wp = torch.zeros((64, 160), dtype=torch.bfloat16, device="cuda")
    wp[:, :147] = wt.permute(0, 2, 3, 1).reshape(64, 147).cuda().bfloat16()            # k = (kh*7+kw)*3 + c
    out = torch.full((n, h // 2, w // 2, 64), 7.0, dtype=torch.bfloat16, device="cuda")
    L = lib()
    md, isd = (mean.cuda(), (1.0 / std).cuda()) if normalize else (None, None)
    imgd, scd, shd = img.cuda(), scale.cuda(), shift.cuda()                # keep the device tensors alive across the launch
    check(L.mi355det_resnet_stem_fwd(ptr(imgd), ptr(md), ptr(isd), ptr(wp), ptr(scd), ptr(shd), 1, ptr(out), 64, n, h, w, stream_ptr()), "resnet_stem_fwd")
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert float((got - want).abs().max()) <= 1.5e-2 * float(want.abs().max())
    assert L.mi355det_resnet_stem_fwd(ptr(imgd), None, None, ptr(wp), None, None, 1, ptr(out), 64, n, h + 8, w, stream_ptr()) == -1
