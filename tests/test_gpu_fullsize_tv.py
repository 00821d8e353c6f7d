"""Full-size (800 px) checks of the kernels that BASELINE configs 3-5 (RetinaNet-R50 bs 16, Faster R-CNN bs 2 per GPU, RetinaNet-R101 with the
1204-class LVIS head bs 8) launch at their real geometry - VERDICT r3 item 5: under `-m gpu` those configs ran at 128 px, so the fp32-head
igemm8 tile with Cout 11 008, the split-K data gradient of the small pyramid levels, the direct ResNet stem at 800 x 800 and the two-stage
column sums on 160 000 rows met their launch geometry only in benches.  The oracles are too slow at this size: the checks are
size-independent properties (linearity, agreement of two independent routes to the same tensor) plus spot comparisons of small crops against
torch fp32 on the same device.  Shapes: tvision/retinanet.py:84-95,179-185, tvision/rpn.py:230-280, utilities/resnet.py:173-176."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def gen(seed):
    return torch.Generator(device=dev()).manual_seed(seed)


def randn(shape, seed, scale=1.0, dtype=torch.bfloat16):
    return (torch.randn(shape, device=dev(), generator=gen(seed)) * scale).to(dtype)


def test_lvis_cls_logits_forward_100x100_level_batch8():
    """cls_logits of the 1204-class head on the 100 x 100 pyramid level at batch 8 (retinanet.py:84-95: 256 -> 9 * 1204 = 10 836 channels,
    3x3, bias, fp32 written straight into the level-concatenated [N, sum HWA, K] tensor): linearity in the input, and a 10-row strip against
    torch's fp32 convolution of the same bf16 operands."""
    from object_detectors_amd import ops
    n, hw, cin, A, K = 8, 100, 256, 9, 1204
    cout = A * K
    wt = torch.randn(cout, cin, 3, 3, device=dev(), generator=gen(1)) * 0.01
    bias = torch.randn(cout, device=dev(), generator=gen(2)) * 0.1
    shape = ops.conv_shape(n, hw, hw, cin, cout, 3, 1, out_ld=cout)
    wf, _ = ops.pack_weights(shape, wt, want_dgrad=False)
    x1 = randn((n, hw, hw, cin), 3)
    x3 = (x1.float() * 3).bfloat16()
    x1 = torch.where(x3.float() == x1.float() * 3, x1, torch.zeros_like(x1))      # keep the values whose triple is a bf16 number
    x3 = (x1.float() * 3).bfloat16()
    tot = hw * hw * A + 64 * A                     # the level sits inside a longer concatenated tensor
    outs = []
    for x in (x1, x3):
        out = torch.full((n, tot, K), -7.0, device=dev())
        ops.conv_fwd_ex(shape, x, wf, out[:, 32 * A:], shift=bias, out_f32=True, out_image_stride=tot * K)
        outs.append(out)
    torch.cuda.synchronize()
    y1, y3 = outs
    assert float(y1[:, :32 * A].min()) == -7.0 == float(y1[:, :32 * A].max()) and float(y1[:, 32 * A + hw * hw * A:].max()) == -7.0      # nothing outside the level
    lv1 = y1[:, 32 * A:32 * A + hw * hw * A].view(n, hw * hw, cout)
    lv3 = y3[:, 32 * A:32 * A + hw * hw * A].view(n, hw * hw, cout)
    scale = float(lv3.abs().max())
    assert float(((lv3 - bias) - 3 * (lv1 - bias)).abs().max()) <= 1e-4 * scale
    # rows 0..9 of image 5 against torch (fp32 accumulation in both: 2e-3 of max covers the summation order)
    xs = x1[5, :12].float().permute(2, 0, 1).unsqueeze(0)
    ref = F.conv2d(xs, wt.bfloat16().float(), bias, padding=1)[0, :, :10].permute(1, 2, 0).reshape(10 * hw, cout)
    got = lv1[5, :10 * hw]
    assert float((got - ref).abs().max()) <= 2e-3 * float(ref.abs().max())


@pytest.mark.parametrize("hw", [13, 7])
def test_lvis_cls_logits_data_gradient_split_k_small_levels(hw):
    """The few-pixel / deep-reduction data gradient of cls_logits on the 13 x 13 and 7 x 7 levels at batch 8 (K = 9 * 10 880 per output
    element, 1 352 / 392 pixels): the split-K form (mi355det_conv_dgrad_ws: channel ranges -> fp32 partial tiles -> one rounding) against
    the un-split kernel and against torch autograd in fp32 on the same operands."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import check, lib, ptr, stream_ptr
    n, cin, creal, cout = 8, 256, 9 * 1204, 10880          # the engine pads the head's channel count to a multiple of 64 (zero weight rows)
    wt = torch.zeros(cout, cin, 3, 3, device=dev())
    wt[:creal] = torch.randn(creal, cin, 3, 3, device=dev(), generator=gen(4)) * 0.01
    shape = ops.conv_shape(n, hw, hw, cin, cout, 3, 1)
    _, wd = ops.pack_weights(shape, wt)
    dy = torch.zeros((n, hw, hw, cout), device=dev(), dtype=torch.bfloat16)
    dy[..., :creal] = randn((n, hw, hw, creal), 5, 0.05)
    L = lib()
    nbytes = L.mi355det_conv_dgrad_workspace(C.byref(shape))
    assert nbytes > 0, "this shape is meant to take the split-K route"
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev())
    dx_split = torch.full((n, hw, hw, cin), 3.0, device=dev(), dtype=torch.bfloat16)
    dx_plain = torch.full((n, hw, hw, cin), 3.0, device=dev(), dtype=torch.bfloat16)
    check(L.mi355det_conv_dgrad_ws(C.byref(shape), ptr(dy), ptr(wd), ptr(dx_split), None, 0, ptr(ws), nbytes, stream_ptr()), "conv_dgrad_ws")
    check(L.mi355det_conv_dgrad(C.byref(shape), ptr(dy), ptr(wd), ptr(dx_plain), None, 0, stream_ptr()), "conv_dgrad")
    xr = torch.zeros((n, cin, hw, hw), device=dev(), requires_grad=True)
    F.conv2d(xr, wt.bfloat16().float(), padding=1).backward(dy.float().permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1)
    m = float(ref.abs().max())
    assert float((dx_split.float() - ref).abs().max()) <= 1e-2 * m
    assert float((dx_plain.float() - ref).abs().max()) <= 1e-2 * m
    assert float((dx_split.float() - dx_plain.float()).abs().max()) <= 1e-2 * m


def test_resnet_stem_direct_16_images_800px_against_the_im2col_route():
    """utilities/resnet.py:173-176 at RetinaNet's batch 16 / 800 x 800: the direct 7x7/2 kernel (mi355det_resnet_stem_fwd) against the im2col +
    GEMM route it replaced (two independent routes to the same tensor) and a 24-row crop against torch fp32."""
    from object_detectors_amd import ops
    from object_detectors_amd._lib import check, lib, ptr, stream_ptr
    n, h, w = 16, 800, 800
    img = torch.rand((n, 3, h, w), device=dev(), generator=gen(6))
    wt = (torch.randn((64, 3, 7, 7), device=dev(), generator=gen(7)) * (2.0 / 147) ** 0.5).bfloat16().float()
    scale = 1.0 + 0.3 * torch.randn(64, device=dev(), generator=gen(8))
    shift = 0.2 * torch.randn(64, device=dev(), generator=gen(9))
    mean, std = torch.tensor([0.485, 0.456, 0.406], device=dev()), torch.tensor([0.229, 0.224, 0.225], device=dev())
    wp = torch.zeros((64, 160), dtype=torch.bfloat16, device=dev())
    wp[:, :147] = wt.permute(0, 2, 3, 1).reshape(64, 147).bfloat16()
    out = torch.full((n, h // 2, w // 2, 64), 7.0, dtype=torch.bfloat16, device=dev())
    isd = 1.0 / std
    check(lib().mi355det_resnet_stem_fwd(ptr(img), ptr(mean), ptr(isd), ptr(wp), ptr(scale), ptr(shift), 1, ptr(out), 64, n, h, w, stream_ptr()), "resnet_stem_fwd")
    col = ops.im2col_nchw(img, 7, 2, 3, 160, mean=mean, inv_std=isd)
    shape = ops.conv_shape(n, h // 2, w // 2, 160, 64, 1, 1)
    wm = torch.zeros((64, 160), device=dev())
    wm[:, :147] = wt.permute(0, 2, 3, 1).reshape(64, 147)
    wf, _ = ops.pack_weights(shape, wm.view(64, 1, 1, 160), want_dgrad=False, ohwi=True)
    y = torch.zeros((n, h // 2, w // 2, 64), device=dev(), dtype=torch.bfloat16)
    ops.conv_fwd_ex(shape, col, wf, y, scale=scale, shift=shift, relu=True)
    torch.cuda.synchronize()
    m = float(y.float().abs().max())
    assert m > 0 and float((out.float() - y.float()).abs().max()) <= 1.6e-2 * m          # two bf16 roundings of differently ordered sums
    xn = ((img[11:12, :, :48 + 6] - mean[None, :, None, None]) / std[None, :, None, None]).bfloat16().float()
    ref = torch.relu(F.conv2d(xn, wt, stride=2, padding=3) * scale[None, :, None, None] + shift[None, :, None, None])[0, :, :24].permute(1, 2, 0)
    assert float((out[11, :24].float() - ref).abs().max()) <= 1.5e-2 * float(ref.abs().max())


def test_tower_bias_gradient_column_sums_160k_rows():
    """Bias gradient of a RetinaNet head-tower convolution (256 -> 256, 3x3) on the 100 x 100 level at batch 16: the two-stage fixed-order
    column sums over 160 000 gradient rows (colsum8p + fold) against a float64 sum, bit-reproducible, and one tap of the weight gradient of the
    same launch against a float64 contraction."""
    from object_detectors_amd import ops
    n, hw, c = 16, 100, 256
    shape = ops.conv_shape(n, hw, hw, c, c, 3, 1)
    x = randn((n, hw, hw, c), 10)
    gy = randn((n, hw, hw, c), 11, 0.1)
    runs = []
    for _ in range(2):
        dw = torch.zeros(c, 9 * c, device=dev())
        db = torch.zeros(c, device=dev())
        ops.conv_wgrad(shape, x, gy, dw, dbias=db)
        runs.append((dw, db))
    torch.cuda.synchronize()
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][0], runs[1][0])
    ref = gy.view(-1, c).double().sum(0)
    assert torch.allclose(runs[0][1].double(), ref, rtol=1e-4, atol=1e-4 * float(gy.view(-1, c).double().abs().sum(0).max()))
    refw = torch.einsum("pc,pk->ck", gy.view(-1, c)[:, :8].double(), x.view(-1, c).double())
    got = runs[0][0].view(c, 9, c)[:8, 4, :].double()
    assert torch.allclose(got, refw, rtol=2e-3, atol=2e-3 * float(refw.abs().max()))


def test_fasterrcnn_step_800px_batch2_fused_routes_equal_the_composed_routes():
    """One Faster R-CNN-R50-FPN training step at the BASELINE per-GPU batch (2 images, 800 x 800, 159 882 anchors per image, 2000 / 1000
    proposals: tvision/rpn.py:230-280, roi_heads.py:783-848), repeated: the one-call proposal filter (mi355det_rpn_proposals) and the fused RoI
    sampling (mi355det_roi_match / _roi_sample) against the composed routes they replace, and the FIRST call of a fresh model against the
    later ones.  Round 4 found with this test that the backward of the tvision engines freed gradient buffers whose raw pointers stayed in
    the call list: the box head's weight packs, allocated during the first call, were overwritten by every later backward (loss_classifier
    12.2 in call 1, ln 91 = 4.5 from a zeroed fc7 afterwards).  Same weights, same seed: every call must now report the same losses - up to
    the order of equal objectness scores at the top-k boundary, which moves a sample or two (1e-4 relative on the classifier loss)."""
    from object_detectors_amd.tvision import frcnn
    torch.manual_seed(0)
    m = frcnn.fasterrcnn_resnet50_fpn(num_classes=91, device=dev())
    x = torch.rand((2, 3, 800, 800), device=dev(), generator=gen(12))
    t = [{"boxes": torch.tensor([[100.0, 120.0, 400.0, 500.0], [300.0, 250.0, 720.0, 640.0], [50.0, 600.0, 180.0, 780.0]], device=dev()),
          "labels": torch.tensor([5, 17, 44], device=dev())},
         {"boxes": torch.tensor([[10.0, 10.0, 790.0, 790.0], [350.0, 360.0, 420.0, 450.0]], device=dev()), "labels": torch.tensor([1, 90], device=dev())}]
    m.train()
    m.engine.plan(2, 800, 800, True)                      # the plan build draws random tuning inputs: keep it out of the seeded region
    xf = torch.randn(512, 256, 7, 7, device=dev(), generator=gen(13))
    res, probes = {}, []
    saved = (frcnn._RPN_FUSED, frcnn._ROI_FUSED, frcnn._RPN_LOSS_FUSED)
    try:
        for tag, fused in (("first", True), ("fused", True), ("composed", False), ("fused2", True)):
            frcnn._RPN_FUSED = frcnn._ROI_FUSED = frcnn._RPN_LOSS_FUSED = fused
            for p in m.head_parameters():
                p.grad = None
            torch.manual_seed(1234)                       # the sampler's torch.randperm draws (tvision/_utils.py:25-76)
            losses = m(x, t)
            torch.cuda.synchronize()
            res[tag] = ({k: float(v) for k, v in losses.items()}, m.engine.flat_g.clone())
            with torch.no_grad():
                probes.append(m.box_head(xf).clone())     # a fixed input through the box head after every backward
    finally:
        frcnn._RPN_FUSED, frcnn._ROI_FUSED, frcnn._RPN_LOSS_FUSED = saved
    for p in probes[1:]:
        assert torch.equal(p, probes[0])                  # nothing the backward touches belongs to anyone else
    l0, g0 = res["first"]
    assert all(torch.isfinite(torch.tensor(list(l0.values()))))
    assert float(probes[0].float().abs().mean()) > 1e-3   # the box head is alive
    for tag in ("fused", "composed", "fused2"):
        l, g = res[tag]
        for k in ("loss_classifier", "loss_box_reg"):
            assert abs(l[k] - l0[k]) <= 2e-3 * abs(l0[k]) + 1e-6, (tag, k, l, l0)
        for k in ("loss_objectness", "loss_rpn_box_reg"):
            assert abs(l[k] - l0[k]) <= 1e-5 * abs(l0[k]) + 1e-7, (tag, k, l, l0)
        rel = float((g - g0).norm() / g0.norm())
        assert rel < 2e-2, (tag, rel)
