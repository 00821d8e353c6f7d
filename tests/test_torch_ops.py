"""`torch.ops.mi355det.*` (object_detectors_amd/torch_ops.py): the detection kernels registered with torch.library.  On the CPU: the
operators exist with the expected schemas, fake-tensor shape inference works (what torch.compile / export need), and there is no CPU
implementation to fall back to.  On the GPU (-m gpu): they give what the ctypes wrappers give and are differentiable."""
import pytest
import torch

import object_detectors_amd.torch_ops as tops


def test_operators_are_registered_with_schemas():
    for name in tops.OPS:
        assert hasattr(torch.ops.mi355det, name), name
    s = str(torch.ops.mi355det.nms.default._schema)
    assert "Tensor boxes" in s and "float iou_threshold" in s and "-> Tensor" in s
    s = str(torch.ops.mi355det.roi_align.default._schema)
    assert "pooled_height" in s and "bool aligned" in s


def test_fake_tensor_shape_inference():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        b1, b2 = torch.empty(5, 4, device="cuda"), torch.empty(7, 4, device="cuda")
        assert torch.ops.mi355det.box_iou(b1, b2).shape == (5, 7)
        x = torch.empty(2, 11, 13, device="cuda")
        assert torch.ops.mi355det.sigmoid_focal_loss_sum(x, x, 0.25, 2.0).shape == ()
        feat, rois = torch.empty(2, 16, 20, 24, device="cuda"), torch.empty(9, 5, device="cuda")
        assert torch.ops.mi355det.roi_align(feat, rois, 0.25, 7, 7, 2, False).shape == (9, 16, 7, 7)
        assert torch.ops.mi355det.bbox_iou(torch.empty(3, 1, 4, device="cuda"), torch.empty(1, 8, 4, device="cuda"), 1, True).shape == (3, 8)


def test_no_cpu_implementation():
    with pytest.raises(NotImplementedError):
        torch.ops.mi355det.box_iou(torch.zeros(2, 4), torch.zeros(3, 4))


@pytest.mark.gpu
def test_custom_ops_match_the_wrappers_and_differentiate():
    from object_detectors_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    ctr, wh = torch.rand((300, 2), generator=g) * 200, torch.rand((300, 2), generator=g) * 50 + 2
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).to(dev)
    scores = torch.rand(300, generator=g).to(dev)
    idxs = torch.randint(0, 4, (300,), generator=g).to(dev)
    assert torch.equal(torch.ops.mi355det.box_iou(boxes[:50], boxes[50:120]), ops.box_iou(boxes[:50], boxes[50:120]))
    assert torch.equal(torch.ops.mi355det.nms(boxes, scores, 0.5), ops.nms(boxes, scores, 0.5))
    assert torch.equal(torch.ops.mi355det.batched_nms(boxes, scores, idxs, 0.5), ops.nms(boxes, scores, 0.5, idxs=idxs))
    # focal loss: value and gradient against the fused forward + gradient call
    x = (torch.randn((64, 91), generator=g) * 2).to(dev).requires_grad_(True)
    t = (torch.rand((64, 91), generator=g) > 0.97).float().to(dev)
    loss = torch.ops.mi355det.sigmoid_focal_loss_sum(x, t, 0.25, 2.0)
    (loss * 3.0).backward()
    l2, g2 = ops.sigmoid_focal_loss_sum(x.detach(), t, 0.25, 2.0)
    torch.testing.assert_close(loss.detach(), l2.reshape(()), rtol=1e-6, atol=0)
    torch.testing.assert_close(x.grad, 3.0 * g2.reshape(x.shape), rtol=1e-6, atol=0)
    # RoIAlign: forward and feature gradient against the multi-level wrapper
    feat = torch.randn((2, 8, 20, 24), generator=g).to(dev).requires_grad_(True)
    rois = torch.tensor([[0, 4.0, 4.0, 60.0, 50.0], [1, 10.0, 8.0, 90.0, 70.0], [0, 0.0, 0.0, 95.0, 79.0]], device=dev)
    out = torch.ops.mi355det.roi_align(feat, rois, 0.25, 7, 7, 2, False)
    want = ops.roi_align_multi([feat.detach()], rois, (7, 7), [0.25], 2, False, 0, 0)
    assert torch.equal(out.detach(), want)
    go = torch.randn(out.shape, generator=g).to(dev)
    out.backward(go)
    want_g = ops.roi_align_multi([feat.detach()], rois, (7, 7), [0.25], 2, False, 0, 0, grad_out=go)[0]
    torch.testing.assert_close(feat.grad, want_g, rtol=1e-5, atol=1e-6)                  # the scatter-add ends in fp32 atomics: order may differ
