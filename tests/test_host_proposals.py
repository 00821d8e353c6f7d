"""Host-side logic of the round-3 one-call routes (no GPU): limits that send a batch to the composed route, the clip-limit cache, and the
no-CPU-fallback rule (ops raise on CPU tensors instead of computing something)."""
import pytest
import torch


def test_fused_sampling_limits_fall_back_to_the_composed_route():
    from object_detectors_amd.tvision.roi_heads import RoIHeadTargets
    tg = RoIHeadTargets()
    t = lambda m: {"boxes": torch.zeros((m, 4)), "labels": torch.zeros((m,), dtype=torch.int64)}
    assert tg.fused_ok(4, 2000, [t(3), t(1), t(7), t(2)])
    assert not tg.fused_ok(4, 2000, [t(3), t(0), t(7), t(2)])             # an image without ground truth: the composed route raises like the reference
    assert not tg.fused_ok(2, 2000, [t(3), t(1025)])                      # more than 1024 ground-truth boxes in an image
    assert not tg.fused_ok(2, 8000, [t(3), t(300)])                       # more than 8192 candidates (proposals + ground truth)
    assert not tg.fused_ok(65, 100, [t(1)] * 65)                          # more than 64 images
    big = RoIHeadTargets(batch_size_per_image=2048)
    assert not big.fused_ok(2, 2000, [t(3), t(1)])                        # more samples per image than the kernel's sort holds


def test_clip_limits_cache_layout():
    from object_detectors_amd.tvision.postprocess import _clip_limits
    lim = _clip_limits([(480, 640), (512, 512)], torch.device("cpu"), torch.float32)
    assert lim.shape == (2, 1, 4)
    assert lim.reshape(2, 4).tolist() == [[640.0, 480.0, 640.0, 480.0], [512.0, 512.0, 512.0, 512.0]]
    assert _clip_limits([(480, 640), (512, 512)], torch.device("cpu"), torch.float32) is lim


@pytest.mark.parametrize("call", ["rpn_proposals", "retina_detections", "roi_detections", "rpn_loss", "topk_segments"])
def test_one_call_routes_have_no_cpu_fallback(call):
    from object_detectors_amd import ops
    z = torch.zeros
    with pytest.raises(ValueError, match="CUDA/HIP"):
        if call == "rpn_proposals":
            ops.rpn_proposals(z((1, 8)), z((1, 8, 4)), z((8, 4)), z((1, 4)), [8], 4, 4, 0.7)
        elif call == "retina_detections":
            ops.retina_detections([z((1, 8, 3))], [z((1, 8, 4))], [z((8, 4))], z((1, 4)), 0.0, 4, 0.5, 4)
        elif call == "roi_detections":
            ops.roi_detections(z((1, 8, 3)), z((1, 8, 12)), z((1, 8, 4)), z((1, 4)), 0.05, 16, (10.0, 10.0, 5.0, 5.0), 0.5, 4)
        elif call == "rpn_loss":
            ops.rpn_loss(z((8, 1)), z((8, 4)), z(8), z((8, 4)), torch.zeros(1, dtype=torch.int64), torch.zeros(2, dtype=torch.int64))
        else:
            ops.topk_segments(z((1, 8)), [8], 4)
