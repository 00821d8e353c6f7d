"""Input-side transform on the GPU (tvision/transform.py mirror, mi355det_resize_bilinear / resize_boxes) against the REFERENCE's
GeneralizedRCNNTransform and F.interpolate outputs (fixture g14) and the oracle at full size."""
import numpy as np
import pytest

from oracle import detrand
from oracle import tv_oracle as tv
from tests.test_oracle_transform import fixture_images

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def test_generalized_rcnn_transform_eval_train_and_postprocess(golden):
    from object_detectors_amd.tvision.transform import GeneralizedRCNNTransform
    g = golden("g14_transform")
    imgs = fixture_images(g)
    t = GeneralizedRCNNTransform(800, 1333, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    t.eval()
    targets = [{"boxes": T(g[f"boxes{i}"])} for i in range(len(imgs))]
    il, tg = t([T(i) for i in imgs], targets)
    assert list(il.tensors.shape) == g["eval_batch_shape"].tolist()            # 800 x 1344: the 480x730 image resizes to 800 x 1216, padded to /32
    assert [list(s) for s in il.image_sizes] == g["eval_image_sizes"].tolist()
    out = il.tensors.cpu().numpy()
    np.testing.assert_allclose(out[:, :, ::13, ::17], g["eval_batch_sample"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(out.astype(np.float64).sum((1, 2, 3)), g["eval_batch_sum"], rtol=1e-6)
    want, _s, _b = tv.rcnn_transform(imgs)                                      # every element, not only the stored stride
    np.testing.assert_allclose(out, want, rtol=1e-5, atol=1e-4)
    assert targets[0]["boxes"].data_ptr() != tg[0]["boxes"].data_ptr()          # the caller's dicts are not modified
    for i in range(len(imgs)):
        np.testing.assert_allclose(tg[i]["boxes"].cpu().numpy(), g[f"eval_boxes{i}"], rtol=1e-6)
    back = t.postprocess([{"boxes": tg[i]["boxes"]} for i in range(len(imgs))], il.image_sizes, [tuple(int(v) for v in s) for s in g["shapes"]])
    for i in range(len(imgs)):
        np.testing.assert_allclose(back[i]["boxes"].cpu().numpy(), g[f"post_boxes{i}"], rtol=1e-6)
    # training mode: the min size is drawn with torch's global RNG exactly as the reference's torch_choice does
    t2 = GeneralizedRCNNTransform((640, 672, 704, 736, 768, 800), 1333, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    t2.train()
    torch.manual_seed(77)
    il2, _ = t2([T(i) for i in imgs[:3]], None)
    assert [list(s) for s in il2.image_sizes] == g["train_image_sizes"].tolist() and list(il2.tensors.shape) == g["train_batch_shape"].tolist()
    np.testing.assert_allclose(il2.tensors.cpu().numpy()[:, :, ::13, ::17], g["train_batch_sample"], rtol=1e-5, atol=1e-4)
    with pytest.raises(ValueError):
        t([T(imgs[0])[None]])
    with pytest.raises(TypeError):
        t([(T(imgs[0]) * 255).to(torch.uint8)])


def test_yolo_multiscale_interpolate(golden):
    from object_detectors_amd.tvision.transform import interpolate_bilinear
    g = golden("g14_transform")
    x = detrand.uniform(8300, (2, 3, 416, 416), -2.0, 2.0)
    for size in (320, 608):
        y = interpolate_bilinear(T(x), size).cpu().numpy()
        assert y.shape == (2, 3, size, size)
        np.testing.assert_allclose(y[:, :, ::7, ::11], g[f"yolo_ms_{size}_sample"], rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(y.astype(np.float64).sum((1, 2, 3)), g[f"yolo_ms_{size}_sum"], rtol=1e-6, atol=1e-2)


def test_resize_identity_and_full_size_properties():
    """Size-independent properties at the BASELINE size (32 x 3 x 640 x 640): same-size resize is the identity, a constant image stays
    constant, and resizing commutes with adding a constant (linearity)."""
    from object_detectors_amd.tvision.transform import interpolate_bilinear
    x = torch.randn(32, 3, 640, 640, device="cuda:0")
    assert torch.equal(interpolate_bilinear(x, 640), x)
    y = interpolate_bilinear(x, 960)
    assert tuple(y.shape) == (32, 3, 960, 960)
    c = interpolate_bilinear(torch.full((1, 3, 640, 640), 2.5, device="cuda:0"), 352)
    assert float((c - 2.5).abs().max()) < 1e-6
    y2 = interpolate_bilinear(x + 1.0, 960)
    assert float((y2 - (y + 1.0)).abs().max()) < 1e-5
    assert float(y.min()) >= float(x.min()) - 1e-5 and float(y.max()) <= float(x.max()) + 1e-5      # convex combinations


def test_retinanet_mirror_runs_the_transform_on_image_lists():
    """A list of images of DIFFERENT sizes through the RetinaNet mirror: transform (normalise, resize, pad) -> engine without its fused
    normalisation -> detections mapped back to each input frame; the head outputs equal the oracle network on the oracle-transformed batch."""
    from object_detectors_amd.tvision.retinanet import retinanet_resnet50_fpn
    from oracle import retina_oracle as ro
    sd = ro.det_state(7000)
    m = retinanet_resnet50_fpn(num_classes=91, device=torch.device("cuda:0"), min_size=160, max_size=256)
    m.load_state_dict(sd)
    m.eval()
    imgs = [detrand.uniform(8400, (3, 120, 150), 0.0, 1.0), detrand.uniform(8401, (3, 200, 100), 0.0, 1.0)]
    det = m([T(i) for i in imgs])
    batch, sizes, _ = tv.rcnn_transform(imgs, 160, 256)
    p = m.engine._last_plan
    assert (p.n, p.H, p.W) == (2, batch.shape[2], batch.shape[3]) and not m.engine.normalize
    with torch.no_grad():
        ref = ro.forward(sd, torch.from_numpy(batch), do_normalize=False)
    got = p.logits.cpu()
    assert float((got - ref["cls_logits"]).abs().max()) / float(ref["cls_logits"].abs().max()) < 4e-2
    for d, (h, w) in zip(det, [(120, 150), (200, 100)]):
        b = d["boxes"]
        assert b.shape[0] == d["scores"].shape[0] <= 300
        if b.numel():          # clipped to the RESIZED image, then scaled back: inside the original frame
            assert float(b[:, 0::2].max()) <= w + 1e-3 and float(b[:, 1::2].max()) <= h + 1e-3 and float(b.min()) >= 0.0
    # a ready batch keeps the fused normalisation
    m(torch.rand(2, 3, 128, 128, device="cuda:0"))
    assert m.engine.normalize
