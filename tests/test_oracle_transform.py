"""oracle/tv_oracle.py input-transform restatement against the reference's GeneralizedRCNNTransform / F.interpolate outputs (fixture g14)."""
import numpy as np

from oracle import detrand
from oracle import tv_oracle as tv


def fixture_images(g):
    return [detrand.uniform(8000 + i, (3, int(h), int(w)), 0.0, 1.0) for i, (h, w) in enumerate(g["shapes"])]


def test_rcnn_transform_eval(golden):
    g = golden("g14_transform")
    imgs = fixture_images(g)
    boxes = [g[f"boxes{i}"] for i in range(len(imgs))]
    batch, sizes, nb = tv.rcnn_transform(imgs, boxes=boxes)
    assert list(batch.shape) == g["eval_batch_shape"].tolist()
    assert [list(s) for s in sizes] == g["eval_image_sizes"].tolist()          # incl. 375x500 -> 799x1066 (float32 scale) and the max_size case
    # a source coordinate near 500 carries float32 rounding of 3e-5, and ATen's vectorised CPU kernel contracts scale*(i+0.5)-0.5 into an FMA: 1e-4 absolute
    np.testing.assert_allclose(batch[:, :, ::13, ::17], g["eval_batch_sample"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(batch.astype(np.float64).sum((1, 2, 3)), g["eval_batch_sum"], rtol=1e-6)
    for i in range(len(imgs)):
        np.testing.assert_allclose(nb[i], g[f"eval_boxes{i}"], rtol=1e-6)
        h, w = g["shapes"][i]
        np.testing.assert_allclose(tv.resize_boxes(nb[i], sizes[i], (h, w)), g[f"post_boxes{i}"], rtol=1e-6)


def test_yolo_multiscale_interpolate(golden):
    g = golden("g14_transform")
    x = detrand.uniform(8300, (2, 3, 416, 416), -2.0, 2.0)
    for size in (320, 608):
        y = tv.bilinear_resize(x, size, size)
        np.testing.assert_allclose(y[:, :, ::7, ::11], g[f"yolo_ms_{size}_sample"], rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(y.astype(np.float64).sum((1, 2, 3)), g[f"yolo_ms_{size}_sum"], rtol=1e-6, atol=1e-2)
