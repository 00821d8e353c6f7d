"""Micro-benchmark of the HBM-bound detection kernels at the BASELINE.json sizes (SURVEY §8d algorithmic bytes).
Prints one JSON line per kernel: time, algorithmic bytes, achieved GB/s, fraction of the 8 TB/s HBM peak."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import ops
from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
from object_detectors_amd.tvision._utils import Matcher
from object_detectors_amd.tvision.anchor_utils import AnchorGenerator
dev = torch.device('cuda:0')
ANCHORS = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def report(name, us, nbytes, note=""):
    gbs = nbytes / us / 1e3
    print(json.dumps({"kernel": name, "us": round(us, 2), "algorithmic_MB": round(nbytes / 1e6, 3), "GB/s": round(gbs, 1),
                      "frac_of_8TBs": round(gbs / 8000, 4), "note": note}), flush=True)

torch.manual_seed(0)
bs, N, C, M = 32, 25200, 80, 7
crit = YOLOForw(anchors=ANCHORS, num_classes=C, img_size=640).to(dev)
heads = [torch.randn(bs, g, g, 256, device=dev)[..., :255].permute(0, 3, 1, 2) for g in (20, 40, 80)]   # engine-native NHWC views
targets = [{"bbox": torch.cat([torch.rand(M, 2) * .6 + .2, torch.rand(M, 2) * .3 + .02], 1).to(dev),
            "category_id": torch.randint(0, C, (M,)).to(dev)} for _ in range(bs)]
geom = crit._geom([20, 40, 80])
boxes, labels, off, counts = ops.flatten_targets(targets, dev)
report("yolo_assign (get_target, bs=32, N=25200, M=7)", timeit(lambda: ops.yolo_assign(geom, boxes, off, bs, counts)), bs * (M * 16 + N), "latency-bound: 0.8 MB")
grads = [torch.zeros(bs, g, g, 256, device=dev, dtype=torch.bfloat16)[..., :255].permute(0, 3, 1, 2) for g in (20, 40, 80)]
gv, _k = ops.head_views(grads, 255, dtype=torch.bfloat16)
report("yolo_loss fwd+bwd (assign + noobj + positives + reduce)", timeit(lambda: crit._loss_impl(heads, targets, True, grad_views=gv, grad_is_bf16=True)),
       bs * N * 4 + bs * N * 2 + bs * N + bs * M * 85 * 4 * 2, "conf planes + bf16 conf grads + noobj bytes + positive rows")
report("yolo_decode (bs=32)", timeit(lambda: crit(heads)), 2 * bs * N * 85 * 4)
pred = crit(heads)
report("yolo_candidates (score+compaction)", timeit(lambda: ops.yolo_candidates(pred, 0.1)), bs * N * 85 * 4, "reads decoded predictions once")
_r, _v, sc_, lb_ = crit.last_decode_scores
report("yolo_candidates with fused decode scores", timeit(lambda: ops.yolo_candidates(pred, 0.1, score=sc_, label=lb_)), bs * N * 8, "score+label arrays only")
for n in (1000, 5000, 10000):
    c = torch.rand(bs, n, 2, device=dev) * 600 + 20
    s = torch.exp(torch.rand(bs, n, 2, device=dev) * 3.2 + 2.0)
    P = torch.cat([c - s / 2, c + s / 2, torch.rand(bs, n, 1, device=dev), torch.randint(0, 80, (bs, n, 1), device=dev).float()], 2).contiguous()
    cnt = torch.full((bs,), n, device=dev, dtype=torch.int32)
    report(f"nms_majority bs=32 n={n}", timeit(lambda: ops.nms_majority_batched(P, cnt, 0.6, 80), 5), bs * (n * 24 + n * 8), f"+ bit mask {bs * n * n // 8 / 1e6:.1f} MB implementation traffic")
    b1, s1 = P[0, :, :4].contiguous(), P[0, :, 4].contiguous()
    for K in (1, 90, 1203):
        idxs = torch.randint(0, K, (n,), device=dev)
        report(f"batched_nms n={n} K={K}", timeit(lambda: ops.nms(b1, s1, 0.5, idxs=idxs), 5), n * 20 + n * 8)
class IL: pass
for tag, sizes, grids in (("retina N=120087", tuple((x, int(x * 2 ** (1 / 3)), int(x * 2 ** (2 / 3))) for x in [32, 64, 128, 256, 512]), [100, 50, 25, 13, 7]),
                          ("rpn N=159882", ((32,), (64,), (128,), (256,), (512,)), [200, 100, 50, 25, 13])):
    il = IL(); il.tensors = torch.zeros(1, 3, 800, 800); il.image_sizes = [(800, 800)]
    ag = AnchorGenerator(sizes, ((0.5, 1.0, 2.0),) * 5)
    fm = [torch.zeros(1, 1, g, g, device=dev) for g in grids]
    anchors = ag(il, fm)[0]
    Nn = anchors.shape[0]
    report(f"anchor_grid {tag}", timeit(lambda: ag(il, fm)), Nn * 16)
    gt = torch.rand(M, 2, device=dev) * 400
    gt = torch.cat([gt, gt + torch.rand(M, 2, device=dev) * 380 + 16], 1)
    mt = Matcher(0.5, 0.4, True)
    report(f"match_anchors (box_iou+Matcher fused) {tag} M=7", timeit(lambda: mt.match_boxes(gt, anchors)), 2 * (Nn * 16 + M * 16) + Nn * 8, "anchors read twice (low-quality rescue)")
    report(f"box_iou [7,{Nn}] materialised", timeit(lambda: ops.box_iou(gt, anchors)), Nn * 16 + M * Nn * 4)
matched = mt.match_boxes(gt, anchors)
for K in (91, 1204):
    Nr = 120087
    lg = torch.randn(Nr, K, device=dev)
    lab = torch.randint(1, K, (M,), device=dev)
    m2 = torch.where(torch.rand(Nr, device=dev) < 0.01, torch.randint(0, M, (Nr,), device=dev), torch.full((Nr,), -1, device=dev))
    report(f"retina_cls_loss fused fwd+bwd N=120087 K={K}", timeit(lambda: ops.retina_cls_loss_sum(lg, m2, lab, 0.25, 2.0), 10), 2 * Nr * K * 4)
x = torch.randn(1, 120000 * 3, device=dev)
report("topk rows=1 n=360000 k=2000", timeit(lambda: ops.topk_rows(x, 2000), 10), 360000 * 4 * 4, "4 passes over the row")
x8 = torch.randn(1, 90000 * 91, device=dev)
report("topk rows=1 n=8.19M (RetinaNet level 0: 90000 anchors x 91) k=1000", timeit(lambda: ops.topk_rows(x8, 1000, min_value=-2.944), 10), x8.numel() * 4 * 4, "4 passes over the row")
x16 = torch.randn(16, 90000 * 91, device=dev)
report("topk rows=16 n=8.19M k=1000", timeit(lambda: ops.topk_rows(x16, 1000, min_value=-2.944), 5), x16.numel() * 4 * 4, "4 passes over the rows")
from object_detectors_amd.tvision.roi_align import MultiScaleRoIAlign
feats = {str(i): torch.randn(2, 256, 800 // s, 800 // s, device=dev) for i, s in enumerate((4, 8, 16, 32))}
props = [torch.cat([torch.rand(512, 2, device=dev) * 500, torch.rand(512, 2, device=dev) * 280 + 520], 1) for _ in range(2)]
msra = MultiScaleRoIAlign(["0", "1", "2", "3"], 7, 2)
report("MultiScaleRoIAlign 1024 RoIs x 256 ch 7x7", timeit(lambda: msra(feats, props, [(800, 800)] * 2), 10), 1024 * 256 * 49 * 4 * 5, "output + 4 gathered samples per bin-sample (upper bound)")
