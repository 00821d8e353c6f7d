"""Kernel-level timing source for the detection kernels under rocprofv3 --kernel-trace --stats (decode, focal, top-k, NMS)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd import ops
from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
dev = torch.device('cuda:0')
ANCHORS = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
bs = 32
crit = YOLOForw(anchors=ANCHORS, num_classes=80, img_size=640).to(dev)
heads = [torch.randn(bs, g, g, 256, device=dev)[..., :255].permute(0, 3, 1, 2) for g in (20, 40, 80)]
for _ in range(10): crit(heads)
Nr = 120087
for K in (91, 1204):
    lg = torch.randn(Nr, K, device=dev); m2 = torch.randint(-2, 7, (Nr,), device=dev); lab = torch.randint(1, K, (7,), device=dev)
    for _ in range(10): ops.retina_cls_loss_sum(lg, m2, lab, 0.25, 2.0)
x = torch.randn(1, 360000, device=dev)
x8 = torch.randn(1, 90000 * 91, device=dev)
for _ in range(10): ops.topk_rows(x, 2000); ops.topk_rows(x8, 1000, min_value=-2.944)
torch.cuda.synchronize()
