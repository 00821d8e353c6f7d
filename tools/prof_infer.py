"""Eval-mode YOLOv3 forward for a rocprofv3 kernel trace: 5 timed forwards of the frozen-inference plan (tools/bench_yolo_infer.py setup)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from object_detectors_amd.yolo.nets.engine import YoloV3Engine
dev = torch.device("cuda:0")
eng = YoloV3Engine("darknet_53", 3, 80, device=dev, seed=0)
imgs, _ = bench.synth_batch(32, 640, 0, dev)
eng.training = False
eng.freeze_inference(True)
for _ in range(3):
    eng.forward(imgs, training=False)
torch.cuda.synchronize()
marker = torch.zeros(1, device=dev)
for _ in range(5):
    marker.add_(1)            # step marker for tools/infer_table.py (elementwise add kernel on the same stream)
    eng.forward(imgs, training=False)
torch.cuda.synchronize()
