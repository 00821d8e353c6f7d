"""NMS kernels under rocprofv3: rocprofv3 --kernel-trace --stats -- python3 tools/prof_nms.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detectors_amd.tvision import boxes as box_ops
from object_detectors_amd.yolo.utilities import helper
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for n in (1000, 10000):
    c = torch.rand((n, 2), generator=g) * 600 + 20
    wh = torch.exp(torch.rand((n, 2), generator=g) * 3.2 + 2.0)
    b = torch.cat([c - wh / 2, c + wh / 2], 1).to(dev)
    s = torch.rand((n,), generator=g).to(dev)
    idx = torch.randint(0, 90, (n,), generator=g).to(dev)
    for _ in range(5):
        keep = box_ops.batched_nms(b, s, idx, 0.5)
    torch.cuda.synchronize()
    print(n, "kept", keep.numel())
    P = torch.cat([b, s[:, None], idx[:, None].float()], 1).unsqueeze(0).repeat(32, 1, 1).contiguous()
    for _ in range(3):
        out = helper.nms_majority(P[0].clone())
    torch.cuda.synchronize()
