import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import detrand, net_oracle
from object_detectors_amd.yolo.nets.engine import YoloV3Engine
bname = sys.argv[1]; px = int(sys.argv[2]); bs = int(sys.argv[3])
dev = torch.device('cuda:0')
eng = YoloV3Engine(bname, 3, 80, device=dev)
sd = net_oracle.det_state(bname, 5000)
eng.load_reference_state_dict(sd)
x = detrand.uniform(4242, (bs, 3, px, px), -2.0, 2.0)
outs = eng.forward(torch.from_numpy(x).to(dev), training=True)
plan = eng._last_plan
rec = {}
with torch.no_grad():
    net_oracle.forward(sd, torch.from_numpy(x), bname, True, lambda t: t.bfloat16().float(), record=rec)
for name, r in plan.layers.items():
    z = r['z'].float().permute(0, 3, 1, 2).cpu()
    zo, yo = rec[name]
    e = (z - zo)
    print(f"{name:40s} z rms_rel {float(e.pow(2).mean().sqrt() / zo.pow(2).mean().sqrt()):.4f}  max_rel {float(e.abs().max() / zo.abs().max()):.4f}")
