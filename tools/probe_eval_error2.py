"""max-relative head errors (eval / train) of engine and of the bf16-storage oracle vs the fp32 oracle, damped or undamped weights,
running statistics = batch statistics.   python tools/probe_eval_error2.py backbone px bs damp"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import detrand, net_oracle
from object_detectors_amd.yolo.nets.engine import YoloV3Engine, bn_name
bname, px, bs, damp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
dev = torch.device('cuda:0')
sd = net_oracle.det_state(bname, 5000)
for k in sd:
    if k.endswith('.bn2.weight'): sd[k] = sd[k] * damp
x = detrand.uniform(4242, (bs, 3, px, px), -2.0, 2.0)
rec = {}
net_oracle.forward(sd, torch.from_numpy(x), bname, training=True, record=rec)
for name, (z, y) in rec.items():
    b = bn_name(name)
    if b + '.running_mean' in sd:
        sd[b + '.running_mean'] = z.mean((0, 2, 3)); sd[b + '.running_var'] = z.var((0, 2, 3), unbiased=True)
eng = YoloV3Engine(bname, 3, 80, device=dev); eng.load_reference_state_dict(sd)
q = lambda t: t.bfloat16().float()
relmax = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
for training in (False, True):
    outs = eng.forward(torch.from_numpy(x).to(dev), training=training)
    o32 = net_oracle.forward(sd, torch.from_numpy(x), bname, training=training)
    o16 = net_oracle.forward(sd, torch.from_numpy(x), bname, training=training, quant=q)
    print(bname, px, bs, damp, 'train' if training else 'eval', ' '.join(f"head{k}: eng {relmax(o.cpu(), o32[k]):.4f} q {relmax(o16[k], o32[k]):.4f}" for k, o in enumerate(outs)), flush=True)
