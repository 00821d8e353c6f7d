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
q = lambda t: t.bfloat16().float()
for mode, quant in (('fp32', None), ('bf16q', q)):
    s2 = {k: v.clone() for k, v in sd.items()}
    for k, v in s2.items():
        if v.dtype == torch.float32: v.requires_grad_(not k.endswith(('running_mean', 'running_var')))
    ro = net_oracle.forward(s2, torch.from_numpy(x), bname, True, quant)
    for k, o in enumerate(outs):
        r = ro[k].detach().numpy(); e = np.abs(o.cpu().numpy() - r)
        print(mode, 'out', k, 'max_err/max', e.max() / np.abs(r).max(), 'rms_err/rms', np.sqrt((e**2).mean()) / np.sqrt((r**2).mean()))
    cots = [detrand.uniform(4300 + k, tuple(o.shape), -1.0, 1.0) for k, o in enumerate(outs)]
    sum((o * torch.from_numpy(c)).sum() for o, c in zip(ro, cots)).backward()
    if mode == 'fp32':
        eng.backward([torch.from_numpy(c).to(dev) for c in cots])
        got = eng.reference_state_dict(grads=True)
    worst = []
    for n, v in s2.items():
        if v.grad is None: continue
        gg = got[n].cpu().double(); og = v.grad.double()
        cos = float((gg * og).sum() / (gg.norm() * og.norm() + 1e-30)); rel = float((gg - og).norm() / (og.norm() + 1e-30))
        worst.append((rel, cos, n))
    worst.sort(reverse=True)
    print(mode, 'grad worst rel:', [(round(a, 3), round(b, 4), n) for a, b, n in worst[:6]])
    print(mode, 'grad median rel:', sorted(w[0] for w in worst)[len(worst) // 2])
