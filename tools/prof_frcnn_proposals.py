"""Host-issue time of the segments of rpn_filter_proposals (no synchronisation inside: what the Python side costs per training step).
    python tools/prof_frcnn_proposals.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detectors_amd import ops
from object_detectors_amd.tvision import postprocess as pp

dev = torch.device("cuda:0")
N, levels = 4, [200 * 200 * 3, 100 * 100 * 3, 50 * 50 * 3, 25 * 25 * 3, 13 * 13 * 3]
A = sum(levels)
g = torch.Generator(device=dev).manual_seed(0)
obj = torch.randn((N, A), device=dev, generator=g)
ctr = torch.rand((N, A, 2), device=dev, generator=g) * 800
wh = torch.rand((N, A, 2), device=dev, generator=g) * 200 + 4
props = torch.cat([ctr - wh / 2, ctr + wh / 2], -1)
shapes = [(800, 800)] * N
for _ in range(5):
    pp.rpn_filter_proposals(props, obj, shapes, levels, 2000, 2000)
torch.cuda.synchronize()
T = {}
def seg(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
reps = 20
for _ in range(reps):
    t0 = time.perf_counter()
    idx_parts, off = [], 0
    for li, n in enumerate(levels):
        k = min(2000, n)
        _v, idx, _c = ops.topk_rows(obj[:, off:off + n], k)
        idx_parts.append(idx + off)
        off += n
    seg("per-level top-k (5 x topk_rows + offset)", t0); t0 = time.perf_counter()
    top_idx = torch.cat(idx_parts, dim=1)
    lv = torch.cat([torch.full((min(2000, n),), li, dtype=torch.int64, device=dev) for li, n in enumerate(levels)]).unsqueeze(0).expand(N, -1)
    batch = torch.arange(N, device=dev)[:, None]
    seg("cat + level / batch index tensors", t0); t0 = time.perf_counter()
    o = torch.sigmoid(obj[batch, top_idx])
    p = props[batch, top_idx]
    seg("gathers + sigmoid", t0); t0 = time.perf_counter()
    hw = torch.tensor([[float(s[1]), float(s[0])] for s in shapes], device=dev, dtype=p.dtype)
    lim = hw.repeat(1, 2)[:, None, :]
    boxes = torch.minimum(p.clamp(min=0), lim)
    ws, hs = boxes[..., 2] - boxes[..., 0], boxes[..., 3] - boxes[..., 1]
    valid = (ws >= 1e-3) & (hs >= 1e-3) & (o >= 0.0)
    masked = torch.where(valid, o, torch.full_like(o, float("-inf")))
    seg("clip + validity mask", t0); t0 = time.perf_counter()
    keeps, counts = [], []
    ar = torch.arange(boxes.shape[1], device=dev)
    for i in range(N):
        keep, cnt = ops.nms_raw(boxes[i], masked[i], 0.7, idxs=lv[i])
        good = (ar < cnt) & valid[i][keep.clamp(max=boxes.shape[1] - 1)]
        keeps.append(keep)
        counts.append(good.sum())
    seg("per-image NMS + kept counts", t0); t0 = time.perf_counter()
    counts = torch.stack(counts).clamp(max=2000).tolist()
    seg("the one read-back (waits for the device)", t0); t0 = time.perf_counter()
    fb = [boxes[i][keeps[i][:counts[i]]] for i in range(N)]
    fs = [o[i][keeps[i][:counts[i]]] for i in range(N)]
    seg("final gathers", t0)
torch.cuda.synchronize()
print("| segment | host ms per call |\n|---|---|")
for k, v in T.items():
    print(f"| {k} | {v / reps * 1e3:.3f} |")
print(f"| total | {sum(T.values()) / reps * 1e3:.3f} |")
