"""Every BASELINE.json configuration exercised end to end on the GPU:
  config 1  8 images, darknet_21 eval -> YOLOForw decode -> score filter -> nms_majority (yolo/test.py plumbing);
  config 4  Faster R-CNN data parallel: 2 ranks, engine buckets (GradSync) + box-head parameters (ParamGradSync);
  config 5  RetinaNet ResNet-101-FPN with the LVIS head (1204 classes) against the oracle: forward, tf-idf focal loss, gradients.
(configs 2 and 3 are the subjects of test_gpu_engine.py / test_gpu_retina_engine.py.)"""
import os
import sys

import numpy as np
import pytest

from oracle import detrand, net_oracle
from oracle import retina_oracle as ro
from oracle import tv_oracle as tv
from oracle import yolo_oracle as yo

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ANCHORS = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)


# ------------------------------------------------------------------------------------------------ config 1
def test_config1_inference_chain_8_images():
    """model(images) -> yolo_loss(out) -> get_abs_coord, conf * max class, score filter -> nms_majority, exactly the sequence of
    yolo/procedures/test_one_epoch.py:16-37, on 8 images with darknet_21 in eval mode; checked against the oracle chain (decode ->
    postprocess incl. majority NMS) fed with the ENGINE's head outputs, and the heads themselves against the fp32 network oracle."""
    from object_detectors_amd.yolo.nets.engine import YoloV3Engine, bn_name
    from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
    from object_detectors_amd.yolo.procedures.test_one_epoch import postprocess
    bs, px = 8, 128
    sd = net_oracle.det_state("darknet_21", 5000)
    for k_ in sd:                                # damped residual branches (trained-network-like): the comparison is not lost in chaos
        if k_.endswith(".bn2.weight"):
            sd[k_] = sd[k_] * 0.2
    x = detrand.uniform(4343, (bs, 3, px, px), -2.0, 2.0)
    rec = {}
    net_oracle.forward(sd, torch.from_numpy(x), "darknet_21", training=True, record=rec)
    for name, (z, _y) in rec.items():        # running statistics := batch statistics, so eval mode sees normalised activations
        b = bn_name(name)
        if b + ".running_mean" in sd:
            sd[b + ".running_mean"] = z.mean((0, 2, 3))
            sd[b + ".running_var"] = z.var((0, 2, 3), unbiased=True)
    eng = YoloV3Engine("darknet_21", 3, 80, device=dev())
    eng.load_reference_state_dict(sd)
    crit = YOLOForw(anchors=ANCHORS, num_classes=80, img_size=px).to(dev())
    with torch.no_grad():
        heads = eng.forward(torch.from_numpy(x).to(dev()), training=False)
        ref_heads = net_oracle.forward(sd, torch.from_numpy(x), "darknet_21", training=False)
        q_heads = net_oracle.forward(sd, torch.from_numpy(x), "darknet_21", training=False, quant=lambda t_: t_.bfloat16().float())
        for k in range(3):      # within the bf16-STORAGE noise of the format (fp32 oracle with bf16-rounded storage: 3-8 % of max after 40 layers)
            assert rel(heads[k].cpu(), ref_heads[k]) < 1.3 * rel(q_heads[k], ref_heads[k]) + 0.01, k
        pred = crit(heads)
        assert tuple(pred.shape) == (bs, 3 * (4 * 4 + 8 * 8 + 16 * 16), 85)
        conf = 0.02
        res = postprocess(pred, conf, 0.6, criterion=crit)
    spec = yo.YoloSpec(ANCHORS, 80, px)
    heads_np = [h.float().cpu().numpy() for h in heads]
    want = yo.postprocess(yo.decode(spec, heads_np), conf, 0.6)
    assert len(res) == len(want) and len(want) >= 1
    total = 0
    for got, (cand, fin) in zip(res, want):
        got = got.cpu().numpy()
        assert got.shape == fin.shape, (got.shape, fin.shape, cand.shape)
        np.testing.assert_allclose(got[:, :5], fin[:, :5], rtol=1e-4, atol=1e-3)
        assert np.array_equal(got[:, 5], fin[:, 5])
        total += got.shape[0]
    assert total >= 8


# ------------------------------------------------------------------------------------------------ config 5
PX5, BS5, K5 = 128, 2, 1204


@pytest.fixture(scope="module")
def r101():
    from object_detectors_amd.tvision.engine import RetinaNetEngine
    sd = ro.det_state(7300, K5, body="resnet101")
    eng = RetinaNetEngine(K5, 9, 3, device=dev(), seed=0, body="resnet101")
    eng.load_reference_state_dict(sd)
    x = torch.from_numpy(detrand.uniform(4242, (BS5, 3, PX5, PX5), 0.0, 1.0))
    return eng, sd, x


def test_config5_r101_lvis_forward(r101):
    eng, sd, x = r101
    assert list(eng.reference_state_dict().keys()) == [k for k, _ in ro.state_keys(K5, body="resnet101")]
    out = eng.forward(x.to(dev()), training=False)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = ro.forward(sd, x, num_classes=K5)
    p = eng._last_plan
    for li, (a, r) in enumerate(zip(p.features, ref["features"]), 3):
        assert rel(a.buf.float().permute(0, 3, 1, 2).cpu(), r) < 5e-2, ("P", li)
    assert tuple(out["cls_logits"].shape) == (BS5, 3069, K5)
    assert rel(out["cls_logits"].cpu(), ref["cls_logits"]) < 5e-2 and rel(out["bbox_regression"].cpu(), ref["bbox_regression"]) < 5e-2


def test_config5_r101_lvis_train_step(r101):
    """Whole step at K = 1204 with a tf-idf row (the LVIS recipe, retinanet.py:137): matcher bit-exact, both losses, the logit gradients
    of the fused focal kernel and the parameter gradients against the oracle."""
    eng, sd, x = r101
    rng = np.random.default_rng(5)
    targets, gts = [], []
    for i in range(BS5):
        m = 4 + i
        tl = rng.uniform(0, PX5 * 0.5, (m, 2)).astype(np.float32)
        wh = rng.uniform(PX5 * 0.1, PX5 * 0.45, (m, 2)).astype(np.float32)
        boxes, labels = np.concatenate([tl, tl + wh], 1), rng.integers(1, K5, (m,)).astype(np.int64)
        gts.append((boxes, labels))
        targets.append({"boxes": torch.from_numpy(boxes).to(dev()), "labels": torch.from_numpy(labels).to(dev())})
    tfidf = detrand.uniform(77, (K5,), 0.6, 1.6)
    losses = eng.train_step(x.to(dev()), targets, class_scale=torch.from_numpy(tfidf).to(dev()))
    torch.cuda.synchronize()
    p = eng._last_plan
    cl, rl, mis, (gc, gr) = tv.retinanet_loss(p.logits.cpu().numpy(), p.bbox_reg.cpu().numpy(), p.anchors.cpu().numpy(), gts, tfidf=tfidf)
    assert np.array_equal(p.matched.cpu().numpy(), np.stack(mis))
    np.testing.assert_allclose(losses.cpu().numpy(), [cl, rl], rtol=5e-4)
    # the fused focal kernel writes the class gradient as bf16 into the level buffers of the cls_logits backward (no fp32 tensor)
    np.testing.assert_allclose(p.head_gradient("cls_logits").cpu().numpy(), gc, rtol=6e-3, atol=1e-7)
    np.testing.assert_allclose(p.gbbox.cpu().numpy(), gr, rtol=1e-5, atol=1e-9)
    # parameter gradients of the R101 body / FPN / heads vs fp32 autograd of the oracle driven by the oracle's loss gradient
    sdg = {k: v.clone() for k, v in sd.items()}
    for s in eng.specs:
        if s.trainable:
            sdg[s.name + ".weight"].requires_grad_(True)
            if s.bias:
                sdg[s.name + ".bias"].requires_grad_(True)
    ref = ro.forward(sdg, x, num_classes=K5)
    _cl, _rl, _m, (gc2, gr2) = tv.retinanet_loss(ref["cls_logits"].detach().numpy(), ref["bbox_regression"].detach().numpy(), p.anchors.cpu().numpy(), gts,
                                                  tfidf=tfidf)
    ((ref["cls_logits"] * torch.from_numpy(gc2)).sum() + (ref["bbox_regression"] * torch.from_numpy(gr2)).sum()).backward()
    got = eng.reference_state_dict(grads=True)
    worst = 1.0
    for s in eng.specs:
        if not s.trainable or ".extra_blocks." in s.name:
            continue
        g, r = got[s.name + ".weight"].cpu().double().reshape(-1), sdg[s.name + ".weight"].grad.double().reshape(-1)
        c = float((g @ r) / (g.norm() * r.norm() + 1e-30))
        worst = min(worst, c)
        assert c > 0.98, (s.name, c)
    assert got["backbone.body.layer3.22.conv3.weight"].abs().sum() > 0          # the 23rd block of layer3 exists and trains


# ------------------------------------------------------------------------------------------------ config 4 (2 ranks)
def _worker_frcnn(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = torch.device("cuda:0")
        torch.cuda.set_device(d)
        from object_detectors_amd.parallel import GradSync, ParamGradSync
        from object_detectors_amd.tvision.frcnn import fasterrcnn_resnet50_fpn
        torch.manual_seed(0)                                                      # same box-head initialisation on every rank
        m = fasterrcnn_resnet50_fpn(num_classes=21, device=d, rpn_pre_nms_top_n_train=200, rpn_post_nms_top_n_train=100,
                                    box_batch_size_per_image=32, seed=0)
        for sp in m.engine.specs:
            if sp.bn and sp.bn.endswith(".bn3"):
                m.engine.buffers[sp.bn + ".weight"].fill_(0.2)
        m.engine.refresh_frozen()
        g = torch.Generator().manual_seed(300 + rank)                             # different data per rank
        x = torch.rand((2, 3, 128, 128), generator=g).to(d)
        t = [{"boxes": torch.tensor([[8.0 + 10 * rank, 12.0, 70.0, 90.0], [40.0, 30.0, 120.0, 100.0 + rank]], device=d),
              "labels": torch.tensor([3, 7 + rank], device=d)} for _ in range(2)]
        m.train()

        def step():
            for p in m.head_parameters():
                p.grad = None
            torch.manual_seed(1234 + rank)                                        # the samplers draw from the global RNG: same draw in both passes
            return m(x, t)
        # 1. local gradients, then the plain average as the expectation.  (The first call builds and autotunes the plan; a tile
        #    configuration changes the summation order, and one flipped bf16 rounding can change a proposal and with it the sampled RoIs,
        #    so both measured passes run on the tuned plan.)
        step()
        step()
        torch.cuda.synchronize()
        want_flat = m.engine.flat_g.clone()
        dist.all_reduce(want_flat)
        want_flat /= world
        want_head = [p.grad.clone() for p in m.head_parameters()]
        for w in want_head:
            dist.all_reduce(w)
            w /= world
        # 2. the overlapped path: engine buckets inside backward + one flattened all-reduce of the box head
        sync = GradSync(m.engine.flat_g, bucket_mb=16)
        sync.install(m.engine._last_plan)
        m.head_grad_sync = ParamGradSync(m.head_parameters())
        step()
        sync.wait()
        m.head_grad_sync.wait()
        torch.cuda.synchronize()
        e1 = float((m.engine.flat_g - want_flat).abs().max()) / (float(want_flat.abs().max()) + 1e-30)
        e2 = max(float((p.grad - w).abs().max()) / (float(w.abs().max()) + 1e-30) for p, w in zip(m.head_parameters(), want_head))
        # 3. identical gradients on both ranks afterwards
        chk = torch.stack([m.engine.flat_g.double().abs().sum(), sum(p.grad.double().abs().sum() for p in m.head_parameters())]).cpu()
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        q.put((rank, len(sync.buckets), e1, e2, float(((hi - lo) / hi).max())))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, -1, repr(e) + traceback.format_exc(), 0.0, 0.0))
    finally:
        dist.destroy_process_group()


def test_config4_two_rank_fasterrcnn_gradient_sync():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + os.getpid() % 200
    procs = [ctx.Process(target=_worker_frcnn, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, nb, e1, e2, spread in res:
        assert nb >= 1, (rank, nb, e1)
        assert e1 < 5e-3 and e2 < 5e-3, (rank, e1, e2)          # BN-free network, but RoIAlign backward and weight gradients use fp32 atomics
        assert spread == 0.0, (rank, spread)
