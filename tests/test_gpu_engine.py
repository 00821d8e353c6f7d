"""GPU parity of the YOLOv3 engine (Darknet + head, fwd/bwd).

Deep randomly-initialised BN networks amplify rounding differences (~1.15x per layer measured on the
golden network), so an end-to-end bf16-vs-fp32 comparison of the 75-conv graph is only meaningful at a
loose tolerance.  The tight checks are therefore LOCAL and exhaustive: every layer's forward and
backward is compared with a plain PyTorch fp32 evaluation of that layer on the engine's own stored
operands, and every activation gradient must equal the sum of its consumers' contributions (which pins
the graph wiring: residual skips, concat slices, upsample, branch points).  End-to-end runs against
oracle/net_oracle.py (pinned to the reference's golden outputs) use a damped-residual weight set.
"""
import numpy as np
import pytest

from oracle import detrand, net_oracle
from oracle import yolo_oracle as yo

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

ANCHORS = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]


def dev():
    return torch.device("cuda:0")


def bf16q(t):
    return t.bfloat16().float()


def make_engine(bname, wseed, damp=None):
    from object_detectors_amd.yolo.nets.engine import YoloV3Engine
    eng = YoloV3Engine(bname, 3, 80, device=dev())
    sd = net_oracle.det_state(bname, wseed)
    if damp:
        for k in sd:
            if k.endswith(".bn2.weight"):
                sd[k] = sd[k] * damp
    eng.load_reference_state_dict(sd)
    return eng, sd


def view(a):
    return a.buf.view(a.n, a.h, a.w, -1)[..., a.ch_off:a.ch_off + a.c].float().permute(0, 3, 1, 2).cpu()


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("bname,px,bs", [("darknet_21", 96, 2), ("darknet_53", 96, 2)])
def test_engine_local_parity_and_wiring(bname, px, bs):
    from object_detectors_amd.yolo.nets.engine import bn_name
    eng, sd = make_engine(bname, 5000)
    x = detrand.uniform(4242, (bs, 3, px, px), -2.0, 2.0)
    outs = eng.forward(torch.from_numpy(x).to(dev()), training=True)
    cots = [detrand.uniform(4300 + k, tuple(o.shape), -1.0, 1.0) for k, o in enumerate(outs)]
    eng.backward([torch.from_numpy(c).to(dev()) for c in cots])
    plan = eng._last_plan
    got = eng.reference_state_dict(grads=True)
    contrib = {}      # id(Act) -> expected gradient (sum of consumer contributions), fp32 torch

    def add(act, t):
        contrib[id(act)] = contrib.get(id(act), 0) + t
    acts = {}
    bad = []
    for rec in plan.ops:
        if rec["kind"] == "up":
            xa, cat, c_up = rec["x"], rec["cat"], rec["c_up"]
            gcat = view(cat.grad)
            g = gcat[:, :c_up]
            add(xa, g.reshape(g.shape[0], g.shape[1], xa.h, 2, xa.w, 2).sum((3, 5)))
            add(rec["skip_to"], gcat[:, c_up:])
            acts[id(xa)], acts[id(rec["skip_to"])] = xa, rec["skip_to"]
            # forward of the upsample + concat
            up = F.interpolate(view(xa), scale_factor=2, mode="nearest")
            assert torch.equal(view(cat)[:, :c_up], up)
            continue
        s, xa, name = rec["spec"], rec["x"], rec["name"]
        if xa is not None:              # the stem reads the image, not an activation
            acts[id(xa)] = xa
            xin = view(xa).requires_grad_(True)
        if rec["kind"] == "out":
            k = rec["k"]
            w = bf16q(sd[name + ".weight"]).requires_grad_(True)
            bb = sd[name + ".bias"].clone().requires_grad_(True)
            y = F.conv2d(xin, w, bb)
            o = outs[k].cpu()
            if (o - y.detach()).abs().max() > 1e-2 * y.abs().max():
                bad.append((name, "fwd"))
            y.backward(bf16q(torch.from_numpy(cots[k])))
            if rel(got[name + ".weight"].cpu(), w.grad) > 2e-2 or rel(got[name + ".bias"].cpu(), bb.grad) > 2e-2:
                bad.append((name, "dW/db", rel(got[name + ".weight"].cpu(), w.grad)))
            add(xa, xin.grad)
            continue
        a, b = rec["a"], bn_name(name)
        # ---- forward: z = conv(x), a = lrelu(bn(z)) (+res) on the engine's own operands
        if name == "backbone.conv1":
            # the stem never stores z (csrc/stem_kernels.hip recomputes it from the image in every pass): the reference z stands in
            zz = F.conv2d(bf16q(torch.from_numpy(x)), bf16q(sd[name + ".weight"]), padding=1)
            assert rec["z"] is None
            zg = zz
        else:
            zg = rec["z"].float().permute(0, 3, 1, 2).cpu()
            zz = F.conv2d(xin.detach(), bf16q(sd[name + ".weight"]), stride=s.stride, padding=(s.k - 1) // 2)
        if (zg - zz).abs().max() > 1e-2 * zz.abs().max():
            bad.append((name, "z", float((zg - zz).abs().max() / zz.abs().max())))
        z = zg.clone().requires_grad_(True)
        gam, bet = sd[b + ".weight"].clone().requires_grad_(True), sd[b + ".bias"].clone().requires_grad_(True)
        y = F.leaky_relu(F.batch_norm(z, None, None, gam, bet, True, 0.1, 1e-5), 0.1)
        yy = y.detach() + (view(rec["res"]) if rec["res"] is not None else 0)
        if (view(a) - yy).abs().max() > 1.5e-2 * yy.abs().max():
            bad.append((name, "a", float((view(a) - yy).abs().max() / yy.abs().max())))
        # ---- backward given the engine's gradient of a
        G = view(a.grad)
        y.backward(G)
        if rel(got[b + ".weight"].cpu(), gam.grad) > 2e-2 or rel(got[b + ".bias"].cpu(), bet.grad) > 8e-2:
            bad.append((name, "dgamma/dbeta", rel(got[b + ".weight"].cpu(), gam.grad), rel(got[b + ".bias"].cpu(), bet.grad)))
        dz = bf16q(z.grad)
        if name == "backbone.conv1":
            xi = bf16q(torch.from_numpy(x)).requires_grad_(True)
            w = bf16q(sd[name + ".weight"]).requires_grad_(True)
            F.conv2d(xi, w, padding=1).backward(dz)
        else:
            w = bf16q(sd[name + ".weight"]).requires_grad_(True)
            F.conv2d(xin, w, stride=s.stride, padding=(s.k - 1) // 2).backward(dz)
            add(xa, xin.grad)
        if rel(got[name + ".weight"].cpu(), w.grad) > 6e-2:
            bad.append((name, "dW", rel(got[name + ".weight"].cpu(), w.grad)))
        if rec["res"] is not None:
            add(rec["res"], G)
            acts[id(rec["res"])] = rec["res"]
    # ---- wiring: every activation gradient == sum of its consumers' contributions
    for key, exp in contrib.items():
        act = acts[key]
        g = view(act.grad)
        if rel(g, exp) > 3e-2:
            bad.append(("grad-sum", act.c, act.h, rel(g, exp)))
    assert not bad, bad[:12]


def test_engine_end_to_end_vs_oracle(golden):
    """End to end against the pinned torch oracle with bf16-rounded storage, damped residual branches."""
    bname, px, bs = "darknet_21", 128, 4
    eng, sd = make_engine(bname, 5000, damp=0.2)
    x = detrand.uniform(4242, (bs, 3, px, px), -2.0, 2.0)
    outs = eng.forward(torch.from_numpy(x).to(dev()), training=True)
    sdq = {k: v.clone() for k, v in sd.items()}
    for k, v in sdq.items():
        if v.dtype == torch.float32:
            v.requires_grad_(not k.endswith(("running_mean", "running_var")))
    ref_outs = net_oracle.forward(sdq, torch.from_numpy(x), bname, training=True, quant=bf16q)
    errs = []
    for k, o in enumerate(outs):
        r = ref_outs[k].detach()
        errs.append(rel(o.cpu(), r))
    assert max(errs) < 0.08, errs
    cots = [detrand.uniform(4300 + k, tuple(o.shape), -1.0, 1.0) for k, o in enumerate(outs)]
    sum((o * torch.from_numpy(c)).sum() for o, c in zip(ref_outs, cots)).backward()
    eng.backward([torch.from_numpy(c).to(dev()) for c in cots])
    got = eng.reference_state_dict(grads=True)
    coss = []
    for n, v in sdq.items():
        if v.grad is None:
            continue
        gg, og = got[n].cpu().double(), v.grad.double()
        coss.append((float((gg * og).sum() / (gg.norm() * og.norm() + 1e-30)), n))
    coss.sort()
    assert coss[len(coss) // 10][0] > 0.8, coss[:8]          # 90 % of the tensors: cosine > 0.8
    assert coss[0][0] > 0.5, coss[:8]


def test_engine_golden_reference_outputs(golden):
    """Against the REFERENCE modules' own outputs (fixture g8, 64 px): loose, bf16 through 40-75 chaotic layers (2x2 maps: BN over 8
    values).  The eval-mode half needs the same tolerance only because it runs on the ENGINE's running statistics, which that noisy
    train-mode pass produced; test_engine_vs_reference_256px is the meaningful network-level comparison."""
    g = golden("g8_network")
    for bname in ("darknet_21", "darknet_53"):
        wseed, xseed, cseed, px, bs = [int(v) for v in g[bname + "_meta"]]
        eng, sd = make_engine(bname, wseed)
        x = detrand.uniform(xseed, (bs, 3, px, px), -2.0, 2.0)
        outs = eng.forward(torch.from_numpy(x).to(dev()), training=True)
        for k, o in enumerate(outs):
            ref = torch.from_numpy(g[f"{bname}_out{k}"])
            assert rel(o.cpu(), ref) < 0.45, (bname, k, rel(o.cpu(), ref))      # chaotic (2x2 maps, BN over 8 values): measured 0.2-0.36
        # first-layer running statistics are exact to bf16 input rounding
        np.testing.assert_allclose(eng.buffers["backbone.bn1.running_mean"].cpu().numpy(), g[bname + "_rm_stem"], rtol=2e-2, atol=2e-3)
        np.testing.assert_allclose(eng.buffers["backbone.bn1.running_var"].cpu().numpy(), g[bname + "_rv_stem"], rtol=2e-2, atol=2e-3)
        ev = eng.forward(torch.from_numpy(x).to(dev()), training=False)
        for k, o in enumerate(ev):
            ref = torch.from_numpy(g[f"{bname}_evalout{k}"])
            assert rel(o.cpu(), ref) < 0.45, (bname, "eval", k, rel(o.cpu(), ref))


def test_engine_vs_reference_256px(golden):
    """Network-level parity at a size where it means something: the REFERENCE YoloHead (Darknet-53) at 256 px, batch 4 (fixture g8b:
    8x8 / 16x16 / 32x32 maps, BatchNorm over 256-4096 values; residual branches damped x0.2 like trained / zero-init-residual networks),
    eval mode on the reference's own running statistics and train mode on batch statistics.
    Yardstick: bf16 STORAGE alone (fp32 arithmetic, activations and weights rounded to bf16 between layers: the oracle with quant=bf16)
    sits 4-10 % of the largest value away from the fp32 reference after 75 layers (max over ~10^5-10^6 elements; measured eval 0.047 /
    0.084 / 0.103, train 0.051 / 0.080 / 0.126 for the engine against 0.050 / 0.086 / 0.107 and 0.051 / 0.078 / 0.102 for that yardstick),
    so the bound is relative to it: engine error <= 1.3 x storage-noise error + 0.01, and <= 0.16 outright."""
    g = golden("g8b_network256")
    wseed, xseed, px, bs = [int(v) for v in g["meta"]]
    eng, sd = make_engine("darknet_53", wseed, damp=float(g["damp"][0]))
    off = 0
    for name, size in zip(g["bn_names"], g["bn_sizes"]):
        sd[str(name) + ".running_mean"] = torch.from_numpy(g["running_mean"][off:off + int(size)].copy())
        sd[str(name) + ".running_var"] = torch.from_numpy(g["running_var"][off:off + int(size)].copy())
        off += int(size)
    eng.load_reference_state_dict(sd)
    x = detrand.uniform(xseed, (bs, 3, px, px), -2.0, 2.0)
    xd = torch.from_numpy(x).to(dev())
    for mode, training in (("eval", False), ("train", True)):
        outs = eng.forward(xd, training=training)
        o32 = net_oracle.forward(sd, torch.from_numpy(x), "darknet_53", training=training)
        o16 = net_oracle.forward(sd, torch.from_numpy(x), "darknet_53", training=training, quant=bf16q)
        for k, o in enumerate(outs):
            ref = torch.from_numpy(g[f"{mode}_out{k}"])
            step = o.shape[-1] // 8
            amax = float(g[f"{mode}_out{k}_absmax"][0])
            assert float((o32[k][:, :, ::step, ::step] - ref).abs().max()) / amax < 1e-3, (mode, k)      # the oracle IS the reference (fp32 both)
            err = float((o.cpu()[:, :, ::step, ::step] - ref).abs().max()) / amax
            err_q = float((o16[k] - o32[k]).abs().max()) / amax
            assert err < 1.3 * err_q + 0.01 and err < 0.16, (mode, k, err, err_q)


def test_engine_error_envelope_undamped_weights():
    """Undamped random weights make the 75-layer train-mode map chaotic: bf16 STORAGE noise grows ~1.15x per layer whoever does the
    arithmetic - the fp32 oracle evaluated with bf16-rounded storage ends 0.12-0.23 (relative L2) away from the fp32 run at the heads
    (256 px, batch 4).  The engine must stay inside that envelope AT EVERY LAYER (its error <= 1.3 x the storage-noise error + 0.003):
    it adds nothing of its own.  (This is also why the 64-px g8 test above needs 0.45 even in eval mode: its running statistics come
    from such a noisy train-mode pass over 2x2 maps.  Eval mode is exactly as accurate as its bf16 storage allows: test_engine_vs_reference_256px.)"""
    eng, sd = make_engine("darknet_53", 5000)
    x = detrand.uniform(4242, (4, 3, 256, 256), -2.0, 2.0)
    outs = eng.forward(torch.from_numpy(x).to(dev()), training=True)
    torch.cuda.synchronize()
    plan = eng._last_plan
    rec32, rec16 = {}, {}
    o32 = net_oracle.forward(sd, torch.from_numpy(x), "darknet_53", training=True, record=rec32)
    o16 = net_oracle.forward(sd, torch.from_numpy(x), "darknet_53", training=True, quant=bf16q, record=rec16)
    for k, o in enumerate(outs):
        e_eng, e_q = rel(o.cpu(), o32[k]), rel(o16[k], o32[k])
        assert e_eng < 1.3 * e_q + 3e-3 and e_eng < 0.35, ("train head", k, e_eng, e_q)
    for name, r in plan.layers.items():
        if r.get("res") is not None:
            continue
        e_eng, e_q = rel(view(r["a"]), rec32[name][1]), rel(rec16[name][1], rec32[name][1])
        assert e_eng < 1.3 * e_q + 3e-3, (name, e_eng, e_q)


def test_yolohead_module_autograd_and_fused_step():
    """drop-in modules: loss.backward() through YoloHead + YOLOForw == fused engine.train_step."""
    from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
    from object_detectors_amd.yolo.nets.yolohead import YoloHead
    from tests.helpers import synth_targets
    cfg = {"backbone": {"backbone_name": "darknet_21", "backbone_pretrained": ""}, "dataset": {"anchors": ANCHORS},
           "yolo": {"classes": 80}}
    model = YoloHead(cfg).to(dev())
    crit = YOLOForw(anchors=ANCHORS, num_classes=80, img_size=128).to(dev())
    x = torch.from_numpy(detrand.uniform(5, (2, 3, 128, 128), -2, 2)).to(dev())
    tg = [{"bbox": torch.from_numpy(b).to(dev()), "category_id": torch.from_numpy(l).to(dev())}
          for b, l in synth_targets(77, (5, 3), 80)]
    model.train()
    for p in model.parameters():
        p.grad = None
    out = model(x)
    assert [tuple(o.shape) for o in out] == [(2, 255, 4, 4), (2, 255, 8, 8), (2, 255, 16, 16)]
    loss, sub, stats = crit(out, tg)
    loss.backward()
    g_auto = model.engine.flat_g.clone()
    assert all(p.grad is not None for p in model.parameters())
    out12 = model.engine.train_step(x, tg, crit)
    g_fused = model.engine.flat_g
    np.testing.assert_allclose(out12[0].item(), loss.item(), rtol=1e-5)
    # fused path rounds the head gradient to bf16 once more: same direction
    cos = float((g_auto.double() * g_fused.double()).sum() / (g_auto.double().norm() * g_fused.double().norm()))
    assert cos > 0.999
    # the criterion on the engine's NHWC views equals the oracle on the same logits
    heads = [o.detach().cpu().numpy() for o in model.engine.forward(x, training=True)]
    spec = yo.YoloSpec(ANCHORS, 80, 128)
    ref = yo.yolo_loss(spec, heads, synth_targets(77, (5, 3), 80), want_grad=False)
    np.testing.assert_allclose(out12[0].item(), ref["loss"], rtol=2e-3)
    # reference-format checkpoint round trip
    sd = model.state_dict()
    m2 = YoloHead(cfg).to(dev())
    m2.load_state_dict({"module." + k: v for k, v in sd.items()})
    for k, v in m2.state_dict().items():
        assert torch.equal(v.cpu(), sd[k].cpu()), k
    assert list(sd.keys()) == [k for k, _ in net_oracle.state_keys("darknet_21")]


@pytest.mark.gpu
def test_flat_optimizers_match_torch():
    """mi355det_sgd_step / adam_step vs torch.optim.SGD / Adam (initialize.py:38,41) over several steps."""
    from object_detectors_amd.optim import FlatAdam, FlatSGD
    torch.manual_seed(0)
    n = 100003
    w0 = torch.randn(n, device="cuda")
    grads = [torch.randn(n, device="cuda") for _ in range(4)]
    for kind in ("sgd", "sgd_nesterov", "adam"):
        w = torch.zeros(n + 1, device="cuda")[:n]
        w.copy_(w0)
        g = torch.zeros_like(w)
        ref = w0.clone().requires_grad_(True)
        if kind == "adam":
            mine = FlatAdam(w, g, lr=1e-2, weight_decay=5e-4)
            theirs = torch.optim.Adam([ref], lr=1e-2, weight_decay=5e-4)
        else:
            mine = FlatSGD(w, g, lr=1e-2, momentum=0.9, weight_decay=5e-4, nesterov=kind.endswith("nesterov"))
            theirs = torch.optim.SGD([ref], lr=1e-2, momentum=0.9, weight_decay=5e-4, nesterov=kind.endswith("nesterov"))
        for gi in grads:
            g.copy_(gi * 4.0)
            ref.grad = gi.clone()
            mine.step(grad_scale=0.25, zero_grad=True)
            theirs.step()
            assert float(g.abs().max()) == 0.0
        torch.testing.assert_close(w, ref.detach(), rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
def test_dynamic_loss_scaler_skips_overflow_steps():
    """apex amp semantics on the flat buffers (initialize.py:44-45, train_one_epoch.py:88-96): an inf / nan gradient skips the step, leaves
    parameters and optimizer state untouched and halves the scale; clean steps equal torch.optim with the unscaled gradient; the scale
    doubles after `scale_window` clean steps."""
    from object_detectors_amd.optim import DynamicLossScaler, FlatAdam, FlatSGD
    torch.manual_seed(1)
    n = 50001
    w0 = torch.randn(n, device="cuda")
    grads = [torch.randn(n, device="cuda") for _ in range(5)]
    for kind in ("sgd", "adam"):
        w = w0.clone()
        g = torch.zeros_like(w)
        ref = w0.clone().requires_grad_(True)
        if kind == "adam":
            mine, theirs = FlatAdam(w, g, lr=1e-2), torch.optim.Adam([ref], lr=1e-2)
        else:
            mine, theirs = FlatSGD(w, g, lr=1e-2, momentum=0.9), torch.optim.SGD([ref], lr=1e-2, momentum=0.9)
        scaler = DynamicLossScaler(init_scale=1024.0, scale_window=2)
        applied = []
        for it, gi in enumerate(grads):
            g.copy_(gi * scaler.loss_scale)
            if it in (0, 3):                                  # overflow on the very first step (SGD's first-step flag) and later
                g[it * 7 + 5] = float("inf") if it == 0 else float("nan")
            before = w.clone()
            ok = scaler.step(mine, zero_grad=True)
            applied.append(ok)
            assert float(g.abs().max()) == 0.0               # cleared either way
            if ok:
                ref.grad = gi.clone()
                theirs.step()
            else:
                assert torch.equal(w, before)
        assert applied == [False, True, True, False, True]
        torch.testing.assert_close(w, ref.detach(), rtol=2e-5, atol=2e-6)
        # 1024 -> 512 (overflow) -> 2 clean steps -> 1024 -> overflow -> 512 -> one clean step
        assert scaler.loss_scale == 512.0 and scaler.skipped_steps == 2 and mine.steps == 3
    # odd length / unaligned tail of the check kernel
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    from object_detectors_amd._lib import check, lib
    v = torch.zeros(1027, device="cuda")
    v[1026] = float("-inf")
    check(lib().mi355det_grad_nonfinite(v.data_ptr(), v.numel(), flag.data_ptr(), torch.cuda.current_stream().cuda_stream), "grad_nonfinite")
    assert int(flag.item()) == 1
    flag.zero_()
    v[1026] = 3.0e38
    check(lib().mi355det_grad_nonfinite(v.data_ptr(), v.numel(), flag.data_ptr(), torch.cuda.current_stream().cuda_stream), "grad_nonfinite")
    assert int(flag.item()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [dict(), dict(class_loss=0, reduction="mean"), dict(class_loss=2, img_freq="synthetic")])
def test_fused_step_with_loss_scale_and_criterion_options(opts):
    """The amp usage pattern (optim.DynamicLossScaler docstring): engine.train_step(..., grad_scale=S) produces S x the gradients (S a power of
    two: exact up to the rounding order of the atomics), the guarded step with grad_scale 1/S then equals the unscaled step; for every class-loss
    form the fused bf16-gradient path agrees with loss.backward() through the modules."""
    from object_detectors_amd.optim import DynamicLossScaler, FlatSGD
    from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
    from object_detectors_amd.yolo.nets.yolohead import YoloHead
    from tests.helpers import synth_targets
    opts = dict(opts)
    if opts.get("img_freq") == "synthetic":
        opts["img_freq"] = np.exp(detrand.uniform(91, (80,), -4.0, 2.0)).astype(np.float32)
    cfg = {"backbone": {"backbone_name": "darknet_21", "backbone_pretrained": ""}, "dataset": {"anchors": ANCHORS}, "yolo": {"classes": 80}}
    torch.manual_seed(3)
    model = YoloHead(cfg).to(dev())
    crit = YOLOForw(anchors=ANCHORS, num_classes=80, img_size=128, **opts).to(dev())
    x = torch.from_numpy(detrand.uniform(6, (2, 3, 128, 128), -2, 2)).to(dev())
    tg = [{"bbox": torch.from_numpy(b).to(dev()), "category_id": torch.from_numpy(l).to(dev())} for b, l in synth_targets(78, (4, 6), 80)]
    model.train()
    eng = model.engine
    # autograd through the modules vs the fused step
    for p in model.parameters():
        p.grad = None
    loss, _sub, _stats = crit(model(x), tg)
    loss.backward()
    g_auto = eng.flat_g.clone()
    out12 = eng.train_step(x, tg, crit)
    g1 = eng.flat_g.clone()
    np.testing.assert_allclose(out12[0].item(), loss.item(), rtol=1e-5)
    cos = float((g_auto.double() * g1.double()).sum() / (g_auto.double().norm() * g1.double().norm()))
    assert cos > 0.999, cos
    # loss scale: gradients x 1024
    S = 1024.0
    out12s = eng.train_step(x, tg, crit, grad_scale=S)
    gS = eng.flat_g.clone()
    np.testing.assert_allclose(out12s[0].item(), out12[0].item(), rtol=1e-6)        # the reported loss is not scaled
    rel = float((gS.double() / S - g1.double()).norm() / g1.double().norm())
    assert rel < 2e-3, rel                                                          # power-of-two scale: only atomic-order noise
    # the scaler's step with the scaled gradient == a plain step with the unscaled one
    w0 = eng.flat_w.clone()
    opt = FlatSGD.for_engine(eng, lr=1e-2, momentum=0.9)
    scaler = DynamicLossScaler(init_scale=S)
    assert scaler.step(opt) is True
    w_scaled = eng.flat_w.clone()
    eng.flat_w.copy_(w0)
    eng.flat_g.copy_(g1)
    FlatSGD.for_engine(eng, lr=1e-2, momentum=0.9).step()
    assert float((eng.flat_w - w_scaled).abs().max()) <= 1e-2 * 2e-3 * float(g1.abs().max()) + 1e-9
    # an overflowing gradient is skipped and halves the scale
    eng.flat_g.copy_(gS)
    eng.flat_g[123] = float("inf")
    before = eng.flat_w.clone()
    assert scaler.step(opt) is False and scaler.loss_scale == S / 2
    assert torch.equal(eng.flat_w, before)


def test_frozen_inference_matches_and_follows_weight_loads():
    """freeze_inference (what YoloHead.eval() switches on): the packs and folded BatchNorm rows are built once per evaluation loop;
    outputs are bit-identical to the per-batch rebuild, and a weight load or a training forward refreshes them."""
    eng, sd = make_engine("darknet_21", 5000, damp=0.2)
    x1 = torch.from_numpy(detrand.uniform(77, (2, 3, 96, 96), -2.0, 2.0)).to(dev())
    x2 = torch.from_numpy(detrand.uniform(78, (2, 3, 96, 96), -2.0, 2.0)).to(dev())
    base1 = [o.clone() for o in eng.forward(x1, training=False)]
    base2 = [o.clone() for o in eng.forward(x2, training=False)]
    eng.freeze_inference(True)
    for _ in range(2):
        for x, base in ((x1, base1), (x2, base2)):
            outs = eng.forward(x, training=False)
            assert all(torch.equal(o, b) for o, b in zip(outs, base))
    plan = eng._last_plan
    assert plan._const_epoch == eng._static_epoch > 0
    sd2 = {k: (v * 1.5 if k.endswith("conv_out.weight") else v) for k, v in sd.items()}
    eng.load_reference_state_dict(sd2)                     # un-freezes this epoch: the next forward re-packs
    outs = eng.forward(x1, training=False)
    assert not torch.equal(outs[0], base1[0])
    eng.freeze_inference(False)
    ref = eng.forward(x1, training=False)
    assert all(torch.equal(o, r) for o, r in zip(outs, ref))
    eng.freeze_inference(True)
    eng.forward(x1, training=False)
    e0 = eng._static_epoch
    eng.forward(x1, training=True)                         # a training forward (an optimizer step may follow) invalidates the frozen state
    assert eng._static_epoch > e0


@pytest.mark.gpu
def test_frozen_inference_over_an_epoch_loop():
    """The reference's epoch loop (train, validate, train, validate: yolo/main.py, test_one_epoch.py:10-16) through YoloHead.train() /
    .eval(): the second and later validation loops must see the weights and running statistics of the training steps in between.  The
    version counter is strictly monotonic - a cached eval plan of epoch 1 can never match the frozen state of epoch 2 (round-3 defect:
    freeze / unfreeze reset the counter, so eval 2 ran on the folded BatchNorm rows and bf16 packs of epoch 1)."""
    from object_detectors_amd.optim import FlatSGD
    from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
    from object_detectors_amd.yolo.nets.yolohead import YoloHead
    from tests.helpers import synth_targets
    cfg = {"backbone": {"backbone_name": "darknet_21", "backbone_pretrained": ""}, "dataset": {"anchors": ANCHORS}, "yolo": {"classes": 80}}
    torch.manual_seed(5)
    model = YoloHead(cfg).to(dev())
    crit = YOLOForw(anchors=ANCHORS, num_classes=80, img_size=96).to(dev())
    eng = model.engine
    x = torch.from_numpy(detrand.uniform(79, (2, 3, 96, 96), -2.0, 2.0)).to(dev())
    xe = torch.from_numpy(detrand.uniform(80, (2, 3, 96, 96), -2.0, 2.0)).to(dev())
    tg = [{"bbox": torch.from_numpy(b).to(dev()), "category_id": torch.from_numpy(l).to(dev())} for b, l in synth_targets(81, (3, 5), 80)]
    opt = FlatSGD.for_engine(eng, lr=1e-2, momentum=0.9)
    seen = []
    for epoch in range(3):
        model.train()
        assert eng._frozen is False
        eng.train_step(x, tg, crit)
        opt.step()
        model.eval()
        assert eng._frozen is True
        with torch.no_grad():
            a = [o.clone() for o in eng.forward(xe, training=False)]
            b = [o.clone() for o in eng.forward(xe, training=False)]      # second batch of the loop: served from the frozen constants
        assert all(torch.equal(u, v) for u, v in zip(a, b))
        # the same evaluation with nothing cached: a fresh, unfrozen rebuild of packs and folded BatchNorm rows
        v0 = eng._static_epoch
        eng.freeze_inference(False)
        ref = [o.clone() for o in eng.forward(xe, training=False)]
        assert eng._static_epoch > v0
        assert all(torch.equal(u, v) for u, v in zip(a, ref)), "epoch %d: frozen eval ran on stale constants" % epoch
        seen.append(a[0])
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])      # the weights did move between the epochs
