"""fp16 storage (the reference's apex-O2 format, yolo/batch_files/sample.txt:28-44, yolo/procedures/initialize.py:44-45):
the *_f16 twins of the convolution-engine entry points (include/mi355det_f16.h - the same sources compiled with -DMI355_F16=1) and
YoloV3Engine(storage="fp16").

Per-layer checks compare with a plain PyTorch fp32 evaluation of the same op on the same fp16-rounded operands (tolerance 4e-3 of max: fp16
has 11 significand bits against bf16's 8, so the bar is 5x tighter than the bf16 tests' 2e-2); the engine checks mirror
tests/test_gpu_engine.py against oracle/net_oracle.py with fp16-rounded storage; the loss-scaled training step follows apex's pattern
(train_one_epoch.py:88-94)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

from oracle import detrand, net_oracle  # noqa: E402

ANCHORS = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
TOL = 4e-3


def dev():
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy(detrand.uniform(seed, shape, -1.0, 1.0)).float() * scale


def nhwc16(t):
    return t.permute(0, 2, 3, 1).contiguous().to(dev()).half()


def q(t):
    return t.half().float()


@pytest.mark.parametrize("case", [(2, 24, 24, 128, 256, 3, 1), (2, 16, 16, 64, 128, 3, 2), (3, 13, 13, 256, 128, 1, 1), (2, 26, 26, 32, 64, 3, 2),
                                  (1, 40, 40, 256, 512, 3, 1), (2, 20, 20, 512, 256, 1, 1), (2, 32, 32, 64, 32, 1, 1), (1, 20, 20, 512, 1024, 3, 1)])
def test_conv_fwd_dgrad_wgrad_fp16(case):
    """mi355det_conv_{fwd,dgrad,wgrad}_f16 (+ BN partial statistics) against torch fp32 on the same fp16 operands; every tile configuration
    the tuner may pick for the shape (mi355det_conv_autotune_mode_f16 times them and the tuned choice is used afterwards)."""
    from object_detectors_amd import ops
    n, h, w, cin, cout, k, s = case
    x = q(rnd((n, cin, h, w), 1))
    wt = q(rnd((cout, cin, k, k), 2, (2.0 / (cin * k * k)) ** 0.5))
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, stride=s, padding=k // 2)
    gy = q(rnd(tuple(y_ref.shape), 3))
    y_ref.backward(gy)
    with ops.storage("fp16"):
        shape = ops.conv_shape(n, h, w, cin, cout, k, s)
        wf, wd = ops.pack_weights(shape, wt.to(dev()))
        assert wf.dtype == torch.float16 and wd.dtype == torch.float16
        xd, gyd = nhwc16(x), nhwc16(gy)
        rows = ops.conv_stats_rows(shape)
        for tuned in (False, True):
            if tuned:
                ops.lib().mi355det_conv_autotune_mode(1)
            y = torch.full((n, shape.ho, shape.wo, cout), 5.0, device=dev(), dtype=torch.float16)
            stats = torch.zeros((rows + 64, 2, ops.cout_pad_of(cout)), device=dev())
            dx = torch.full((n, h, w, cin), 5.0, device=dev(), dtype=torch.float16)
            try:
                ops.conv_fwd(shape, xd, wf, y, stats=stats)
                ops.conv_dgrad(shape, gyd, wd, dx)
            finally:
                ops.lib().mi355det_conv_autotune_mode(0)
            if tuned:      # the tuning pass leaves valid outputs of the chosen configuration; run once more outside the mode
                ops.conv_fwd(shape, xd, wf, y, stats=stats)
                ops.conv_dgrad(shape, gyd, wd, dx)
            dw = torch.zeros((cout, k, k, cin), device=dev())
            ops.conv_wgrad(shape, xd, gyd, dw)
            torch.cuda.synchronize()
            yr = y_ref.detach().permute(0, 2, 3, 1)
            assert float((y.float().cpu() - yr).abs().max()) <= TOL * float(yr.abs().max())
            gr = xr.grad.permute(0, 2, 3, 1)
            assert float((dx.float().cpu() - gr).abs().max()) <= TOL * float(gr.abs().max())
            dwr = wr.grad.permute(0, 2, 3, 1)
            assert float((dw.cpu() - dwr).abs().max()) <= 1e-3 * float(dwr.abs().max())          # fp32 accumulation of exact fp16 products
            yf = y.float().reshape(-1, cout).double()
            st = stats[:rows].double().sum(0)
            torch.testing.assert_close(st[0, :cout], yf.sum(0), rtol=1e-3, atol=1e-2)
            torch.testing.assert_close(st[1, :cout], (yf * yf).sum(0), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("c,pixels,res", [(32, 777, False), (64, 1000, True), (256, 300, False)])
def test_bn_passes_and_elementwise_fp16(c, pixels, res):
    """bn_act_fwd / bn_act_bwd_reduce_det / bn_act_bwd_apply / add / upsample2x on fp16 tensors against torch autograd in fp32."""
    from object_detectors_amd import _lib
    from object_detectors_amd._lib import check, ptr, stream_ptr
    L = _lib.storage_lib("fp16")
    L0 = _lib.lib()
    d = dev()
    z = q(rnd((pixels, c), 11))
    r = q(rnd((pixels, c), 12))
    g = q(rnd((pixels, c), 13, 0.05))
    gamma, beta = 1.0 + 0.5 * rnd((c,), 14), 0.2 * rnd((c,), 15)
    zr, gr_, br_ = z.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    zz = zr.t().reshape(1, c, pixels, 1)
    y = F.leaky_relu(F.batch_norm(zz, torch.zeros(c), torch.ones(c), gr_, br_, True, 0.1, 1e-5), 0.1)
    out_ref = y.reshape(c, pixels).t() + (r if res else 0)
    out_ref.backward(g)
    zd = z.to(d).half()
    part = torch.zeros(3, 2, c, device=d)
    part[0, 0] = z.double().sum(0).float().to(d)
    part[1, 1] = (z.double() ** 2).sum(0).float().to(d)
    ss = torch.empty(4 * c, device=d)
    rm, rv = torch.zeros(c, device=d), torch.ones(c, device=d)
    gam_d, bet_d = gamma.to(d), beta.to(d)
    check(L0.mi355det_bn_finalize(ptr(part), 3, c, c, pixels, ptr(gam_d), ptr(bet_d), 1e-5, 0.1, ptr(rm), ptr(rv), ptr(ss), stream_ptr()))
    out = torch.empty(pixels, c, dtype=torch.float16, device=d)
    rd = r.to(d).half() if res else None
    check(L.mi355det_bn_act_fwd(ptr(zd), c, ptr(ss), c, pixels, 0.1, ptr(rd), c, ptr(out), c, stream_ptr()))
    assert float((out.float().cpu() - out_ref.detach()).abs().max()) < TOL * float(out_ref.abs().max())
    gd = g.to(d).half()
    nb = L.mi355det_bn_act_bwd_reduce_workspace(c, pixels)
    ws = torch.empty(nb, dtype=torch.uint8, device=d)
    sums = torch.empty(2 * c, device=d)
    check(L.mi355det_bn_act_bwd_reduce_det(ptr(gd), c, None, 0, ptr(zd), c, ptr(ss), c, pixels, 0.1, ptr(sums), ptr(ws), nb, stream_ptr()))
    dz = torch.empty(pixels, c, dtype=torch.float16, device=d)
    dg, db = torch.zeros(c, device=d), torch.zeros(c, device=d)
    check(L.mi355det_bn_act_bwd_apply(ptr(gd), c, None, 0, ptr(zd), c, ptr(ss), ptr(sums), None, c, pixels, 0.1, ptr(dz), c, ptr(dg), ptr(db),
                                      stream_ptr()))
    assert float((dz.float().cpu() - zr.grad).abs().max()) < TOL * float(zr.grad.abs().max())
    np.testing.assert_allclose(dg.cpu(), gr_.grad, rtol=5e-3, atol=5e-3 * float(gr_.grad.abs().max()))
    np.testing.assert_allclose(db.cpu(), br_.grad, rtol=5e-3, atol=5e-3 * float(br_.grad.abs().max()))
    # add: exact sum of two fp16 values rounded once
    o2 = torch.empty_like(out)
    check(L.mi355det_add_bf16(ptr(zd), c, ptr(gd), c, c, pixels, ptr(o2), c, stream_ptr()))
    assert torch.equal(o2.cpu(), (z + g).half())


def _engine_and_oracle(bname, seed, storage):
    from object_detectors_amd.yolo.nets.engine import YoloV3Engine
    sd = net_oracle.det_state(bname, seed)
    for k in sd:
        if k.endswith(".bn2.weight"):
            sd[k] = sd[k] * 0.2
    eng = YoloV3Engine(bname, 3, 80, device=dev(), storage=storage)
    eng.load_reference_state_dict(sd)
    return eng, sd


def test_engine_fp16_forward_and_gradients_track_the_fp16_storage_oracle():
    """darknet_21, 128 px, batch 4, damped residual branches: the fp16 engine's heads and parameter gradients against the fp32
    oracle, bounded by what fp16-rounded storage costs the oracle itself (and at least as close as the bf16 engine on the same inputs)."""
    bname = "darknet_21"
    x = torch.from_numpy(detrand.uniform(41, (4, 3, 128, 128), -2.0, 2.0))
    errs = {}
    for storage in ("fp16", "bf16"):
        eng, sd = _engine_and_oracle(bname, 5000, storage)
        outs = [o.float().cpu() for o in eng.forward(x.to(dev()), training=True)]
        ref = net_oracle.forward({k: v.clone() for k, v in sd.items()}, x, bname, training=True, quant=None, update_running=False)
        qref = net_oracle.forward({k: v.clone() for k, v in sd.items()}, x, bname, training=True,
                                  quant=(lambda t: t.half().float()) if storage == "fp16" else (lambda t: t.bfloat16().float()), update_running=False)
        e_eng = [float((o - r.detach()).abs().max() / r.detach().abs().max()) for o, r in zip(outs, ref)]
        e_orc = [float((o.detach() - r.detach()).abs().max() / r.detach().abs().max()) for o, r in zip(qref, ref)]
        errs[storage] = (e_eng, e_orc)
        # head gradients of a fixed pattern -> parameter gradients vs autograd through the fp32 oracle
        hg = [torch.from_numpy(detrand.uniform(50 + i, tuple(o.shape), -1.0, 1.0)) * 1e-2 for i, o in enumerate(outs)]
        S = 256.0 if storage == "fp16" else 1.0          # loss scale: keeps the activation gradients in fp16's normal range
        eng.backward([g.to(dev()) * S for g in hg])
        torch.cuda.synchronize()
        g_eng = {k: v.double().cpu() / S for k, v in eng.reference_state_dict(grads=True).items()}
        sdr = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")) else v.clone())
               for k, v in sd.items()}
        outs_r = net_oracle.forward(sdr, x, bname, training=True, quant=None, update_running=False)
        torch.autograd.backward(list(outs_r), hg)
        keys = [k for k in g_eng if sdr[k].grad is not None]
        a = torch.cat([g_eng[k].reshape(-1) for k in keys])
        r = torch.cat([sdr[k].grad.double().reshape(-1) for k in keys])
        errs[storage + "_g"] = (float((a * r).sum() / (a.norm() * r.norm())), float((a - r).norm() / r.norm()), a)
    (e16, o16), (eb, ob) = errs["fp16"], errs["bf16"]
    print("fp16 engine / fp16 oracle / bf16 engine / bf16 oracle head errors:", e16, o16, eb, ob)
    for a, b in zip(e16, o16):
        assert a <= 1.5 * b + 0.01, (e16, o16)
    assert max(e16) < max(eb), (e16, eb)                # three more mantissa bits must show
    # parameter gradients against autograd through the fp32 oracle: fp16 storage is the closer of the two formats
    (c16, r16, g16), (cb, rb, gb) = errs["fp16_g"], errs["bf16_g"]
    print("gradient cos / rel vs fp32 autograd: fp16 %.5f / %.4f, bf16 %.5f / %.4f" % (c16, r16, cb, rb))
    # (measured on MI355X: fp16 0.979 / 0.205, bf16 0.857 / 0.533 - this random-weight net amplifies storage rounding)
    assert c16 > 0.95 and c16 > cb + 0.05 and r16 < 0.6 * rb, (c16, r16, cb, rb)
    assert float((g16 * gb).sum() / (g16.norm() * gb.norm())) > 0.8


def test_engine_fp16_train_step_with_dynamic_loss_scale():
    """apex's pattern on the fp16 engine: scaled criterion gradient -> HIP backward -> guarded optimizer step with 1/S; an overflow (inf in
    the gradient) skips the step and halves the scale; the loss falls over a few steps."""
    from object_detectors_amd.optim import DynamicLossScaler, FlatSGD
    from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
    from tests.helpers import synth_targets
    eng, _sd = _engine_and_oracle("darknet_21", 5000, "fp16")
    crit = YOLOForw(anchors=ANCHORS, num_classes=80, img_size=128).to(dev())
    opt = FlatSGD.for_engine(eng, lr=1e-4, momentum=0.9, weight_decay=5e-4)
    scaler = DynamicLossScaler(init_scale=1024.0)
    x = torch.from_numpy(detrand.uniform(9000, (8, 3, 128, 128), -2.0, 2.0)).to(dev())
    tg = [{"bbox": torch.from_numpy(b).to(dev()), "category_id": torch.from_numpy(l).to(dev())} for b, l in synth_targets(9500, [3, 1, 5, 2, 4, 2, 6, 3], 80)]
    losses = []
    for _ in range(6):
        out12 = eng.train_step(x, tg, crit, grad_scale=scaler.loss_scale)
        assert scaler.step(opt) is True
        losses.append(float(out12[0]))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    eng.train_step(x, tg, crit, grad_scale=scaler.loss_scale)
    eng.flat_g[7] = float("inf")
    before, s0 = eng.flat_w.clone(), scaler.loss_scale
    assert scaler.step(opt) is False and scaler.loss_scale == s0 / 2 and torch.equal(eng.flat_w, before)


def test_yolohead_maps_apex_opt_to_storage():
    from object_detectors_amd.yolo.nets.yolohead import YoloHead
    cfg = {"backbone": {"backbone_name": "darknet_21", "backbone_pretrained": ""}, "dataset": {"anchors": ANCHORS}, "yolo": {"classes": 80}}
    assert YoloHead(dict(cfg)).engine.storage == "bf16"
    m = YoloHead(dict(cfg, apex_opt="O2")).to(dev())
    assert m.engine.storage == "fp16"
    m.eval()
    with torch.no_grad():
        outs = m(torch.from_numpy(detrand.uniform(3, (2, 3, 96, 96), -2.0, 2.0)).to(dev()))
    assert all(torch.isfinite(o).all() for o in outs)
    assert YoloHead(dict(cfg, apex_opt="O2", storage="bf16")).engine.storage == "bf16"
