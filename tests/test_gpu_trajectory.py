"""Training-trajectory parity (the stand-in for the north star's "mAP within 0.1", which needs a dataset this pipeline does not have):
12 SGD steps of the reference recipe (yolo/procedures/train_one_epoch.py:58-96, initialize.py:37-45: SGD momentum 0.9, weight decay
5e-4) on darknet_21 / 128 px / batch 8, the same seeded batch sequence fed to

    E   the engine: YoloV3Engine.train_step (HIP forward, fused criterion, HIP backward) + FlatSGD (fused HIP optimizer), bf16 storage
    A   the oracle in plain fp32: oracle/net_oracle.py + oracle/yolo_oracle.py criterion + torch.optim.SGD          (= the reference at O0)
    B   the oracle with activations / packed weights rounded to bf16 (what the engine stores)
    C   the oracle with the same tensors rounded to fp16 (what the reference's apex-O2 recipe stores, yolo/batch_files/sample.txt:28-44)

Asserted: E tracks A per step (loss), in the BatchNorm running statistics and in the final weights, no worse than arm B's own distance
from A allows - i.e. the engine's trajectory error IS the storage format's, and arm C measures what bf16 costs against fp16 storage.
The table is printed (pytest -s) and written to gpurun_out/ for DESIGN.md."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import detrand, net_oracle  # noqa: E402
from oracle import yolo_oracle as yo  # noqa: E402
from tests.helpers import synth_targets  # noqa: E402

ANCHORS = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
BNAME, PX, BS, STEPS, LR = "darknet_21", 128, 8, 12, 1e-4


def _batch(step):
    x = detrand.uniform(9000 + step, (BS, 3, PX, PX), -2.0, 2.0)
    tg = synth_targets(9500 + 31 * step, [3, 1, 5, 2, 4, 2, 6, 3], 80)
    return x, tg


DAMP = [0.2]


def _state():
    sd = net_oracle.det_state(BNAME, 5000)
    for k in sd:
        if k.endswith(".bn2.weight"):
            sd[k] = sd[k] * DAMP[0]                   # damped residual branches: the random-weight net is otherwise chaotic
    return sd


def _oracle_run(quant):
    sd = {k: v.clone() for k, v in _state().items()}
    params = [v.requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var"))]
    opt = torch.optim.SGD(params, lr=LR, momentum=0.9, weight_decay=5e-4)
    spec = yo.YoloSpec(ANCHORS, 80, PX)
    losses = []
    for step in range(STEPS):
        x, tg = _batch(step)
        outs = net_oracle.forward(sd, torch.from_numpy(x), BNAME, training=True, quant=quant, update_running=True)
        res = yo.yolo_loss(spec, [o.detach().numpy() for o in outs], tg, want_grad=True)
        opt.zero_grad()
        torch.autograd.backward(list(outs), [torch.from_numpy(g) for g in res["grads"]])
        opt.step()
        losses.append(float(res["loss"]))
    return losses, {k: v.detach().clone() for k, v in sd.items()}


RECORD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tune_trajectory.json")


LOSS_SCALE = 1024.0      # fp16 arm: a constant power-of-two loss scale (apex's dynamic scaler settles on one; a power of two changes no mantissa)


def _engine_run(record=None, storage="bf16"):
    """record: raw tune-record bytes imported LOCKED before the engine is built (the plan build then times nothing it covers), or None = the
    plan build times its candidates on this box.  storage "fp16": the engine's fp16 arm, trained with a loss scale like the reference's apex
    recipe (train_one_epoch.py:88-94).  Returns (losses, final state, the record the run ended with)."""
    from object_detectors_amd import tune
    from object_detectors_amd.optim import FlatSGD
    from object_detectors_amd.yolo.nets.engine import YoloV3Engine
    from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
    tune.clear()
    if record is not None:
        tune.import_bytes(record, replace=True, lock=True)
    dev = torch.device("cuda:0")
    eng = YoloV3Engine(BNAME, 3, 80, device=dev, storage=storage)
    eng.load_reference_state_dict(_state())
    crit = YOLOForw(anchors=ANCHORS, num_classes=80, img_size=PX).to(dev)
    opt = FlatSGD.for_engine(eng, lr=LR, momentum=0.9, weight_decay=5e-4)
    S = LOSS_SCALE if storage == "fp16" else 1.0
    losses = []
    for step in range(STEPS):
        x, tg = _batch(step)
        t = [{"bbox": torch.from_numpy(b).to(dev), "category_id": torch.from_numpy(l).to(dev)} for b, l in tg]
        out12 = eng.train_step(torch.from_numpy(x).to(dev), t, crit, grad_scale=S)
        opt.step(grad_scale=1.0 / S)
        losses.append(float(out12[0]))
    torch.cuda.synchronize()
    used = tune.export_bytes()
    tune.clear()
    return losses, {k: v.detach().cpu() for k, v in eng.reference_state_dict().items()}, used


def _committed_record():
    from object_detectors_amd import tune
    if not os.path.exists(RECORD):
        return None
    with open(RECORD) as f:
        return tune.loads(f.read())


_ORACLE = {}


def _oracle_arm(name):
    if name not in _ORACLE:
        q = {"fp32": None, "bf16": lambda t: t.bfloat16().float(), "fp16": lambda t: t.half().float()}[name]
        _ORACLE[name] = _oracle_run(q)
    return _ORACLE[name]


def _cmp(sd, ref):
    """cosine / relative distance over all conv + BN parameters, and over the BatchNorm running statistics."""
    pw = [k for k in ref if ref[k].dtype == torch.float32 and not k.endswith(("running_mean", "running_var"))]
    a = torch.cat([sd[k].double().reshape(-1) for k in pw])
    b = torch.cat([ref[k].double().reshape(-1) for k in pw])
    rs = [k for k in ref if k.endswith(("running_mean", "running_var"))]
    ra = torch.cat([sd[k].double().reshape(-1) for k in rs])
    rb = torch.cat([ref[k].double().reshape(-1) for k in rs])
    return {"weights_cos": float((a * b).sum() / (a.norm() * b.norm())), "weights_rel": float((a - b).norm() / b.norm()),
            "running_rel": float((ra - rb).norm() / rb.norm())}


def _update_cmp(sd, ref, init):
    """the same for the UPDATE (final - initial weights): the part of the state the 12 steps actually produced"""
    pw = [k for k in ref if ref[k].dtype == torch.float32 and not k.endswith(("running_mean", "running_var"))]
    a = torch.cat([(sd[k].double() - init[k].double()).reshape(-1) for k in pw])
    b = torch.cat([(ref[k].double() - init[k].double()).reshape(-1) for k in pw])
    return {"update_cos": float((a * b).sum() / (a.norm() * b.norm())), "update_rel": float((a - b).norm() / b.norm())}


def test_twelve_sgd_steps_track_the_fp32_oracle():
    from object_detectors_amd import tune
    init = _state()
    lossA, sdA = _oracle_arm("fp32")
    lossB, sdB = _oracle_arm("bf16")
    lossC, sdC = _oracle_arm("fp16")
    rec = _committed_record()
    lossE, sdE, used = _engine_run(rec)
    lossH, sdH, usedH = _engine_run(rec if rec is not None else used, storage="fp16")      # the engine with fp16 storage (its choices have their own keys)
    assert all(np.isfinite(lossH)), lossH
    used = usedH                                                                              # (a superset of the bf16 run's record)
    os.makedirs("gpurun_out", exist_ok=True)
    if rec is None:
        tune.save(os.path.join("gpurun_out", "tune_trajectory.json"), used)      # to be committed as tests/golden/tune_trajectory.json
    else:
        # the plan build honoured the locked record: every choice of the record is unchanged, nothing it covers was timed again
        have = dict(((t, k), v) for t, k, v in tune.to_entries(used))
        for t, k, v in tune.to_entries(rec):
            assert have[(t, k)] == v, (t, k, v, have[(t, k)])
        missing = sorted(set(have) - set((t, k) for t, k, _v in tune.to_entries(rec)))
        assert not missing, "shapes of this test that tests/golden/tune_trajectory.json does not cover (regenerate it): %r" % (missing,)
    rel = lambda l: [abs(a - b) / abs(b) for a, b in zip(l, lossA)]
    table = {"config": f"{BNAME} {PX}px bs{BS}, {STEPS} SGD steps lr {LR} momentum 0.9 wd 5e-4, residual BN gammas x0.2",
             "tune_record": "tests/golden/tune_trajectory.json (locked)" if rec is not None else "timed on this box",
             "loss_fp32_oracle": [round(v, 4) for v in lossA],
             "loss_engine": [round(v, 4) for v in lossE], "loss_engine_fp16": [round(v, 4) for v in lossH], "loss_bf16_oracle": [round(v, 4) for v in lossB], "loss_fp16_oracle": [round(v, 4) for v in lossC],
             "loss_rel_err": {"engine": [round(v, 4) for v in rel(lossE)], "engine_fp16": [round(v, 4) for v in rel(lossH)], "bf16_oracle": [round(v, 4) for v in rel(lossB)],
                              "fp16_oracle": [round(v, 4) for v in rel(lossC)]},
             "final_state_vs_fp32_oracle": {"engine": {**_cmp(sdE, sdA), **_update_cmp(sdE, sdA, init)},
                                            "engine_fp16": {**_cmp(sdH, sdA), **_update_cmp(sdH, sdA, init)},
                                            "bf16_oracle": {**_cmp(sdB, sdA), **_update_cmp(sdB, sdA, init)},
                                            "fp16_oracle": {**_cmp(sdC, sdA), **_update_cmp(sdC, sdA, init)}},
             "engine_vs_bf16_oracle": {**_cmp(sdE, sdB), **_update_cmp(sdE, sdB, init)}}
    print("trajectory:", json.dumps(table))
    with open(os.path.join("gpurun_out", "trajectory_parity.json"), "w") as f:
        json.dump(table, f, indent=1)
    # the loss falls, on every arm (twelve steps at lr 1e-4 from a random initialisation: ~20 %)
    assert lossA[-1] < 0.9 * lossA[0] and lossE[-1] < 0.9 * lossE[0]
    eE, eB, eC = rel(lossE), rel(lossB), rel(lossC)
    fin = table["final_state_vs_fp32_oracle"]
    E, B = fin["engine"], fin["bf16_oracle"]
    # This random-weight net amplifies ANY rounding of its activations: rounding the stored tensors to fp16 - the reference's
    # apex-O2 recipe - already moves single steps by 6-11 %.  Round 3 saw the engine's worst step between 5.8 and 13.7 % on three boxes with
    # identical code, because the plan build chose tile configurations / split counts by timing.  Since round 4 the choices are DATA (the
    # locked record above) and every kernel of the step is fixed-order, so this run is the same on every box
    # (test_same_record_same_trajectory below asserts bit-identity and measures what another record moves): the single-step bar is back.
    assert max(eE) < 0.12, (eE, eB, eC)
    mean = lambda v: sum(v) / len(v)
    assert mean(eE) < 0.05 and mean(eE) < max(mean(eB), mean(eC)) + 0.02, (eE, eB, eC)
    assert E["weights_cos"] > 0.9999 and E["weights_rel"] < B["weights_rel"] * 1.3 + 1e-3, fin
    assert E["running_rel"] < 0.03 and E["running_rel"] < B["running_rel"] + 0.01, fin
    assert E["update_cos"] > 0.5 and E["update_cos"] > B["update_cos"] - 0.1, fin
    # the fp16-storage engine against the fp16-storage oracle arm, same bars
    eH, Hh, Cc = rel(lossH), fin["engine_fp16"], fin["fp16_oracle"]
    assert max(eH) < 0.14 and mean(eH) < 0.05 and mean(eH) < max(mean(eB), mean(eC)) + 0.02, (eH, eC)
    assert Hh["weights_cos"] > 0.9999 and Hh["weights_rel"] < Cc["weights_rel"] * 1.3 + 1e-3, fin
    assert Hh["running_rel"] < 0.03 and Hh["running_rel"] < Cc["running_rel"] + 0.01, fin
    assert Hh["update_cos"] > 0.5 and Hh["update_cos"] > Cc["update_cos"] - 0.1, fin


def _alt_record(raw):
    """Another legal set of choices for the same shapes: a different wide tile configuration, the other stride-2 data-gradient form, half
    the weight-gradient split count (the library falls back to the nearest valid split)."""
    from object_detectors_amd import tune
    out = []
    for t, k, v in tune.to_entries(raw):
        if t == "igemm":
            v2 = v if v in (0, 29, 30, 31) else (2 if v == 1 else 1)
        elif t == "s2cat":
            v2 = 1 - v
        else:
            v2 = max(1, v // 2)
        out.append((t, k, v2))
    return tune.from_entries(out)


def test_same_record_same_trajectory():
    """"same choices => same trajectory", shown on ONE box.  Two runs under the same locked record are BIT-identical
    (losses of all twelve steps and every final tensor: the kernels of the step are fixed-order, the BatchNorm-backward sums included since
    round 4); a run under a DIFFERENT record (other tiles / forms / split counts, i.e. other summation orders) is a different trajectory,
    and the table records by how much - that spread, not the code, is what moved round 3's worst step between boxes."""
    from object_detectors_amd import tune
    lossA, _sdA = _oracle_arm("fp32")
    rec = _committed_record()
    if rec is None:
        _l, _s, rec = _engine_run(None)
    l1, s1, u1 = _engine_run(rec)
    l2, s2, u2 = _engine_run(rec)
    assert u1 == u2
    assert l1 == l2, (l1, l2)
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k
    alt = _alt_record(rec)
    l3, s3, _u3 = _engine_run(alt)
    rel = lambda l: [abs(a - b) / abs(b) for a, b in zip(l, lossA)]
    between = [abs(a - b) / abs(b) for a, b in zip(l3, l1)]
    n_changed = sum(1 for a, b in zip(tune.to_entries(rec), tune.to_entries(alt)) if a != b)
    table = {"record_entries": len(tune.to_entries(rec)), "entries_changed_in_alt": n_changed,
             "loss_record": [round(v, 4) for v in l1], "loss_alt_record": [round(v, 4) for v in l3],
             "rel_err_vs_fp32_record": [round(v, 4) for v in rel(l1)], "rel_err_vs_fp32_alt": [round(v, 4) for v in rel(l3)],
             "rel_diff_between_records": [round(v, 5) for v in between],
             "worst_step_vs_fp32": {"record": round(max(rel(l1)), 4), "alt": round(max(rel(l3)), 4)},
             "same_record_twice_bit_identical": True}
    print("tune sensitivity:", json.dumps(table))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "trajectory_tune_sensitivity.json"), "w") as f:
        json.dump(table, f, indent=1)
    assert max(rel(l3)) < 0.2                                   # any legal record stays a valid training run
