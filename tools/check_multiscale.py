"""Multi-scale sanity of the YOLOv3 training step (train_one_epoch.py:15-26,64-69 picks a new size every 10 batches): one step per size with
the round-3 fused kernels, the same step with them switched off (subprocess), losses side by side.   python tools/check_multiscale.py"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SIZES = [320, 416, 480, 608]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from object_detectors_amd.yolo.nets.engine import YoloV3Engine
    from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw
    from oracle import detrand
    from tests.helpers import synth_targets
    dev = torch.device("cuda:0")
    anchors = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
    eng = YoloV3Engine("darknet_53", 3, 80, device=dev)
    out = {}
    for px in SIZES:
        crit = YOLOForw(anchors=anchors, num_classes=80, img_size=px).to(dev)
        x = torch.from_numpy(detrand.uniform(100 + px, (4, 3, px, px), -2, 2)).to(dev)
        tg = synth_targets(200 + px, [3, 1, 5, 2], 80)
        t = [{"bbox": torch.from_numpy(b).to(dev), "category_id": torch.from_numpy(l).to(dev)} for b, l in tg]
        o = eng.train_step(x, t, crit)
        torch.cuda.synchronize()
        g = eng.flat_g
        out[px] = [float(o[0]), float(g.norm()), bool(torch.isfinite(g).all())]
    print("RESULT " + json.dumps(out))
else:
    res = {}
    # (round 3 compared the fused stem / one-pass stem backward / one-launch finalisation with their unfused forms through environment knobs;
    #  the losers are gone, so the two arms are now the two storage formats: the same sizes must train in both)
    for tag, env in (("fused", {}), ("unfused", {"MI355DET_STORAGE": "fp16"})):
        e = dict(os.environ); e.update(env)
        p = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        assert line, p.stderr[-2000:]
        res[tag] = json.loads(line[0][7:])
    ok = True
    for px in SIZES:
        a, b = res["fused"][str(px)], res["unfused"][str(px)]
        rel_l, rel_g = abs(a[0] - b[0]) / abs(b[0]), abs(a[1] - b[1]) / b[1]
        print(f"{px} px: loss {a[0]:.4f} / {b[0]:.4f} (rel {rel_l:.2e})   |grad| {a[1]:.4f} / {b[1]:.4f} (rel {rel_g:.2e})   finite {a[2]} / {b[2]}")
        ok &= a[2] and b[2] and rel_l < 2e-2 and rel_g < 5e-2
    print("multi-scale check:", "ok" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
