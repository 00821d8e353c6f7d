#!/usr/bin/env python3
"""Headline benchmark: images/sec of the YOLOv3 (Darknet-53) training step, forward + backward
(+ gradient all-reduce when N > 1), 640x640 synthetic COCO-shaped batches, per-GPU batch 32.

    python bench.py --gpus N --steps 10 --warmup 3          (N > 1: starts its own N ranks as child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement).  `roofline` prices the dominant kernel
(the MFMA implicit-GEMM forward convolution) from HIP events recorded live around its launches in the
timed steps; `cpu_baseline` times the CPU restatement (oracle/, kind "port") on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ANCHORS = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def synth_batch(bs, px, rank, device, m=7, classes=80):
    """SURVEY §8d synthetic inputs: images randn, M GT/img, xc,yc~U(.2,.8), w,h~U(.02,.32), class U{0..79}."""
    g = torch.Generator().manual_seed(rank)
    imgs = torch.randn((bs, 3, px, px), generator=g)
    targets = []
    for _ in range(bs):
        xy = torch.rand((m, 2), generator=g) * 0.6 + 0.2
        wh = torch.rand((m, 2), generator=g) * 0.30 + 0.02
        lab = torch.randint(0, classes, (m,), generator=g)
        targets.append({"bbox": torch.cat([xy, wh], 1).to(device), "category_id": lab.to(device)})
    return imgs.to(device), targets


def conv_fwd_flops(plan):
    """(index in plan.fwd, algorithmic FLOPs, name) of every forward-convolution launch (2*M*Cout*K, true channel counts).  The stem
    (3->32, K = 27) is computed by the two recompute kernels of csrc/stem_kernels.hip: both launches are timed, its FLOPs count once."""
    from object_detectors_amd._lib import lib
    L = lib()
    recs = [r for r in plan.ops if r["kind"] in ("cbl", "out")]
    out, k = [], 0
    stem_fl = 2.0 * plan.n * plan.H * plan.W * 32 * 27
    for i, (f, _a) in enumerate(plan.fwd):
        if f is L.mi355det_conv_fwd or f is L.mi355det_stem_l1_fwd:
            r = recs[k]
            k += 1
            s = r["spec"]
            shp = r["shp"]
            fl = 2.0 * shp.n * shp.ho * shp.wo * s.cout * s.cin * s.k * s.k
            # the fused launch computes the stem's convolution (again) and layer1.ds_conv: the algorithmic FLOPs of both count once
            out.append((i, fl + (stem_fl if f is L.mi355det_stem_l1_fwd else 0.0), r["name"] + (" + backbone.conv1" if f is L.mi355det_stem_l1_fwd else "")))
        elif f is L.mi355det_stem_fwd_stats:
            out.append((i, 0.0, "backbone.conv1 (statistics pass)"))
        elif f is L.mi355det_stem_fwd_apply:
            out.append((i, stem_fl, "backbone.conv1"))
    return out


def pmc_traffic(batch, px):
    """HBM bytes per forward-conv launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    runs of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; tools/pmc_summary.py).
    PMC counters cannot be read from inside the process, so the figure is the one measured for bs=32 / 640 px."""
    path = os.path.join(ROOT, "profiles", "r04_pmc_summary.json")
    if batch != 32 or px != 640 or not os.path.exists(path):
        return None, None, None
    d = json.load(open(path))
    n = b = 0.0
    for fam in ("igemm fwd (BN partial stats)", "igemm fwd (fp32 head)", "stem_l1_kernel", "stem_kernel<0>", "stem_kernel<1>"):
        if fam in d and d[fam]["read_MB_per_launch"] is not None and d[fam]["write_MB_per_launch"] is not None:
            n += d[fam]["launches_per_step"]
            b += d[fam]["launches_per_step"] * (d[fam]["read_MB_per_launch"] + d[fam]["write_MB_per_launch"]) * 1e6
    mu = d.get("igemm fwd (BN partial stats)", {}).get("mfma_util_pct")
    return (round(b / n) if n else None), "profiles/r04_pmc_summary.md", (round(mu, 1) if mu is not None else None)


def cpu_baseline(px, sample_bs=2, warmup=1, iters=5):
    """SURVEY 8(d) protocol: the CPU restatement of the SAME step - oracle/net_oracle.py (torch-fp32 Darknet-53 + YoloHead, forward and
    autograd backward) driven by the criterion oracle/yolo_oracle.py:yolo_loss (assignment, six loss terms, head gradients) - on the host
    cores of this box: `warmup` + `iters` iterations on a bounded sample of the workload (`sample_bs` images of the same size and GT
    density), median.  Second leg: the box operations the north star names - pairwise box_iou of the ground truth against the
    25 200 anchors, the Matcher, batched_nms of 2 000 candidates per image - on the C/OpenMP restatement oracle/c/box_oracle.c (all host
    threads), reported next to the step.  Baseline only: a large GPU/CPU ratio says nothing about kernel quality."""
    from oracle import net_oracle
    from oracle import yolo_oracle as yo
    torch.manual_seed(0)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = torch.get_num_threads()
    sd = net_oracle.det_state("darknet_53", 5000)
    for k, v in sd.items():
        if v.dtype == torch.float32:
            v.requires_grad_(not k.endswith(("running_mean", "running_var")))
    imgs, targets = synth_batch(sample_bs, px, 0, torch.device("cpu"))
    tg = [(t["bbox"].numpy(), t["category_id"].numpy()) for t in targets]
    spec = yo.YoloSpec(ANCHORS, 80, px)
    times = []
    for it in range(warmup + iters):
        t0 = time.perf_counter()
        outs = net_oracle.forward(sd, imgs, "darknet_53", training=True)
        res = yo.yolo_loss(spec, [o.detach().numpy() for o in outs], tg, want_grad=True)        # the criterion and its head gradients
        torch.autograd.backward(list(outs), [torch.from_numpy(g) for g in res["grads"]])
        for v in sd.values():
            v.grad = None
        times.append(time.perf_counter() - t0)
    timed = sorted(times[warmup:])
    t = timed[len(timed) // 2]
    out = {"value": round(sample_bs / t, 4), "unit": "images/s", "cores": cores, "torch_threads": threads, "kind": "port",
           "sample": f"{warmup} warm-up + {iters} timed iterations of fwd + criterion + bwd on {sample_bs} image(s) at {px}px, 7 GT/img "
                     f"(oracle/net_oracle.py torch-CPU fp32 + oracle/yolo_oracle.py numpy criterion), median; {sum(times):.0f} s of CPU work"}
    try:
        out["box_ops"] = cpu_box_leg(px)
    except Exception as e:      # the C oracle is test infrastructure: a box without gcc / OpenMP reports why instead of failing the bench
        out["box_ops"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def cpu_box_leg(px, images=8, reps=5):
    """The box operations of the path on oracle/c/box_oracle.c (C + OpenMP): per image box_iou(7 GT x 25 200 anchors) + Matcher(0.5, 0.4) +
    batched_nms(2 000 candidates, 80 classes, IoU 0.5).  -> images/s over `images` images, best of `reps`."""
    import numpy as np
    from oracle import box_oracle_c as bc
    from oracle import detrand
    bc.build()
    n_anchor = sum(3 * (px // s) ** 2 for s in (32, 16, 8))
    data = []
    for i in range(images):
        c = detrand.uniform(100 + i, (n_anchor, 2), 0, px)
        wh = detrand.uniform(200 + i, (n_anchor, 2), 8, px / 2)
        anchors = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        gc = detrand.uniform(300 + i, (7, 2), 0.2 * px, 0.8 * px)
        gwh = detrand.uniform(400 + i, (7, 2), 0.02 * px, 0.32 * px)
        gt = np.concatenate([gc - gwh / 2, gc + gwh / 2], 1).astype(np.float32)
        cand = anchors[:2000] + detrand.uniform(500 + i, (2000, 4), -2, 2).astype(np.float32)
        scores = detrand.uniform(600 + i, (2000,), 0, 1).astype(np.float32)
        idxs = detrand.randint(700 + i, (2000,), 0, 80)
        data.append((gt, anchors, cand, scores, idxs))
    best = 1e30
    kept = 0
    for _ in range(reps):
        t0 = time.perf_counter()
        for gt, anchors, cand, scores, idxs in data:
            q = bc.box_iou(gt, anchors)
            bc.matcher(q, 0.5, 0.4, True)
            kept = len(bc.batched_nms(cand, scores, idxs, 0.5))
        best = min(best, time.perf_counter() - t0)
    return {"value": round(images / best, 2), "unit": "images/s", "kind": "port (oracle/c/box_oracle.c, C + OpenMP)",
            "sample": f"box_iou 7 x {n_anchor} + Matcher + batched_nms of 2000 candidates per image, {images} images, best of {reps}; last keep count {kept}"}


def launch_ranks(n):
    """Start `python -m torch.distributed.run --nproc-per-node n bench.py <same arguments>` as a child process on a free local port,
    relay its output (rank 0 prints the JSON line) and return its exit code: a failed rank makes the whole run fail."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.run(cmd, env=env).returncode


def launcher_selftest(rank, world, mode):
    """CPU rehearsal of the launcher path (tests/test_bench_launcher.py): every rank joins a gloo group and sums its rank; rank 0
    prints one JSON line.  mode "fail": the last rank exits non-zero, which the launcher must report."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "rank_sum": float(t.item())}), flush=True)
    dist.destroy_process_group()
    return 3 if (mode == "fail" and rank == world - 1) else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--px", type=int, default=640)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune-record", default="auto",
                    help="tune record to load (locked) before the plan build: a path, 'none' (time everything on this box), or 'auto' = "
                         "object_detectors_amd/tune_records/yolov3_d53_bs<batch>_<px>_<storage>.json when it exists (the committed step-level record: the "
                         "same kernels, summation orders and choices on every box and rank; tools/tune_step.py)")
    ap.add_argument("--storage", choices=["bf16", "fp16"], default="bf16",
                    help="format of stored activations / gradients / packed weights (fp16: the reference's apex-O2 format, with a loss scale of 1024)")
    ap.add_argument("--atomic-bn-sums", action="store_true",
                    help="A/B: BatchNorm-backward sums ending in fp32 atomics (round 3's form: not reproducible run to run) instead of the fixed-order form")
    ap.add_argument("--no-events", action="store_true", help="skip the per-launch HIP events (pure timing run)")
    ap.add_argument("--event-steps", type=int, default=1,
                    help="timed steps (spread evenly) in which every forward-conv launch is bracketed by HIP events; each event pair "
                         "costs ~6 us of serialisation, so bracketing all 75 launches in all steps would take ~2.5 %% off `value`")
    ap.add_argument("--launcher-selftest", choices=["ok", "fail"], default=None,
                    help="no GPU work: only exercise the rank launcher and the process group (CPU test of `python bench.py --gpus N`)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: this process becomes the launcher (yolo/main.py:38-42 does the same with mp.spawn) - it has not
        # touched the GPU, starts one fresh rank per GPU as CHILD processes and returns their exit code; nothing is re-exec'ed
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (or run `python bench.py --gpus N`, which does)")
    if args.launcher_selftest:
        raise SystemExit(launcher_selftest(rank, world, args.launcher_selftest))
    # MI355DET_BENCH_ONE_GPU=1: rehearsal of the multi-process path on a one-GPU box (every rank on cuda:0, gloo instead of RCCL,
    # which refuses two ranks on one device); never used for reported numbers
    rehearsal = os.environ.get("MI355DET_BENCH_ONE_GPU", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # the step's dependency chain (forward, loss, BN backward, dgrad) runs on a HIGH-priority stream; the engine's side stream
    # (weight gradients, gradient all-reduce) keeps the default priority and fills what the chain leaves free: the HBM-bound BN
    # backward of a layer then overlaps the previous layer's weight-gradient GEMM instead of waiting for it (+1 % measured, same box)
    # Default: single GPU only - with RCCL in the picture the all-reduce kernels would rank below the chain too, which could not be
    # measured on the one-GPU boxes of this pool; MI355DET_STEP_PRIORITY=1 / 0 forces it on / off for an A/B run on a multi-GPU node
    # (MI355DET_GRADSYNC=rs_ag and MI355DET_BUCKET_MB=<MiB> select the collective and the bucket size the same way, parallel.py).
    prio = os.environ.get("MI355DET_STEP_PRIORITY")
    if (world == 1 and prio != "0") or prio == "1":
        from object_detectors_amd.parallel import step_stream
        torch.cuda.set_stream(step_stream(dev))
    from object_detectors_amd.optim import FlatSGD
    from object_detectors_amd.parallel import GradSync
    from object_detectors_amd.yolo.nets.engine import YoloV3Engine
    from object_detectors_amd.yolo.nets.yolo_forw import YOLOForw

    rec_path = args.tune_record
    if rec_path == "auto":
        rec_path = os.path.join(ROOT, "object_detectors_amd", "tune_records", f"yolov3_d53_bs{args.batch}_{args.px}_{args.storage}.json")
        if not os.path.exists(rec_path) or os.environ.get("MI355DET_TUNE_LOAD"):
            rec_path = "none"
    if rec_path != "none":
        from object_detectors_amd import tune
        tune.load(rec_path, replace=False, lock=True)
    eng = YoloV3Engine("darknet_53", 3, 80, device=dev, seed=0, storage=args.storage, deterministic=not args.atomic_bn_sums)
    loss_scale = 1024.0 if eng.storage == "fp16" else 1.0      # apex-style static scale for the timing run (train_one_epoch.py:88-94)
    crit = YOLOForw(anchors=ANCHORS, num_classes=80, img_size=args.px).to(dev)
    imgs, targets = synth_batch(args.batch, args.px, rank, dev)
    sync = GradSync(eng.flat_g).attach(eng) if world > 1 else None      # every plan the engine builds gets the bucket hooks
    # the reference's optimizer (yolo/hydra/optimizer/sgd.yaml: momentum 0.9, weight decay 5e-4) as one fused kernel over
    # the flat buffers; the step is inside the timed region (train_one_epoch.py:96)
    opt = FlatSGD.for_engine(eng, lr=args.lr, momentum=0.9, weight_decay=5e-4)

    def step():
        out12 = eng.train_step(imgs, targets, crit, grad_scale=loss_scale)
        if sync is not None:
            sync.wait()
        opt.step(grad_scale=1.0 / loss_scale)
        return out12

    for _ in range(max(1, args.warmup)):
        out12 = step()
    torch.cuda.synchronize()
    loss0 = float(out12[0])
    plan = eng._last_plan

    # live per-launch events around the dominant kernel (forward conv) — on torch's current stream, which is the
    # stream the plan launches on
    conv_calls = conv_fwd_flops(plan)
    events = []
    if not args.no_events:
        orig_run = plan._run
        idx = {i: fl for i, fl, _n in conv_calls}

        ev_every = max(1, args.steps // max(1, min(args.event_steps, args.steps)))
        state = {"step": 0}

        def run_with_events(calls):
            if calls is not plan.fwd:
                return orig_run(calls)
            k = state["step"]
            state["step"] += 1
            if k % ev_every != 0:
                return orig_run(calls)
            from object_detectors_amd._lib import check
            from object_detectors_amd.yolo.nets.engine import comm_hook
            for i, (fn, a) in enumerate(calls):
                if fn is comm_hook:
                    a[0](*a[1:])
                    continue
                if i in idx:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    st = fn(*a)
                    e1.record()
                    events.append((e0, e1, idx[i]))
                else:
                    st = fn(*a)
                if st != 0:
                    check(st, fn.__name__)
        plan._run = run_with_events

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out12 = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss1 = float(out12[0])
    in_sync = None
    if world > 1:
        # data-parallel invariant: identical initial weights + averaged gradients => identical weights on every rank
        chk = eng.flat_w.double().abs().sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool(float(hi - lo) <= 1e-9 * float(hi))

    if rank == 0:
        ms = 1000.0 * elapsed / args.steps
        value = args.batch * world * args.steps / elapsed
        roof = None
        if events:
            tot_ms = sum(e0.elapsed_time(e1) for e0, e1, _f in events)
            tot_fl = sum(f for _a, _b, f in events)
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            traffic, tsrc, mfma_util = pmc_traffic(args.batch, args.px)
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM bytes/launch", "traffic_source": tsrc,
                    "mfma_util_counter_pct": mfma_util,
                    "kernel": "conv forward, all 75 convolutions of a step: 73 implicit-GEMM launches (igemm8_kernel / igemm_dx_kernel / igemm_kernel as autotuned per shape) + the stem's statistics pass (stem_kernel<0>) and the fused stem + layer1.ds_conv launch (stem_l1_kernel)",
                    "launches": len(events), "avg_launch_us": round(1000.0 * tot_ms / len(events), 2),
                    "gflop_per_launch": round(tot_fl / len(events) / 1e9, 3)}
        line = {
            "metric": f"images/sec (fwd+bwd) YOLOv3 {args.px}px bs={args.batch}", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": eng.storage, "data": "synthetic",
            "tune_record": (os.path.relpath(rec_path, ROOT) if rec_path != "none" else None), "config": {"workload": f"YOLOv3 Darknet-53 training step (fwd+loss+bwd{'+grad all-reduce' if world > 1 else ''}+SGD step), "
                                   f"synthetic COCO {args.px}px, per-GPU bs={args.batch}, 7 GT/img, random-init weights",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "loss_first": round(loss0, 4), "loss_last": round(loss1, 4), "weights_in_sync_across_ranks": in_sync,
            "roofline": roof,
            "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline(args.px),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
