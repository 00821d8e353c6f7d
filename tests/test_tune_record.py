"""The tune record (include/mi355det.h "tune record", object_detectors_amd/tune.py) on the CPU: export / import / lock through the C ABI,
the JSON file form, the environment policy of `plan_build`, and the rank-0 broadcast over gloo with world size 2.  No kernel runs here:
the record is host state of the library (what the GPU tests check is that a LOCKED record is honoured by the plan build:
tests/test_gpu_trajectory.py, tests/test_gpu_conv.py)."""
import os
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REC_A = [("igemm", 123456789012345, 40), ("igemm", 77, 3), ("s2cat", 5, 1), ("wgrad", 99, 24), ("wgrad", 1 << 62, 2)]
REC_B = [("igemm", 123456789012345, 1), ("wgrad", 99, 8), ("wgrad", 4242, 16)]


def test_export_import_roundtrip_and_file_form(tmp_path):
    from object_detectors_amd import tune
    tune.clear()
    assert tune.export_bytes() == b""
    raw = tune.from_entries(REC_A)
    tune.import_bytes(raw, replace=True, lock=False)
    out = tune.export_bytes()
    assert out == raw and tune.to_entries(out) == sorted(REC_A, key=lambda e: (tune.TABLES.index(e[0]), e[1]))
    # add (not replace): REC_B overrides the keys it shares and adds its own
    tune.import_bytes(tune.from_entries(REC_B), replace=False, lock=False)
    merged = dict(((t, k), v) for t, k, v in REC_A)
    merged.update(((t, k), v) for t, k, v in REC_B)
    assert dict(((t, k), v) for t, k, v in tune.to_entries(tune.export_bytes())) == merged
    # file form: equal records are equal files; load(replace) restores exactly
    pa, pb = str(tmp_path / "a.json"), str(tmp_path / "b.json")
    tune.save(pa)
    tune.save(pb, tune.export_bytes())
    assert open(pa).read() == open(pb).read()
    tune.clear()
    tune.load(pa, replace=True, lock=True)
    assert dict(((t, k), v) for t, k, v in tune.to_entries(tune.export_bytes())) == merged
    assert tune.lock(False) is True and tune.lock(False) is False      # load() locked; lock() returns the previous state
    tune.clear()


def test_import_rejects_malformed_records():
    from object_detectors_amd import tune
    tune.clear()
    with pytest.raises(ValueError):
        tune.import_bytes(b"\x00" * 17)                                 # not a whole number of entries
    import struct
    with pytest.raises(ValueError):
        tune.import_bytes(struct.pack("<IiQ", 9, 1, 1))                 # unknown table
    with pytest.raises(ValueError):
        tune.loads('{"format": "something-else", "entries": []}')
    assert tune.export_bytes() == b""                                    # nothing was imported by the failed calls


def test_plan_build_environment_policy(tmp_path, monkeypatch):
    """MI355DET_TUNE_LOAD is imported (locked) before the first build; MI355DET_TUNE_SAVE is written after every build."""
    from object_detectors_amd import tune
    tune.clear()
    src, dst = str(tmp_path / "in.json"), str(tmp_path / "out.json")
    tune.save(src, tune.from_entries(REC_A))
    monkeypatch.setenv("MI355DET_TUNE_LOAD", src)
    monkeypatch.setenv("MI355DET_TUNE_SAVE", dst)
    monkeypatch.setattr(tune, "_env_loaded", False)
    seen = {}

    def build():
        seen["at_build"] = tune.to_entries(tune.export_bytes())          # the record is already there when the plan is built
        tune.import_bytes(tune.from_entries([("wgrad", 31337, 12)]), lock=False)      # "the build tuned one more shape"
        return "plan"
    assert tune.plan_build(build) == "plan"
    assert len(seen["at_build"]) == len(REC_A)
    assert tune.lock(True) is True                                        # the loaded record was locked
    saved = tune.to_entries(tune.loads(open(dst).read()))
    assert ("wgrad", 31337, 12) in saved and len(saved) == len(REC_A) + 1
    tune.clear()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("MI355DET_TUNE_LOAD", None)
    os.environ.pop("MI355DET_TUNE_SAVE", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from object_detectors_amd import tune
    tune.clear()

    def build():      # every rank "times" and lands on its own choices (rank 1's differ from rank 0's and it has one extra shape)
        tune.import_bytes(tune.from_entries(REC_A if rank == 0 else REC_B), lock=False)
        return rank
    assert tune.plan_build(build) == rank
    got = dict(((t, k), v) for t, k, v in tune.to_entries(tune.export_bytes()))
    locked = tune.lock(True)
    # eval-style build on ONE rank only (share=False): no collective, must not hang
    if rank == 1:
        tune.plan_build(lambda: None, share=False)
    q.put((rank, got, locked))
    dist.barrier()
    dist.destroy_process_group()


def test_rank0_record_is_broadcast_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want0 = dict(((t, k), v) for t, k, v in REC_A)
    assert res[0][1] == want0 and res[0][2] is True
    # rank 1: every shape rank 0 chose is rank 0's choice; its own extra shape stays
    want1 = dict(((t, k), v) for t, k, v in REC_B)
    want1.update(want0)
    assert res[1][1] == want1 and res[1][2] is True


def test_refinement_candidates():
    """tune.refine_step's neighbourhood of a choice: every other wide tile configuration, plain <-> shared-pixel-tile for the narrow outputs, the
    other stride-2 form, split counts around the current one (the library falls back to the nearest valid split)."""
    from object_detectors_amd import tune
    assert set(tune._alternatives("igemm", 40)) == set(tune.IGEMM_CANDIDATES) - {40} and 44 in tune._alternatives("igemm", 1)
    assert tune._alternatives("igemm", 0) == (29, 30, 31) and tune._alternatives("igemm", 30) == (0,)
    assert tune._alternatives("s2cat", 1) == (0,) and tune._alternatives("s2cat", 0) == (1,)
    alt = tune._alternatives("wgrad", 48)
    assert 24 in alt and 96 in alt and 48 not in alt and all(a >= 1 for a in alt)
    assert tune._alternatives("wgrad", 1) == (2, tune.WGRAD_FORM8 + 1)
    alt8 = tune._alternatives("wgrad", tune.WGRAD_FORM8 + 14)      # the 256 x 256 weight-gradient kernel: its own split counts, and the other kernel
    assert tune.WGRAD_FORM8 + 7 in alt8 and tune.WGRAD_FORM8 + 28 in alt8 and 14 in alt8 and tune.WGRAD_FORM8 + 14 not in alt8


def test_refine_step_on_a_cost_model(tmp_path):
    """The search of tune.refine_step on a synthetic step whose time is a function of the record: a change worth more than `min_gain_us` is found
    and kept, one worth less is not, entries of the other storage format are never touched, the refined record is left imported and locked, and
    the checkpoint file holds it."""
    import random
    from object_detectors_amd import tune
    tune.clear()
    K_W, K_I, K_S, K_F16 = 1000 * 2, 2000 * 2, 7, 3000 * 2 + 1            # keys end in the format bit (bf16 = 0); the s2cat key carries it in bit 61
    start = [("wgrad", K_W, 64), ("igemm", K_I, 40), ("s2cat", K_S, 1), ("wgrad", K_F16, 64)]
    tune.import_bytes(tune.from_entries(start), replace=True, lock=True)
    rng = random.Random(0)
    seen_f16 = set()

    def cost():      # us per step: wgrad split 32 is 300 us better than 64, 16 is worse again; igemm 44 is 20 us better than 40 (below the bar)
        rec = {(t, k): v for t, k, v in tune.to_entries(tune.export_bytes())}
        seen_f16.add(rec[("wgrad", K_F16)])
        c = 28000.0 + {64: 300.0, 32: 0.0}.get(rec[("wgrad", K_W)], 500.0) + (0.0 if rec[("igemm", K_I)] == 44 else 20.0)
        c += 200.0 if rec[("s2cat", K_S)] == 0 else 0.0
        return c + rng.uniform(0.0, 10.0)

    ck = str(tmp_path / "ck.json")
    lines = []
    s_us, f_us, kept = tune.refine_step(lambda: None, rounds=2, steps=1, min_gain_us=40.0, budget_s=60.0, log=lines.append, checkpoint=ck,
                                        timer=lambda step, steps: cost())
    got = {(t, k): v for t, k, v in tune.to_entries(tune.export_bytes())}
    assert got == {("wgrad", K_W): 32, ("igemm", K_I): 40, ("s2cat", K_S): 1, ("wgrad", K_F16): 64}
    assert kept == 1 and s_us - f_us > 250.0 and seen_f16 == {64}
    assert tune.lock(False) is True
    assert {(t, k): v for t, k, v in tune.to_entries(tune.loads(open(ck).read()))} == got
    assert any("64 -> 32" in l for l in lines)
    assert abs(tune.refine_step.last_drift_us) < 20.0 and not any("WARNING" in l for l in lines)
    # a step that gets faster by itself while the sweep runs (round 4: a model diverging on the repeated batch) is reported
    tune.import_bytes(tune.from_entries(start), replace=True, lock=True)
    calls = [0]

    def drifting(step, steps):
        calls[0] += 1
        return cost() - 4.0 * calls[0]
    lines2 = []
    tune.refine_step(lambda: None, rounds=1, steps=1, min_gain_us=40.0, budget_s=60.0, log=lines2.append, timer=drifting)
    assert tune.refine_step.last_drift_us < -120.0 and any("WARNING" in l and "drifted" in l for l in lines2)
    tune.clear()


def test_weight_gradient_key_fields_and_split_rule():
    """tune.wgrad_key_fields inverts the library's weight-gradient key (csrc/wgrad_kernels.hip:wgrad_key: pixels, cout, cin, ksize * 4 + stride, format
    bit), and tune.wgrad_split_valid is the library's rule for a split count (chunks of whole 64-pixel k-steps, none empty) - the two helpers the
    A/B record tools (tools/tune_ab.py --wgrad8, tools/tune_drop.py) stand on."""
    from object_detectors_amd import tune

    def key(m, cout, cin, ks, stride, f16):      # the arithmetic of wgrad_key, restated
        k = m
        k = k * 4099 + cout
        k = k * 4099 + cin
        k = k * 17 + ks * 4 + stride
        return k * 2 + int(f16)
    for f in [(204800, 256, 128, 3, 1, False), (51200, 512, 256, 3, 2, True), (80000, 819, 256, 3, 1, False), (12800, 255, 1024, 1, 1, False), (392, 36, 256, 3, 1, True)]:
        assert tune.wgrad_key_fields(key(*f)) == f
    # channel counts from 4099 up (the 10 836-channel cls_logits) carry into the pixel field: the key stays a fine hash, the decoding is not defined
    assert tune.wgrad_key_fields(key(80000, 10836, 256, 3, 1, False))[0] != 80000
    # the committed bench record decodes into the YOLOv3 @640 bs-32 weight-gradient shapes
    rec = os.path.join(ROOT, "object_detectors_amd", "tune_records", "yolov3_d53_bs32_640_bf16.json")
    fields = [tune.wgrad_key_fields(k) for t, k, v in tune.to_entries(tune.loads(open(rec).read())) if t == "wgrad"]
    assert (51200, 512, 256, 3, 1, False) in fields and (204800, 256, 128, 3, 1, False) in fields and all(not f[5] for f in fields)
    for m in (64, 507, 3200, 12800, 80000):
        for sp in range(1, 40):
            chunks = [min(m, (i + 1) * (((m + sp - 1) // sp + 63) // 64 * 64)) - i * (((m + sp - 1) // sp + 63) // 64 * 64) for i in range(sp)]
            assert tune.wgrad_split_valid(m, sp) == all(c > 0 for c in chunks), (m, sp)
    assert not tune.wgrad_split_valid(64, 0)
