"""The tune record: the choices the plan build makes by timing, as data (include/mi355det.h, "tune record").

Both engines choose per convolution shape, at plan-build time and by TIMING the candidates on the plan's own buffers, a tile configuration
(forward / data gradient), the form of the stride-2 data gradient and the split count of the weight gradient.  Every choice fixes a summation
order, so the numbers a step produces depend on what was fastest on that box at that moment (the trajectory test's worst
step moved between 5.8 and 13.7 % across boxes of the pool with identical code).  The reference has the same property only per process
(`torch.backends.cudnn.benchmark`, yolo/main.py / detection/train.py) and no way to pin it.  Here the choices can be

  * saved   MI355DET_TUNE_SAVE=<file>  or  tune.save(path)      (JSON, one line per choice, sorted: equal records are equal files)
  * loaded  MI355DET_TUNE_LOAD=<file>  or  tune.load(path)      (locked: a shape that has an entry is never timed again)
  * shared  tune.sync_from_rank0(group)                         rank 0's record is broadcast and overrides the other ranks' choices: the
                                                                same kernels, the same speed and the same summation order on every rank

`plan_build(engine_build)` is what the engines wrap around their plan construction; it implements the three behaviours above.
"""
import ctypes as C
import json
import os

from ._lib import check, lib

ENTRY = 16          # bytes per entry: u32 table, i32 value, u64 key (mi355det_tune_entry)
TABLES = ("igemm", "s2cat", "wgrad")
FORMAT = "mi355det-tune-1"
_env_loaded = False


def export_bytes():
    L = lib()
    n = L.mi355det_tune_export(None, 0)
    buf = (C.c_char * max(n, 1))()
    got = L.mi355det_tune_export(C.cast(buf, C.c_void_p), n)
    if got != n:
        raise RuntimeError("tune record changed size during export")
    return bytes(buf[:n])


def import_bytes(raw, replace=False, lock=True):
    raw = bytes(raw)
    buf = C.create_string_buffer(raw, len(raw)) if raw else None
    check(lib().mi355det_tune_import(C.cast(buf, C.c_void_p) if raw else None, len(raw), 1 if replace else 0), "tune_import")
    if lock:
        lib().mi355det_tune_lock(1)


def lock(on=True):
    return bool(lib().mi355det_tune_lock(1 if on else 0))


def clear():
    check(lib().mi355det_tune_clear(), "tune_clear")


def to_entries(raw):
    """bytes -> sorted list of (table name, key, value)."""
    import struct
    out = []
    for i in range(0, len(raw), ENTRY):
        t, v, k = struct.unpack_from("<IiQ", raw, i)
        out.append((TABLES[t], k, v))
    return out


def from_entries(entries):
    import struct
    ents = sorted((TABLES.index(t), int(k), int(v)) for t, k, v in entries)
    return b"".join(struct.pack("<IiQ", t, v, k) for t, k, v in ents)


def dumps(raw=None):
    raw = export_bytes() if raw is None else raw
    rows = to_entries(raw)
    body = ",\n".join('  ["%s", %d, %d]' % r for r in rows)
    return '{"format": "%s",\n "entries": [\n%s\n]}\n' % (FORMAT, body)


def loads(text):
    d = json.loads(text)
    if d.get("format") != FORMAT:
        raise ValueError("not a %s file" % FORMAT)
    return from_entries(d["entries"])


def save(path, raw=None):
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        f.write(dumps(raw))
    os.replace(tmp, path)


def load(path, replace=False, lock=True):
    with open(path) as f:
        raw = loads(f.read())
    import_bytes(raw, replace=replace, lock=lock)
    return raw


def sync_from_rank0(group=None, src=0):
    """Broadcast rank `src`'s record; every other rank imports it (overriding its own choices) and every rank locks.  Collective: every rank of the group calls it at the same point."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return None
    rank = dist.get_rank(group)
    gsrc = dist.get_global_rank(group, src) if group is not None else src
    payload = [export_bytes() if rank == src else None]
    dist.broadcast_object_list(payload, src=gsrc, group=group)
    if rank != src:
        import_bytes(payload[0], replace=False, lock=False)
    lock(True)            # on every rank, `src` included: a shape of the shared record is never timed again by a later plan build
    return payload[0]


def _dist_world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def plan_build(build, group=None, share=None):
    """Run `build()` (an engine's plan construction, which autotunes) under the record policy:
       - MI355DET_TUNE_LOAD: the file is imported (locked) once per process before the first build: shapes it covers are not timed;
       - N > 1 (a process group is up; `share=False` or MI355DET_TUNE_SHARE=0 switches it off): every rank builds (the tuning passes may
         contain the SyncBN collectives, so no rank can wait for another here), then rank 0's record is broadcast and overrides the others':
         the choices live in the library keyed by shape and are looked up at launch time, so from the first real step on every rank runs the
         same kernels in the same summation order at the same speed.  Collective: every rank must build the same plans in the same order
         (training plans do: the reference broadcasts its multi-scale size from rank 0, train_one_epoch.py:15-26);
       - MI355DET_TUNE_SAVE: the record is written after the build (rank 0 only)."""
    global _env_loaded
    if not _env_loaded:
        _env_loaded = True
        path = os.environ.get("MI355DET_TUNE_LOAD")
        if path:
            load(path, replace=False, lock=True)
    out = build()
    rank, world = _dist_world(group)
    if share is None:
        share = os.environ.get("MI355DET_TUNE_SHARE", "1") != "0"
    if world > 1 and share:
        sync_from_rank0(group)
    path = os.environ.get("MI355DET_TUNE_SAVE")
    if path and rank == 0:
        save(path)
    return out


# ---- step-level refinement ------------------------------------------------------------------------------------------------------------
IGEMM_CANDIDATES = (1, 2, 3, 4, 5, 6, 15, 16, 17, 18, 19, 26, 27, 28, 40, 44, 45)      # conv_kernels.hip:autotune_igemm cands[]
WGRAD_FORM8 = 1 << 16
_NARROW = {0: (29, 30, 31), 29: (0,), 30: (0,), 31: (0,)}


def wgrad_key_fields(key):
    """(pixels n*ho*wo, cout, cin, ksize, stride, fp16 storage) of a weight-gradient key (csrc/wgrad_kernels.hip: wgrad_key).  Defined for channel
    counts below 4099 (the key's radix): larger ones carry into the next field - the key is still a usable hash, but not decodable."""
    f16, key = key % 2, key // 2
    ks, key = key % 17, key // 17
    cin, key = key % 4099, key // 4099
    cout, m = key % 4099, key // 4099
    return m, cout, cin, ks // 4, ks % 4, bool(f16)


def wgrad_split_valid(m, sp):
    """csrc/wgrad_kernels.hip: split_valid - the pixel axis is cut into `sp` chunks of whole 64-pixel k-steps, none of them empty."""
    if sp < 1:
        return False
    chunk = ((m + sp - 1) // sp + 63) // 64 * 64
    return (m + chunk - 1) // chunk == sp


def _alternatives(table, v):
    if table == "igemm":
        return _NARROW[v] if v in _NARROW else tuple(c for c in IGEMM_CANDIDATES if c != v)
    if table == "s2cat":
        return (1 - v,)
    # weight gradient: split count, + WGRAD_FORM8 = the 256 x 256 phase-staggered kernel (csrc/wgrad_kernels.hip: WG_FORM8; the library falls
    # back to the 128 x 128 kernel for shapes the other one does not take, and to the nearest valid split count)
    form, sp = v & WGRAD_FORM8, v & (WGRAD_FORM8 - 1)
    out = []
    for f in (0.5, 0.67, 0.8, 1.25, 1.5, 2.0):
        w = max(1, int(round(sp * f)))
        if w != sp and (w | form) not in out:
            out.append(w | form)
    if form:
        out.append(sp)
    else:      # (a wasted trial where the library falls back: same kernel, same time)
        for w in (sp | WGRAD_FORM8, max(1, sp // 2) | WGRAD_FORM8):
            if w not in out:
                out.append(w)
    return tuple(out)


def refine_step(step, storage="bf16", rounds=1, steps=6, min_gain_us=40.0, budget_s=900.0, skip=0, log=print, checkpoint=None, timer=None):
    """Coordinate descent on the WHOLE STEP over the current tune record (tools/tune_step.py is the command line for the YOLO bench).

    The plan build times every candidate alone - back to back with itself, warm caches, the whole chip.  Inside a training step a data gradient
    shares the chip with a weight gradient on the other stream and every forward convolution starts behind a streaming BatchNorm pass, so the
    isolated winner is not always the step's winner (round 4: halving one weight-gradient split count and changing two tile configurations took
    0.44 ms off a 29.1 ms step).  `step()` runs one complete training step of an engine whose plans are already built; for one entry at a time
    (a shape: every layer of that shape moves together) the other legal values are tried, and a value is kept only if the step got faster by
    more than `min_gain_us`, confirmed against a fresh measurement of the incumbent.  Choices are looked up in the library at launch time, so a
    trial is an import of a modified record + `steps` steps: no plan rebuild.  Leaves the refined record imported and locked; returns
    (start_us, final_us, number of entries changed); `refine_step.last_drift_us` = the start record re-measured at the end minus its first measurement
    (a warning is logged when that exceeds 3 bars).  Entries of the other storage format are left alone.

    `min_gain_us` has to sit above the run-to-run spread of `steps` steps: 40 us is right for the 28 ms YOLO step; the 40 ms RetinaNet-R101-LVIS
    step refined with 40 us collected 53 "improvements" that were 1 % SLOWER than no record on another box (150 us: profiles/r04_ab_results.md §7).
    `step` has to be STATIONARY (learning rate 0): a sweep runs thousands of steps on one batch, a model that diverges meanwhile runs on NaN operands,
    and those steps are faster (RetinaNet-R101-LVIS: 33.3 against 36 ms) - two records of round 4 were refined on such a model and withdrawn.
    `timer(step, steps) -> us per step` replaces the device-synchronised wall clock (tests/test_tune_record.py drives the search on a cost model)."""
    import time

    def wall(step, steps):
        import torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e6

    timer = timer or wall

    def measure(reps):
        return min(timer(step, steps) for _ in range(reps))

    cur = {(t, k): v for t, k, v in to_entries(export_bytes())}
    fmt_bit = 1 if storage == "fp16" else 0

    def mine(t, k):      # the igemm / wgrad keys end in the format bit, the s2cat key carries it in bit 61 (csrc: igemm_key, wgrad_key)
        return ((k >> 61) & 1) == fmt_bit if t == "s2cat" else (k & 1) == fmt_bit

    def apply(d):
        import_bytes(from_entries([(t, k, v) for (t, k), v in d.items()]), replace=True, lock=True)

    apply(cur)
    first = dict(cur)
    base = start = measure(5)
    log(f"start: {base:.0f} us per step")
    t_begin, kept = time.time(), 0
    order = sorted(cur, key=lambda e: (("wgrad", "igemm", "s2cat").index(e[0]), e[1]))
    for rnd in range(rounds):
        changed = False
        for pos, (t, k) in enumerate(order):
            if not mine(t, k) or time.time() - t_begin > budget_s or (rnd == 0 and pos < skip):
                continue
            v0 = cur[(t, k)]
            log(f"  [{time.time() - t_begin:5.0f} s] round {rnd} entry {pos}: {t} {k} (now {v0}; step {base:.0f} us)")
            best_v, best_t = v0, base
            for v in _alternatives(t, v0):
                trial = dict(cur)
                trial[(t, k)] = v
                apply(trial)
                tm = measure(2)
                if tm < best_t - min_gain_us:
                    best_v, best_t = v, tm
            if best_v != v0:
                trial = dict(cur)
                trial[(t, k)] = best_v
                apply(trial)
                conf = measure(4)
                apply(cur)
                inc = measure(4)
                if conf < inc - min_gain_us:
                    cur[(t, k)] = best_v
                    base, kept, changed = conf, kept + 1, True
                    log(f"round {rnd}: {t} {k}: {v0} -> {best_v}  ({inc:.0f} -> {conf:.0f} us per step)")
                    if checkpoint:
                        apply(cur)
                        save(checkpoint)
            apply(cur)
        if not changed:
            break
    # drift check: the START record again, at the end.  A step that got faster or slower by itself during the sweep (a model diverging on the repeated
    # batch, a clock that moved) makes every acceptance above suspect - round 4 lost three records to exactly this.
    apply(first)
    again = measure(5)
    refine_step.last_drift_us = again - start
    if abs(again - start) > 3.0 * min_gain_us:
        log(f"WARNING: the start record measures {again:.0f} us now against {start:.0f} us at the beginning: the step drifted by itself "
            f"({again - start:+.0f} us); do not trust this sweep (is the learning rate 0?)")
    apply(cur)
    final = measure(5)
    log(f"final: {final:.0f} us per step; {kept} entries changed; {time.time() - t_begin:.0f} s of search")
    return start, final, kept
