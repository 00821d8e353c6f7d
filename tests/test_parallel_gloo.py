"""Multi-process (world_size 2, gloo, CPU) test of the data-parallel gradient path: bucket planning over the
flat gradient buffer and averaged all-reduce woven into a backward call list."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_buckets_cover_buffer_once():
    from object_detectors_amd.parallel import plan_buckets
    # 10 layers of 1000 elements, backward order = decreasing offsets
    marks = [(3 * (i + 1), 9000 - 1000 * i) for i in range(10)]
    b = plan_buckets(marks, 10000, 2500)
    covered = sorted((lo, hi) for _p, lo, hi in b)
    assert covered[0][0] == 0 and covered[-1][1] == 10000
    for (a0, a1), (b0, b1) in zip(covered, covered[1:]):
        assert a1 == b0
    # positions are non-decreasing in backward order and every bucket is complete when it fires
    for pos, lo, hi in b:
        done = [o for p, o in marks if p <= pos]
        assert min(done) <= lo
    # tiny bucket size: one bucket per layer; huge: a single bucket at the end
    assert len(plan_buckets(marks, 10000, 1)) == 10
    one = plan_buckets(marks, 10000, 10 ** 9)
    assert one == [(marks[-1][0], 0, 10000)]


def _worker(rank, world, port, q, algo="all_reduce"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from object_detectors_amd.parallel import GradSync
    from object_detectors_amd.yolo.nets.engine import comm_hook

    class FakePlan:
        pass
    n_layers, per = 12, 777
    flat = torch.zeros(n_layers * per)
    plan = FakePlan()
    order = []

    def make_layer(i):
        def f():
            flat[i * per:(i + 1) * per] = float(rank + 1) * (i + 1)      # "wgrad" of layer i
            order.append(i)
            return 0
        f.__name__ = f"layer{i}"
        return f
    plan.bwd = [(make_layer(i), ()) for i in reversed(range(n_layers))]
    plan.bwd_marks = [(k + 1, (n_layers - 1 - k) * per) for k in range(n_layers)]
    sync = GradSync(flat, bucket_mb=per * 4 * 3 / (1 << 20), algo=algo)          # ~3 layers per bucket (3 * 777 elements: odd, so rs_ag has a remainder)
    sync.install(plan)
    hooks = sum(1 for fn, _a in plan.bwd if fn is comm_hook)
    for fn, args in plan.bwd:
        if fn is comm_hook:
            args[0](*args[1:])
        else:
            fn(*args)
    sync.wait()
    expect = torch.cat([torch.full((per,), (1 + 2) / 2.0 * (i + 1)) for i in range(n_layers)])
    ok = bool(torch.allclose(flat, expect)) and hooks >= 3 and order == list(reversed(range(n_layers)))
    q.put((rank, ok, hooks))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["all_reduce", "rs_ag"])
def test_gradsync_world2_gloo(algo):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + (7 if algo == "rs_ag" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, algo)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _r, ok, _h in res), res


def test_gradsync_binds_each_plans_own_side_stream_and_attaches_to_engines():
    """every plan has its own side stream; the bucket hook must launch on the stream of the plan whose backward is running,
    and plans built later (another input size, an LRU rebuild) must get the hooks too."""
    from object_detectors_amd.parallel import GradSync
    from object_detectors_amd.yolo.nets.engine import comm_hook

    class FakePlan:
        training = True

        def __init__(self, stream):
            self.side_stream = stream
            self.bwd = [((lambda: 0), ()) for _ in range(4)]
            self.bwd_marks = [(k + 1, (3 - k) * 100) for k in range(4)]

    class FakeEngine:
        def __init__(self):
            self.plans = {}

    flat = torch.zeros(400)
    sync = GradSync(flat, bucket_mb=200 * 4 / (1 << 20))
    eng = FakeEngine()
    a = FakePlan("stream-A")
    eng.plans["a"] = a
    sync.attach(eng)                                   # existing plans are hooked at attach time
    assert eng.grad_syncs == [sync]
    b = FakePlan("stream-B")
    for gs in eng.grad_syncs:                          # what Engine.plan() does for a newly built plan
        gs.install(b)
    for plan, want in ((a, "stream-A"), (b, "stream-B")):
        hooks = [args for fn, args in plan.bwd if fn is comm_hook]
        assert len(hooks) == 2 and all(h[-1] == want for h in hooks), hooks
        assert [h[1:3] for h in hooks] == [(200, 400), (0, 200)]
    sync.install(a)                                    # idempotent
    assert sum(1 for fn, _ in a.bwd if fn is comm_hook) == 2


def _worker_paramsync(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from object_detectors_amd.parallel import ParamGradSync
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2), requires_grad=False),
              torch.nn.Parameter(torch.zeros(4))]
    params[0].grad = torch.full((5, 3), float(rank + 1))
    params[1].grad = torch.arange(7.0) * (rank + 1)
    params[3].grad = None                                  # a parameter without a gradient on this rank counts as zero
    sync = ParamGradSync(params)
    sync.reduce()
    sync.wait()
    ok = (torch.allclose(params[0].grad, torch.full((5, 3), 1.5)) and torch.allclose(params[1].grad, torch.arange(7.0) * 1.5)
          and params[2].grad is None and torch.equal(params[3].grad, torch.zeros(4)))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_param_grad_sync_world2_gloo():
    """ParamGradSync (the Faster R-CNN box-head parameters outside the engine's flat buffer, detection/train.py:160): one flattened
    all-reduce, average written back into every .grad."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_paramsync, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _r, ok in res), res
