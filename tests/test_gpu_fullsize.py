"""Size-independent properties of the hot path at BASELINE.json's full size (YOLOv3, batch 32, 640 px): the oracle is too slow
there, so the kernels are checked through identities that hold for any input (linearity of the convolution, the per-channel sums
the BN-statistics epilogue must reproduce, the two orthogonality relations of the BatchNorm backward, bit-identity of the two
weight-gradient staging forms).  Layer shapes: darknet.py:41-107 at 640 px."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def randn(shape, seed, scale=1.0):
    g = torch.Generator(device=dev()).manual_seed(seed)
    return (torch.randn(shape, device=dev(), generator=g) * scale).bfloat16()


@pytest.mark.parametrize("cin,cout,k,s,hw", [(128, 256, 3, 1, 80), (256, 128, 1, 1, 80), (32, 64, 3, 2, 640)])
def test_conv_forward_linearity_and_statistics_rows(cin, cout, k, s, hw):
    from object_detectors_amd import ops
    from object_detectors_amd._lib import lib
    n = 32
    shape = ops.conv_shape(n, hw, hw, cin, cout, k, s)
    wt = torch.randn(cout, cin, k, k, device=dev(), generator=torch.Generator(device=dev()).manual_seed(1)) * (2.0 / (cin * k * k)) ** 0.5
    wf, _ = ops.pack_weights(shape, wt, want_dgrad=False)
    # x2 = 2 * x1 exactly in bf16, so x1 + x2 = 3 * x1 needs rounding only once more: compare y(3 x1) with 3 y(x1) in fp32 accumulators'
    # terms through the bf16 outputs (three roundings: tolerance 3 ulp of the largest term)
    x1 = randn((n, hw, hw, cin), 2)
    x3 = (x1.float() * 3).bfloat16()
    exact = (x3.float() == x1.float() * 3)
    x1 = torch.where(exact, x1, torch.zeros_like(x1))         # keep only values whose triple is a bf16 number
    x3 = (x1.float() * 3).bfloat16()
    rows = ops.conv_stats_rows(shape)
    cp = ops.cout_pad_of(cout)
    ys = []
    lib().mi355det_conv_autotune_mode(1)                      # the configuration the engine would pick at this size
    try:
        y = torch.empty(n, shape.ho, shape.wo, cout, dtype=torch.bfloat16, device=dev())
        ops.conv_fwd(shape, x1, wf, y, stats=torch.zeros(rows + 64, 2, cp, device=dev()))
    finally:
        lib().mi355det_conv_autotune_mode(0)
    for x in (x1, x3):
        y = torch.empty(n, shape.ho, shape.wo, cout, dtype=torch.bfloat16, device=dev())
        stats = torch.zeros(rows + 64, 2, cp, device=dev())
        ops.conv_fwd(shape, x, wf, y, stats=stats)
        ys.append((y, stats))
    y1, y3 = ys[0][0].float(), ys[1][0].float()
    scale = y3.abs().max().item()
    assert scale > 0
    assert (y3 - 3 * y1).abs().max().item() <= 3 * 2 ** -8 * scale
    # the BN-statistics epilogue: per-channel sums of the STORED tensor, from whatever tile configuration was tuned
    for y, stats in ys:
        yf = y.float().view(-1, cout)
        s1 = stats[:rows, 0, :cout].double().sum(0)
        s2 = stats[:rows, 1, :cout].double().sum(0)
        ref1, ref2 = yf.double().sum(0), (yf.double() ** 2).sum(0)
        assert torch.allclose(s1, ref1, rtol=1e-4, atol=1e-3 * yf.abs().double().sum(0).max().item())
        assert torch.allclose(s2, ref2, rtol=1e-4)


@pytest.mark.parametrize("c,hw", [(64, 320), (256, 80), (1024, 20)])
def test_bn_backward_orthogonality(c, hw):
    """dz of BatchNorm backward is orthogonal to 1 and to xhat per channel (the two terms the apply pass subtracts)."""
    from object_detectors_amd._lib import check, lib, ptr, stream_ptr
    pixels = 32 * hw * hw
    z = randn((pixels, c), 3, 1.5)
    g = randn((pixels, c), 4)
    zf = z.float()
    mean, var = zf.mean(0), zf.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    gamma = torch.rand(c, device=dev()) + 0.5
    beta = torch.randn(c, device=dev()) * 0.1
    ss = torch.cat([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).contiguous()
    sums = torch.zeros(2 * c, device=dev())
    dz = torch.empty_like(z)
    dg, db = torch.zeros(c, device=dev()), torch.zeros(c, device=dev())
    L = lib()
    check(L.mi355det_bn_act_bwd_reduce(ptr(g), c, None, 0, ptr(z), c, ptr(ss), c, pixels, 0.1, ptr(sums), stream_ptr()))
    check(L.mi355det_bn_act_bwd_apply(ptr(g), c, None, 0, ptr(z), c, ptr(ss), ptr(sums), None, c, pixels, 0.1, ptr(dz), c, ptr(dg), ptr(db), stream_ptr()))
    d = dz.float()
    xhat = (zf - mean) * invstd
    mag = d.abs().sum(0)
    assert (d.sum(0).abs() <= 2e-3 * mag + 1e-3).all()               # bf16 rounding of ~1e6 terms per channel
    assert ((d * xhat).sum(0).abs() <= 2e-3 * (d * xhat).abs().sum(0) + 1e-3).all()
    # the reduce pass against a float64 reduction of the same definition
    y = zf * ss[:c] + ss[c:2 * c]
    dy = torch.where(y > 0, g.float(), g.float() * 0.1)
    assert torch.allclose(sums[:c].double(), dy.double().sum(0), rtol=1e-3, atol=1e-3 * dy.abs().sum(0).max().item())
    assert torch.allclose(sums[c:].double(), (dy * xhat).double().sum(0), rtol=1e-3, atol=1e-3 * (dy * xhat).abs().sum(0).max().item())


def test_weight_gradient_forms_bit_identical_full_size():
    from object_detectors_amd import ops
    from object_detectors_amd._lib import lib
    n, hw, cin, cout = 32, 80, 128, 256
    shape = ops.conv_shape(n, hw, hw, cin, cout, 3, 1)
    x = randn((n, hw, hw, cin), 5)
    gy = randn((n, hw, hw, cout), 6)
    outs = []
    try:
        for general in (1, 0):
            lib().mi355det_debug_set(1, general)
            dw = torch.zeros(cout, 9 * cin, device=dev())
            ops.conv_wgrad(shape, x, gy, dw)
            outs.append(dw)
    finally:
        lib().mi355det_debug_set(1, 0)
    assert torch.equal(outs[0], outs[1])
    # a column of the result against a float64 contraction (centre tap, 8 output channels)
    ref = torch.einsum("pc,pk->ck", gy.view(-1, cout)[:, :8].double(), x.view(-1, cin).double())
    got = outs[1].view(cout, 9, cin)[:8, 4, :].double()
    assert torch.allclose(got, ref, rtol=2e-3, atol=2e-3 * ref.abs().max().item())


@pytest.mark.parametrize("storage", ["bf16", "fp16"])
def test_committed_bench_tune_record_covers_the_bench_plan(storage):
    """object_detectors_amd/tune_records/yolov3_d53_bs32_640_<storage>.json (the step-refined record bench.py loads by default) must cover every
    shape the bench's plan tunes: after loading it locked, building the batch-32 / 640-px training plan adds no entry and changes none.  A
    record that went stale (a changed shape key, a new tunable) would not break anything - the missing shapes are timed on the box - but the
    bench would silently lose its refined, box-independent choices."""
    import os
    from object_detectors_amd import tune
    from object_detectors_amd.yolo.nets.engine import YoloV3Engine
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "object_detectors_amd", "tune_records",
                        f"yolov3_d53_bs32_640_{storage}.json")
    assert os.path.exists(path)
    tune.clear()
    try:
        rec = tune.load(path, replace=True, lock=True)
        eng = YoloV3Engine("darknet_53", 3, 80, device=dev(), seed=0, storage=storage)
        eng.plan(32, 640, 640, True)
        torch.cuda.synchronize()
        after = tune.export_bytes()
        assert after == rec, sorted(set(tune.to_entries(after)) - set(tune.to_entries(rec)))[:8]
    finally:
        tune.clear()
