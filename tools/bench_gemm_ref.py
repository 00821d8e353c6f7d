"""Vendor GEMM (hipBLASLt through torch.mm, bf16) on the GEMM shapes of the large YOLOv3 convolutions: the practical MFMA ceiling
of this box for the same M x N x K, to read the implicit-GEMM numbers against.   python tools/bench_gemm_ref.py"""
import torch
dev = torch.device('cuda:0')
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (m, n, k) in [(204800, 256, 1152), (51200, 512, 2304), (12800, 1024, 4608), (204800, 128, 256), (8192, 8192, 8192), (16384, 4096, 4096)]:
    a = torch.randn(m, k, device=dev).bfloat16()
    b = torch.randn(n, k, device=dev).bfloat16()
    us = timeit(lambda: torch.mm(a, b.t()))
    print(f"M={m:7d} N={n:5d} K={k:5d}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.0f} TFLOP/s", flush=True)
