// Microbenchmark: rate at which a CU can move L2-resident 128-byte rows into LDS, by mechanism.
//   mode 0: LDS-DMA (global_load_lds_dwordx4), mode 1: global_load_dwordx4 -> ds_write_b128, mode 2: global_load only
// Access shape = the convolution's im2col gather: 64 lanes fetch 8 rows x 128 B (row pitch `ld` bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int PER>   // PER = 1 KB pieces per wave per step
__global__ __launch_bounds__(256, 2) void fill_kernel(const char* __restrict__ src, size_t footprint, int ld, int steps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int row = lane >> 3, chunk = lane & 7;
  // each workgroup walks its own stream of tiles through the footprint
  size_t base = ((size_t)blockIdx.x * 7919u * 4096u) % footprint;
  float acc = 0.f;
  const int stage_bytes = 4 * PER * 1024;
  for (int s = 0; s < steps; ++s) {
    char* stage = smem + (s & 1) * stage_bytes;
    uint4 regs[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int piece = wid * PER + i;
      size_t off = (base + (size_t)(piece * 8 + row) * ld + chunk * 16) % footprint;
      off &= ~(size_t)15;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                         (__attribute__((address_space(3))) void*)(stage + piece * 1024), 16, 0, 0);
      } else {
        regs[i] = *(const uint4*)(src + off);
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < PER; ++i) *(uint4*)(stage + (wid * PER + i) * 1024 + lane * 16) = regs[i];
    }
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < PER; ++i) acc += __uint_as_float(regs[i].x ^ regs[i].w);
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");   // keep one step in flight
    base = (base + (size_t)4 * PER * 8 * ld) % footprint;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 2) acc = *(float*)(smem + threadIdx.x * 4);
  if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int PER>
void run(const char* name, const char* src, size_t footprint, int ld, float* sink, int wgs) {
  const int steps = 4000;
  const int lds = 2 * 4 * PER * 1024;
  CHECK(hipFuncSetAttribute((const void*)fill_kernel<MODE, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((fill_kernel<MODE, PER>), dim3(wgs), dim3(256), lds, 0, src, footprint, ld, 100, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((fill_kernel<MODE, PER>), dim3(wgs), dim3(256), lds, 0, src, footprint, ld, steps, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)wgs * steps * 4 * PER * 1024;
  printf("%-34s PER=%d wgs=%4d footprint=%5.1f MB ld=%4d: %7.2f TB/s chip, %6.1f GB/s per CU\n", name, PER, wgs, footprint / 1048576.0, ld,
         bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
  const size_t cap = 512u << 20;
  char* src; float* sink;
  CHECK(hipMalloc(&src, cap)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(src, 1, cap));
  for (size_t fp : {(size_t)8 << 20, (size_t)24 << 20, (size_t)200 << 20}) {
    for (int ld : {128, 256, 1024}) {
      for (int wgs : {256, 512}) {
        run<0, 8>("LDS-DMA global_load_lds x4", src, fp, ld, sink, wgs);
        run<1, 8>("global_load x4 + ds_write_b128", src, fp, ld, sink, wgs);
        run<2, 8>("global_load x4 only", src, fp, ld, sink, wgs);
      }
    }
  }
  run<0, 4>("LDS-DMA", src, (size_t)8 << 20, 256, sink, 1024);
  run<1, 4>("global_load + ds_write", src, (size_t)8 << 20, 256, sink, 1024);
  run<2, 4>("global_load only", src, (size_t)8 << 20, 256, sink, 1024);
  return 0;
}
