// Shared helpers for libmi355det (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <mutex>

#if defined(MI355_F16) && MI355_F16
#include "f16_names.h"      // entry points and cross-file symbols of the fp16-storage objects get the suffix _f16
#endif
#include "../../include/mi355det.h"

#define WAVE 64

namespace mi355 {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b, c);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return MI355DET_ELAUNCH;
  }
  return MI355DET_OK;
}

typedef unsigned short bf16_t;  // raw bits of a 16-bit storage element (bf16; IEEE fp16 in the *_f16 objects, see below)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// ---- storage format of the convolution engine's activations, gradients and packed weights.
// The reference trains either in fp32 (apex O0, its default) or with fp16 storage (apex O2: yolo/batch_files/sample.txt:28-44,
// yolo/procedures/initialize.py:44-45).  This library stores bf16 by default; the files that touch stored activations (conv_kernels, igemm8_kernels,
// wgrad_kernels, dgrad_s2_kernels, elem_kernels, stem_kernels, stem_l1_kernels) are compiled a SECOND time with -DMI355_F16=1 into *_f16 objects
// whose entry points carry the suffix _f16 (f16_names.h; declared in include/mi355det_f16.h): the same loops on the f16 MFMA forms, fp32
// accumulation and fp32 masters unchanged.  s2f / f2s convert one stored element, st16x8_t / st16x4_t are the MFMA operand vectors.
#ifndef MI355_F16
#define MI355_F16 0
#endif
#if MI355_F16
typedef _Float16 st16_scalar_t;
#define MI355_MFMA_BUILTIN_16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define MI355_MFMA_BUILTIN_32 __builtin_amdgcn_mfma_f32_32x32x16_f16
__device__ __forceinline__ float s2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ bf16_t f2s(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }      // round to nearest even, overflow -> inf, NaN stays NaN
#else
typedef __bf16 st16_scalar_t;
#define MI355_MFMA_BUILTIN_16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MI355_MFMA_BUILTIN_32 __builtin_amdgcn_mfma_f32_32x32x16_bf16
__device__ __forceinline__ float s2f(bf16_t v) { return bf2f(v); }
__device__ __forceinline__ bf16_t f2s(float f) { return f2bf(f); }
#endif
typedef __attribute__((ext_vector_type(8))) st16_scalar_t st16x8_t;
typedef __attribute__((ext_vector_type(4))) st16_scalar_t st16x4_t;
typedef __attribute__((ext_vector_type(4))) float mi355_f32x4_t;
typedef __attribute__((ext_vector_type(16))) float mi355_f32x16_t;
// D = A * B + C on the matrix cores in the storage format (v_mfma_f32_16x16x32_{bf16,f16} / 32x32x16: the same rate for both formats)
__device__ __forceinline__ mi355_f32x4_t MI355_MFMA_16x16x32(st16x8_t a, st16x8_t b, mi355_f32x4_t c) { return MI355_MFMA_BUILTIN_16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ mi355_f32x16_t MI355_MFMA_32x32x16(st16x8_t a, st16x8_t b, mi355_f32x16_t c) { return MI355_MFMA_BUILTIN_32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
  return v;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned lo = __shfl_xor((unsigned)v, o, WAVE);
    unsigned hi = __shfl_xor((unsigned)(v >> 32), o, WAVE);
    unsigned long long w = ((unsigned long long)hi << 32) | lo;
    v = w > v ? w : v;
  }
  return v;
}

// order-preserving float -> uint32 (larger float => larger uint); -0 canonicalised to +0
__device__ __forceinline__ unsigned f2ord(float f) {
  f = f + 0.0f;
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  return __uint_as_float(u);
}

inline hipStream_t S(void* s) { return (hipStream_t)s; }

// one-time-per-DEVICE guard for hipFuncSetAttribute: the attribute lives in the device's copy of the code object, so a process that uses a
// second GPU must set it there too (a process-wide flag left kernels with > 64 KB of dynamic LDS unlaunchable on cuda:1).  The device's bit is
// published only AFTER the attributes are set (a second host thread that raced past a bit set up front could launch before the attribute
// applied: ADVICE r3); racing first callers serialise on the mutex, later calls are one relaxed-cost acquire load.
struct DeviceOnce {
  std::atomic<unsigned long long> done{0};
  std::mutex mu;
  template <class F>
  void once(F&& f) {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long b = 1ull << (d & 63);
    if (done.load(std::memory_order_acquire) & b) return;
    std::lock_guard<std::mutex> g(mu);
    if (done.load(std::memory_order_relaxed) & b) return;
    f();
    done.fetch_or(b, std::memory_order_release);
  }
};

}  // namespace mi355
