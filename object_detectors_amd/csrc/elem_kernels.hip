// HBM-bound elementwise / reduction kernels around the convolutions: training BatchNorm2d +
// LeakyReLU(0.1) forward/backward (darknet.py:15-16,19-20,33), nearest x2 upsample into a
// channel slice (yolohead.py:32,80-81), layout converters.  NHWC bf16, 16-byte vector accesses.
#include "common.h"

#include <cstdlib>

using namespace mi355;

namespace {

struct bf8 {
  float v[8];
};
__device__ __forceinline__ bf8 unpack8(const uint4 u) {
  bf8 r;
  r.v[0] = s2f((bf16_t)(u.x & 0xFFFF)); r.v[1] = s2f((bf16_t)(u.x >> 16));
  r.v[2] = s2f((bf16_t)(u.y & 0xFFFF)); r.v[3] = s2f((bf16_t)(u.y >> 16));
  r.v[4] = s2f((bf16_t)(u.z & 0xFFFF)); r.v[5] = s2f((bf16_t)(u.z >> 16));
  r.v[6] = s2f((bf16_t)(u.w & 0xFFFF)); r.v[7] = s2f((bf16_t)(u.w >> 16));
  return r;
}
__device__ __forceinline__ bf8 ld8(const bf16_t* p) {
  const uint4 u = *(const uint4*)p;
  bf8 r;
  r.v[0] = s2f((bf16_t)(u.x & 0xFFFF)); r.v[1] = s2f((bf16_t)(u.x >> 16));
  r.v[2] = s2f((bf16_t)(u.y & 0xFFFF)); r.v[3] = s2f((bf16_t)(u.y >> 16));
  r.v[4] = s2f((bf16_t)(u.z & 0xFFFF)); r.v[5] = s2f((bf16_t)(u.z >> 16));
  r.v[6] = s2f((bf16_t)(u.w & 0xFFFF)); r.v[7] = s2f((bf16_t)(u.w >> 16));
  return r;
}
__device__ __forceinline__ void st8(bf16_t* p, const bf8& r) {
  uint4 u;
  u.x = (unsigned)f2s(r.v[0]) | ((unsigned)f2s(r.v[1]) << 16);
  u.y = (unsigned)f2s(r.v[2]) | ((unsigned)f2s(r.v[3]) << 16);
  u.z = (unsigned)f2s(r.v[4]) | ((unsigned)f2s(r.v[5]) << 16);
  u.w = (unsigned)f2s(r.v[6]) | ((unsigned)f2s(r.v[7]) << 16);
  *(uint4*)p = u;
}

// stats partials [rows][2][c_pad] -> scale/shift/mean/invstd (+ running stats, nn.BatchNorm2d momentum rule)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int rows, int c, int c_pad, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                                                          float* __restrict__ ss) {
  __shared__ double sh[8][32][2];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int ch = blockIdx.x * 32 + cl;
  double s1 = 0, s2 = 0;
  if (ch < c)
    for (int r = rl; r < rows; r += 8) {
      s1 += (double)partial[(size_t)r * 2 * c_pad + ch];
      s2 += (double)partial[(size_t)r * 2 * c_pad + c_pad + ch];
    }
  sh[rl][cl][0] = s1;
  sh[rl][cl][1] = s2;
  __syncthreads();
  if (rl == 0 && ch < c) {
    for (int r = 1; r < 8; ++r) {
      s1 += sh[r][cl][0];
      s2 += sh[r][cl][1];
    }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0) var = 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[ch] * invstd;
    ss[ch] = sc;
    ss[c + ch] = beta[ch] - (float)mean * sc;
    ss[2 * c + ch] = (float)mean;
    ss[3 * c + ch] = invstd;
    if (rmean) {
      rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * (float)mean;
      const double unb = count > 1 ? var * count / (count - 1) : var;
      rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unb;
    }
  }
}

// The same finalisation in ONE launch for 32 < rows <= 8192 (all but the three 320-px layers of YOLOv3 at batch 32 / 640 px): 8 channels x 128 row parts
// per workgroup, every thread's loads independent (<= 16 rows each, 4 in flight), double accumulation, fixed-order LDS fold.  The
// two-launch form below (bn_partial_kernel + bn_finalize_kernel) costs 6 + 7 us of pure launch latency per layer in the forward chain
// conv -> statistics -> activation, where nothing else can run.
__global__ __launch_bounds__(1024) void bn_finalize_wide_kernel(const float* __restrict__ partial, int rows, int c, int c_pad, double count,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                float momentum, float* __restrict__ rmean, float* __restrict__ rvar,
                                                                float* __restrict__ ss) {
  constexpr int CB = 8, PARTS = 1024 / CB;
  __shared__ double sh[PARTS][CB][2];
  const int cl = threadIdx.x & (CB - 1), rl = threadIdx.x / CB;
  const int ch = blockIdx.x * CB + cl;
  double s1 = 0, s2 = 0;
  if (ch < c) {
    int r = rl;
    for (; r + 3 * PARTS < rows; r += 4 * PARTS) {
      const float a0 = partial[(size_t)r * 2 * c_pad + ch], b0 = partial[(size_t)r * 2 * c_pad + c_pad + ch];
      const float a1 = partial[(size_t)(r + PARTS) * 2 * c_pad + ch], b1 = partial[(size_t)(r + PARTS) * 2 * c_pad + c_pad + ch];
      const float a2 = partial[(size_t)(r + 2 * PARTS) * 2 * c_pad + ch], b2 = partial[(size_t)(r + 2 * PARTS) * 2 * c_pad + c_pad + ch];
      const float a3 = partial[(size_t)(r + 3 * PARTS) * 2 * c_pad + ch], b3 = partial[(size_t)(r + 3 * PARTS) * 2 * c_pad + c_pad + ch];
      s1 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      s2 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; r < rows; r += PARTS) {
      s1 += (double)partial[(size_t)r * 2 * c_pad + ch];
      s2 += (double)partial[(size_t)r * 2 * c_pad + c_pad + ch];
    }
  }
  sh[rl][cl][0] = s1;
  sh[rl][cl][1] = s2;
  __syncthreads();
  // fold PARTS -> 8 in parallel (8 threads per channel), then 8 -> 1
  double t1 = 0, t2 = 0;
  if (rl < 8) {
#pragma unroll
    for (int q = 0; q < PARTS / 8; ++q) {
      t1 += sh[rl * (PARTS / 8) + q][cl][0];
      t2 += sh[rl * (PARTS / 8) + q][cl][1];
    }
  }
  __syncthreads();
  if (rl < 8) {
    sh[rl][cl][0] = t1;
    sh[rl][cl][1] = t2;
  }
  __syncthreads();
  if (rl == 0 && ch < c) {
    s1 = 0;
    s2 = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      s1 += sh[q][cl][0];
      s2 += sh[q][cl][1];
    }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0) var = 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[ch] * invstd;
    ss[ch] = sc;
    ss[c + ch] = beta[ch] - (float)mean * sc;
    ss[2 * c + ch] = (float)mean;
    ss[3 * c + ch] = invstd;
    if (rmean) {
      rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * (float)mean;
      const double unb = count > 1 ? var * count / (count - 1) : var;
      rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unb;
    }
  }
}

// stage 1 of the statistics reduction for layers with many pixel tiles: [rows][2][c_pad] -> [chunks][2][c_pad]
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ partial, int rows, int c_pad, int chunk,
                                                         float* __restrict__ out) {
  __shared__ float sh[8][32][2];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int ch = blockIdx.x * 32 + cl;
  const int r0 = blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
  float s1 = 0.f, s2 = 0.f;
  if (ch < c_pad)
    for (int r = r0 + rl; r < r1; r += 8) {
      s1 += partial[(size_t)r * 2 * c_pad + ch];
      s2 += partial[(size_t)r * 2 * c_pad + c_pad + ch];
    }
  sh[rl][cl][0] = s1;
  sh[rl][cl][1] = s2;
  __syncthreads();
  if (rl == 0 && ch < c_pad) {
    for (int r = 1; r < 8; ++r) {
      s1 += sh[r][cl][0];
      s2 += sh[r][cl][1];
    }
    out[(size_t)blockIdx.y * 2 * c_pad + ch] = s1;
    out[(size_t)blockIdx.y * 2 * c_pad + c_pad + ch] = s2;
  }
}

// SyncBN forward: the same fold as bn_finalize_kernel's first half (double accumulation over the partial rows), but the per-channel
// sums leave as doubles so that the cross-rank all-reduce and the finalisation keep the local path's precision
// (sums64 [2][c_pad]: sum x | sum x*x)
__global__ __launch_bounds__(256) void bn_fold_f64_kernel(const float* __restrict__ partial, int rows, int c, int c_pad, double* __restrict__ sums64) {
  __shared__ double sh[8][32][2];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int ch = blockIdx.x * 32 + cl;
  double s1 = 0, s2 = 0;
  if (ch < c)
    for (int r = rl; r < rows; r += 8) {
      s1 += (double)partial[(size_t)r * 2 * c_pad + ch];
      s2 += (double)partial[(size_t)r * 2 * c_pad + c_pad + ch];
    }
  sh[rl][cl][0] = s1;
  sh[rl][cl][1] = s2;
  __syncthreads();
  if (rl == 0 && ch < c_pad) {
    for (int r = 1; r < 8; ++r) {
      s1 += sh[r][cl][0];
      s2 += sh[r][cl][1];
    }
    sums64[ch] = ch < c ? s1 : 0.0;
    sums64[c_pad + ch] = ch < c ? s2 : 0.0;
  }
}

__global__ __launch_bounds__(256) void bn_finalize_f64_kernel(const double* __restrict__ sums64, int c, int c_pad, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                                              float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ ss) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= c) return;
  const double mean = sums64[ch] / count;
  double var = sums64[c_pad + ch] / count - mean * mean;
  if (var < 0) var = 0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[ch] * invstd;
  ss[ch] = sc;
  ss[c + ch] = beta[ch] - (float)mean * sc;
  ss[2 * c + ch] = (float)mean;
  ss[3 * c + ch] = invstd;
  if (rmean) {
    rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * (float)mean;
    const double unb = count > 1 ? var * count / (count - 1) : var;
    rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unb;
  }
}

// final stage of the fused BN-backward reduction: [rows<=256][2][c_pad] -> sums[2*c] (sum dy | sum dy*xhat)
__global__ __launch_bounds__(256) void bn_bwd_sum_kernel(const float* __restrict__ partial, int rows, int c, int c_pad, float* __restrict__ sums) {
  __shared__ float sh[8][32][2];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int ch = blockIdx.x * 32 + cl;
  float s1 = 0.f, s2 = 0.f;
  if (ch < c)
    for (int r = rl; r < rows; r += 8) {
      s1 += partial[(size_t)r * 2 * c_pad + ch];
      s2 += partial[(size_t)r * 2 * c_pad + c_pad + ch];
    }
  sh[rl][cl][0] = s1;
  sh[rl][cl][1] = s2;
  __syncthreads();
  if (rl == 0 && ch < c) {
    for (int r = 1; r < 8; ++r) {
      s1 += sh[r][cl][0];
      s2 += sh[r][cl][1];
    }
    sums[ch] = s1;
    sums[c + ch] = s2;
  }
}

// eval-mode scale/shift from running stats
__global__ void bn_eval_kernel(int c, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rmean,
                               const float* __restrict__ rvar, float eps, float* __restrict__ ss) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const float invstd = 1.0f / sqrtf(rvar[ch] + eps);
  const float sc = gamma[ch] * invstd;
  ss[ch] = sc;
  ss[c + ch] = beta[ch] - rmean[ch] * sc;
  ss[2 * c + ch] = rmean[ch];
  ss[3 * c + ch] = invstd;
}

// a = lrelu(z*scale+shift) [+ residual]
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const bf16_t* __restrict__ z, int z_ld, const float* __restrict__ ss, int c,
                                                         long long pixels, float slope, const bf16_t* __restrict__ res, int res_ld,
                                                         bf16_t* __restrict__ out, int out_ld) {
  const int groups = c >> 3;
  const long long total = pixels * groups;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / groups;
    const int c0 = (int)(i - m * groups) << 3;
    bf8 v = ld8(z + m * z_ld + c0);
    const float4 sa = *(const float4*)(ss + c0), sb = *(const float4*)(ss + c0 + 4);
    const float4 ha = *(const float4*)(ss + c + c0), hb = *(const float4*)(ss + c + c0 + 4);
    const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
    const float sh[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float y = v.v[k] * sc[k] + sh[k];
      v.v[k] = y > 0.f ? y : y * slope;
    }
    if (res) {
      const bf8 r = ld8(res + m * res_ld + c0);
#pragma unroll
      for (int k = 0; k < 8; ++k) v.v[k] += r.v[k];
    }
    st8(out + m * out_ld + c0, v);
  }
}

// per-channel sums of dy and dy*xhat, dy = (g1 [+ g2]) * lrelu'(y).  Block = (c/8) channel groups x
// (256/(c/8)) pixel lanes; partial sums combined in LDS, then one atomic per channel per block.
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16_t* __restrict__ g1, int g1_ld, const bf16_t* __restrict__ g2, int g2_ld,
                                                            const bf16_t* __restrict__ z, int z_ld, const float* __restrict__ ss, int c,
                                                            long long pixels, float slope, float* __restrict__ sums, int pix_per_block, int groups,
                                                            float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float red[256][17];
  // groups = 8-channel groups per workgroup (power of two <= 256); blockIdx.y picks the channel slab when c/8 > groups
  const int gl = threadIdx.x % groups, pl = threadIdx.x / groups, npl = 256 / groups;
  const int cbase = blockIdx.y * groups * 8;
  const bool live = cbase + (gl << 3) < c;          // the last channel slab may be partly empty (channel counts that are not a multiple of groups * 8)
  const int c0 = live ? cbase + (gl << 3) : 0;
  float sc[8], sh[8], mu[8], is[8], a1[8], a2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = ss[c0 + k];
    sh[k] = ss[c + c0 + k];
    mu[k] = ss[2 * c + c0 + k];
    is[k] = ss[3 * c + c0 + k];
    a1[k] = 0.f;
    a2[k] = 0.f;
  }
  const long long mA = (long long)blockIdx.x * pix_per_block, mB = min(pixels, mA + pix_per_block);
  for (long long m = mA + pl; m < mB; m += 4 * npl) {
    uint4 gu[4], zu[4], hu[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {   // issue all loads first: 8-12 x 16 B in flight per lane
      const long long mm = m + (long long)u * npl;
      const bool ok = mm < mB && live;
      gu[u] = ok ? *(const uint4*)(g1 + mm * g1_ld + c0) : make_uint4(0, 0, 0, 0);
      zu[u] = ok ? *(const uint4*)(z + mm * z_ld + c0) : make_uint4(0, 0, 0, 0);
      if (g2) hu[u] = ok ? *(const uint4*)(g2 + mm * g2_ld + c0) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bf8 g = unpack8(gu[u]);
      if (g2) {
        const bf8 h = unpack8(hu[u]);
#pragma unroll
        for (int k = 0; k < 8; ++k) g.v[k] += h.v[k];
      }
      const bf8 zz = unpack8(zu[u]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float y = zz.v[k] * sc[k] + sh[k];
        const float dy = y > 0.f ? g.v[k] : g.v[k] * slope;   // zero-filled tail lanes contribute dy = 0
        a1[k] += dy;
        a2[k] += dy * ((zz.v[k] - mu[k]) * is[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    red[threadIdx.x][k] = a1[k];
    red[threadIdx.x][8 + k] = a2[k];
  }
  __syncthreads();
  // thread t < 2*cb handles (which = t / cb, channel = t % cb) of this workgroup's cb = groups*8 channels
  const int cb = groups * 8;
  // Fixed-order form (part != null; what both engines use since round 4): every workgroup stores its 2 * cb sums as one row of `part` and
  // bn_bwd_fold_rows_kernel (next launch on the stream) adds the rows in row order: the result does not depend on the order in which
  // workgroups finish, so the step's gradients are bit-reproducible (the atomic form differed by ~6e-4 of max from run to run).
  // (Measured alternative: a ticket per channel slab and the last arriver folding the rows inside this launch - agent-scope release by every
  //  workgroup, acquire + 256 KB fold by the last one - cost 8-12 us per launch against 3-4 us for the second launch.)
  float* prow = part ? part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)(2 * cb) : nullptr;
  for (int t = threadIdx.x; t < 2 * cb; t += 256) {
    const int which = t / cb, ch = t - which * cb;
    const int gi = ch >> 3, k = ch & 7;
    float s = 0.f;
    for (int p2 = 0; p2 < npl; ++p2) s += red[p2 * groups + gi][which * 8 + k];
    if (part) prow[t] = s;
    else if (cbase + ch < c) atomicAdd(sums + which * c + cbase + ch, s);
  }
}

// sums[which * c + slab * cb + ch] = sum over the nb rows (in row order within each of 32 row groups, then over the groups in group order) of
// part[slab][row][which * cb + ch].  Grid (column groups of 32 floats, slabs); 256 threads = 8 float4 columns x 32 row groups, every load of a
// thread in flight at once (nb <= 512: at most 16).
__global__ __launch_bounds__(256) void bn_bwd_fold_rows_kernel(const float* __restrict__ part, int nb, int cb, int c, float* __restrict__ sums) {
  __shared__ __attribute__((aligned(16))) float fold[32][32];
  const int w = 2 * cb;                                    // floats per row
  const int col4 = blockIdx.x * 8 + (threadIdx.x & 7), rg = threadIdx.x >> 3;
  const bool live = col4 * 4 < w;
  const float4* base = (const float4*)(part + (size_t)blockIdx.y * nb * (size_t)w) + col4;
  float4 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int r = rg + u * 32;
    v[u] = (live && r < nb) ? base[(size_t)r * (w >> 2)] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 acc = v[0];
#pragma unroll
  for (int u = 1; u < 16; ++u) {
    acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
  }
  *(float4*)&fold[rg][(threadIdx.x & 7) * 4] = acc;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int t = blockIdx.x * 32 + threadIdx.x;           // column of the row
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 32; ++g) s += fold[g][threadIdx.x];
    const int which = t / cb, ch = t - which * cb;
    const int cg = blockIdx.y * cb + ch;
    if (t < w && cg < c) sums[which * c + cg] = s;
  }
}

// dz = scale * (dy - mean(dy) - xhat * mean(dy*xhat));  also dgamma/dbeta (block 0)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ g1, int g1_ld, const bf16_t* __restrict__ g2, int g2_ld,
                                                           const bf16_t* __restrict__ z, int z_ld, const float* __restrict__ ss,
                                                           const float* __restrict__ sums, int c, long long pixels, float slope,
                                                           bf16_t* __restrict__ dz, int dz_ld, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
  const int groups = c >> 3;
  const long long total = pixels * groups;
  const float inv = 1.0f / (float)pixels;
  if (blockIdx.x == 0 && dgamma)
    for (int ch = threadIdx.x; ch < c; ch += 256) {
      dbeta[ch] += sums[ch];
      dgamma[ch] += sums[c + ch];
    }
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / groups;
    const int c0 = (int)(i - m * groups) << 3;
    bf8 g = ld8(g1 + m * g1_ld + c0);
    if (g2) {
      const bf8 h = ld8(g2 + m * g2_ld + c0);
#pragma unroll
      for (int k = 0; k < 8; ++k) g.v[k] += h.v[k];
    }
    const bf8 zz = ld8(z + m * z_ld + c0);
    bf8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ch = c0 + k;
      const float sc = ss[ch], sh = ss[c + ch], mu = ss[2 * c + ch], is = ss[3 * c + ch];
      const float y = zz.v[k] * sc + sh;
      const float dy = y > 0.f ? g.v[k] : g.v[k] * slope;
      const float xh = (zz.v[k] - mu) * is;
      o.v[k] = sc * (dy - sums[ch] * inv - xh * sums[c + ch] * inv);
    }
    st8(dz + m * dz_ld + c0, o);
  }
}

// Row-mapped forms of the two elementwise passes (channels/8 a power of two <= 256, every Darknet layer): a thread keeps ONE
// group of 8 channels and walks over pixels, so the per-channel constants live in registers.  The grid-stride forms above
// re-load 32 (forward) / 48 (backward) table dwords per 16 bytes of data and pay a 64-bit division per element group;
// their table loads, not the tensor, filled the vector-memory pipe (3.6 TB/s against 5.4 for the forward pass).
template <int U>
__global__ __launch_bounds__(256) void bn_act_fwd_rows_kernel(const bf16_t* __restrict__ z, int z_ld, const float* __restrict__ ss, int c,
                                                              long long pixels, float slope, const bf16_t* __restrict__ res, int res_ld,
                                                              bf16_t* __restrict__ out, int out_ld, int pix_per_block, int gshift) {
  const int groups = 1 << gshift, npl = 256 >> gshift;
  const int gl = threadIdx.x & (groups - 1), pl = threadIdx.x >> gshift;
  const int c0 = gl << 3;
  float sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = ss[c0 + k];
    sh[k] = ss[c + c0 + k];
  }
  const long long mA = (long long)blockIdx.x * pix_per_block, mB = min(pixels, mA + pix_per_block);
  for (long long m = mA + pl; m < mB; m += U * npl) {
    uint4 zu[U], ru[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long mm = m + (long long)u * npl;
      if (mm < mB) {
        zu[u] = *(const uint4*)(z + mm * z_ld + c0);
        if (res) ru[u] = *(const uint4*)(res + mm * res_ld + c0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long mm = m + (long long)u * npl;
      if (mm >= mB) break;
      bf8 v = unpack8(zu[u]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float y = v.v[k] * sc[k] + sh[k];
        v.v[k] = y > 0.f ? y : y * slope;
      }
      if (res) {
        const bf8 r = unpack8(ru[u]);
#pragma unroll
        for (int k = 0; k < 8; ++k) v.v[k] += r.v[k];
      }
      st8(out + mm * out_ld + c0, v);
    }
  }
}

template <int U>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows_kernel(const bf16_t* __restrict__ g1, int g1_ld, const bf16_t* __restrict__ g2, int g2_ld,
                                                                const bf16_t* __restrict__ z, int z_ld, const float* __restrict__ ss,
                                                                const float* __restrict__ sums, int c, long long pixels, float slope,
                                                                bf16_t* __restrict__ dz, int dz_ld, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int pix_per_block, int gshift) {
  const int groups = 1 << gshift, npl = 256 >> gshift;
  const int gl = threadIdx.x & (groups - 1), pl = threadIdx.x >> gshift;
  const int c0 = gl << 3;
  const float inv = 1.0f / (float)pixels;
  if (blockIdx.x == 0 && dgamma)
    for (int ch = threadIdx.x; ch < c; ch += 256) {
      dbeta[ch] += sums[ch];
      dgamma[ch] += sums[c + ch];
    }
  float sc[8], sh[8], mu[8], is[8], m1[8], m2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = ss[c0 + k];
    sh[k] = ss[c + c0 + k];
    mu[k] = ss[2 * c + c0 + k];
    is[k] = ss[3 * c + c0 + k];
    m1[k] = sums[c0 + k] * inv;
    m2[k] = sums[c + c0 + k] * inv;
  }
  const long long mA = (long long)blockIdx.x * pix_per_block, mB = min(pixels, mA + pix_per_block);
  for (long long m = mA + pl; m < mB; m += U * npl) {
    uint4 gu[U], hu[U], zu[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long mm = m + (long long)u * npl;
      if (mm < mB) {
        gu[u] = *(const uint4*)(g1 + mm * g1_ld + c0);
        zu[u] = *(const uint4*)(z + mm * z_ld + c0);
        if (g2) hu[u] = *(const uint4*)(g2 + mm * g2_ld + c0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long mm = m + (long long)u * npl;
      if (mm >= mB) break;
      bf8 g = unpack8(gu[u]);
      if (g2) {
        const bf8 h = unpack8(hu[u]);
#pragma unroll
        for (int k = 0; k < 8; ++k) g.v[k] += h.v[k];
      }
      const bf8 zz = unpack8(zu[u]);
      bf8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float y = zz.v[k] * sc[k] + sh[k];
        const float dy = y > 0.f ? g.v[k] : g.v[k] * slope;
        const float xh = (zz.v[k] - mu[k]) * is[k];
        o.v[k] = sc[k] * (dy - m1[k] - xh * m2[k]);
      }
      st8(dz + mm * dz_ld + c0, o);
    }
  }
}

__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const bf16_t* __restrict__ x, int x_ld, int n, int h, int w, int c,
                                                             bf16_t* __restrict__ out, int out_ld) {
  const int groups = c >> 3;
  const long long total = (long long)n * (2 * h) * (2 * w) * groups;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int g = (int)(i % groups);
    const long long p = i / groups;
    const int ox = (int)(p % (2 * w)), oy = (int)((p / (2 * w)) % (2 * h)), b = (int)(p / ((long long)4 * w * h));
    const uint4 v = *(const uint4*)(x + ((long long)(b * h + (oy >> 1)) * w + (ox >> 1)) * x_ld + g * 8);
    *(uint4*)(out + p * out_ld + g * 8) = v;
  }
}

__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const bf16_t* __restrict__ gq, int g_ld, int n, int h, int w, int c,
                                                             bf16_t* __restrict__ out, int out_ld) {
  const int groups = c >> 3;
  const long long total = (long long)n * h * w * groups;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int g = (int)(i % groups);
    const long long p = i / groups;
    const int x = (int)(p % w), y = (int)((p / w) % h), b = (int)(p / ((long long)w * h));
    bf8 a;
#pragma unroll
    for (int k = 0; k < 8; ++k) a.v[k] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const bf8 v = ld8(gq + ((long long)(b * 2 * h + 2 * y + dy) * (2 * w) + 2 * x + dx) * g_ld + g * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) a.v[k] += v.v[k];
      }
    st8(out + p * out_ld + g * 8, a);
  }
}

// out = a + b (bf16, channel slices) — gradient joins of the concat branches
__global__ __launch_bounds__(256) void add_kernel(const bf16_t* __restrict__ a, int a_ld, const bf16_t* __restrict__ b, int b_ld, int c,
                                                  long long pixels, bf16_t* __restrict__ out, int out_ld) {
  const int groups = c >> 3;
  const long long total = pixels * groups;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / groups;
    const int c0 = (int)(i - m * groups) << 3;
    bf8 v = ld8(a + m * a_ld + c0);
    const bf8 w = ld8(b + m * b_ld + c0);
#pragma unroll
    for (int k = 0; k < 8; ++k) v.v[k] += w.v[k];
    st8(out + m * out_ld + c0, v);
  }
}

template <bool OUT_BF16>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, int n, int c, int hw, void* __restrict__ out, int out_ld) {
  const long long total = (long long)n * hw;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = i / hw, p = i - b * hw;
    for (int ch = 0; ch < c; ++ch) {
      const float v = x[(b * c + ch) * hw + p];
      if (OUT_BF16) ((bf16_t*)out)[i * out_ld + ch] = f2s(v);
      else ((float*)out)[i * out_ld + ch] = v;
    }
  }
}

template <bool IN_BF16>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const void* __restrict__ x, int x_ld, int n, int c, int hw, float* __restrict__ out) {
  const long long total = (long long)n * c * hw;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long p = i % hw;
    const int ch = (int)((i / hw) % c);
    const long long b = i / ((long long)hw * c);
    const long long src = (b * hw + p) * x_ld + ch;
    out[i] = IN_BF16 ? s2f(((const bf16_t*)x)[src]) : ((const float*)x)[src];
  }
}

// row-mapped elementwise forms: channels/8 a power of two <= 256; 16 pixels per pixel lane, at most 16384 workgroups
inline bool rows_form(int c, long long pixels, int* gshift, int* ppb, int* blocks) {
  const int groups = c / 8;
  if (c <= 0 || c % 8 != 0 || groups > 256 || (groups & (groups - 1)) != 0) return false;
  int sh = 0;
  while ((1 << sh) < groups) ++sh;
  const int npl = 256 / groups;
  long long per = (long long)npl * 16;
  long long nb = (pixels + per - 1) / per;
  if (nb > 16384) {
    per = ((pixels + 16383) / 16384 + npl - 1) / npl * npl;
    nb = (pixels + per - 1) / per;
  }
  if (per > 0x7FFFFFFFll) return false;
  *gshift = sh;
  *ppb = (int)per;
  *blocks = (int)nb;
  return true;
}

inline int grid_for(long long total) { return (int)min((long long)256 * 16, (total + 255) / 256); }

}  // namespace


#if !MI355_F16      // format-independent (fp32) parts exist once, in the bf16 object
// ---- fused optimizer steps on the flat fp32 parameter / gradient buffers ------------------------------------------------------
// torch.optim.SGD (yolo/procedures/initialize.py:38) and torch.optim.Adam (initialize.py:41) semantics, one pass over
// {w, g, state}: 16-B accesses, grid-stride.  HBM bound: 20 B/param (SGD), 28 B/param (Adam).
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ mom, long long n4, long long n,
                                                   float lr, float momentum, float dampening, float wd, float gscale, int nesterov, int first,
                                                   int zero_grad, const int* __restrict__ skip) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (skip && *skip) {      // amp overflow (train_one_epoch.py:88-96 under apex dynamic loss scaling): the step is skipped, gradients still cleared
    if (zero_grad)
      for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] = 0.f;
    return;
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 wv = ((const float4*)w)[i], gv = ((const float4*)g)[i], mv = ((const float4*)mom)[i];
    float* wp = (float*)&wv; float* gp = (float*)&gv; float* mp = (float*)&mv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float d = gp[j] * gscale + wd * wp[j];
      if (momentum != 0.f) {
        mp[j] = first ? d : momentum * mp[j] + (1.f - dampening) * d;
        d = nesterov ? d + momentum * mp[j] : mp[j];
      }
      wp[j] -= lr * d;
    }
    ((float4*)w)[i] = wv;
    ((float4*)mom)[i] = mv;
    if (zero_grad) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // tail (n not a multiple of 4)
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float d = g[i] * gscale + wd * w[i];
    if (momentum != 0.f) {
      mom[i] = first ? d : momentum * mom[i] + (1.f - dampening) * d;
      d = nesterov ? d + momentum * mom[i] : mom[i];
    }
    w[i] -= lr * d;
    if (zero_grad) g[i] = 0.f;
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                                                    float lr, float b1, float b2, float eps, float wd, float gscale, float bc1, float bc2_sqrt,
                                                    int zero_grad, const int* __restrict__ skip) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (skip && *skip) {
    if (zero_grad)
      for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] = 0.f;
    return;
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float wi = w[i];
    const float d = g[i] * gscale + wd * wi;
    const float mi = b1 * m[i] + (1.f - b1) * d;
    const float vi = b2 * v[i] + (1.f - b2) * d * d;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    w[i] = wi - (lr / bc1) * (mi / denom);
    if (zero_grad) g[i] = 0.f;
  }
}

// amp overflow check (apex LossScaler.update_scale's `_has_overflow`): flag = 1 when any gradient is inf / nan.  One read pass, 4 B/param.
__global__ __launch_bounds__(256) void nonfinite_kernel(const float* __restrict__ g, long long n4, long long n, int* __restrict__ flag) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  bool bad = false;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = ((const float4*)g)[i];
    // (x - x) is 0 for finite x and nan for inf / nan
    const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
    bad |= !(t == 0.f);
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) bad |= !((g[i] - g[i]) == 0.f);
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) *flag = 1;      // every writer stores the same value: no atomic needed
}

#endif

extern "C" {

#if !MI355_F16
int mi355det_bn_finalize(const float* stats, int32_t rows, int32_t c, int32_t c_pad, int64_t count, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, float* scale_shift, void* stream) {
  if (c <= 0 || rows <= 0 || count <= 0) return fail(MI355DET_EINVAL, "%s: bad arguments", "bn_finalize");
  if (rows > 32 && rows <= 8192) {
    hipLaunchKernelGGL(bn_finalize_wide_kernel, dim3((c + 7) / 8), dim3(1024), 0, S(stream), stats, rows, c, c_pad, (double)count, gamma, beta, eps,
                       momentum, running_mean, running_var, scale_shift);
    return check_launch("bn_finalize");
  }
  if (rows > 256) {
    // two stages (deterministic): 64 row-chunks reduced in parallel into the 64 spare rows behind the partials.  (One launch whose
    // last workgroup per channel slab finalises was measured too: the device-scope stores / loads it needs make it 15 us against
    // 7 + 6 us for the two launches, 1011-1017 vs 1020 images/s on the same box; with __threadfence() it writes the L2 back: -5 %.)
    const int chunks = 64, chunk = (rows + chunks - 1) / chunks;
    float* scratch = const_cast<float*>(stats) + (size_t)rows * 2 * c_pad;
    hipLaunchKernelGGL(bn_partial_kernel, dim3((c_pad + 31) / 32, chunks), dim3(256), 0, S(stream), stats, rows, c_pad, chunk, scratch);
    stats = scratch;
    rows = chunks;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 31) / 32), dim3(256), 0, S(stream), stats, rows, c, c_pad, (double)count, gamma, beta, eps,
                     momentum, running_mean, running_var, scale_shift);
  return check_launch("bn_finalize");
}

int mi355det_bn_bwd_sum_partials(const float* partials, int32_t rows, int32_t c, int32_t c_pad, float* sums, void* stream) {
  if (c <= 0 || rows <= 0 || c_pad < c) return fail(MI355DET_EINVAL, "%s: bad arguments", "bn_bwd_sum_partials");
  if (rows > 256) {
    const int chunks = 64, chunk = (rows + chunks - 1) / chunks;
    float* scratch = const_cast<float*>(partials) + (size_t)rows * 2 * c_pad;     // the 64 spare rows behind the partials
    hipLaunchKernelGGL(bn_partial_kernel, dim3((c_pad + 31) / 32, chunks), dim3(256), 0, S(stream), partials, rows, c_pad, chunk, scratch);
    partials = scratch;
    rows = chunks;
  }
  hipLaunchKernelGGL(bn_bwd_sum_kernel, dim3((c + 31) / 32), dim3(256), 0, S(stream), partials, rows, c, c_pad, sums);
  return check_launch("bn_bwd_sum_partials");
}

int mi355det_bn_fold_partials_f64(const float* stats, int32_t rows, int32_t c, int32_t c_pad, double* sums64, void* stream) {
  if (c <= 0 || rows <= 0 || c_pad < c || !stats || !sums64) return fail(MI355DET_EINVAL, "%s: bad arguments", "bn_fold_partials_f64");
  if (rows > 256) {
    const int chunks = 64, chunk = (rows + chunks - 1) / chunks;
    float* scratch = const_cast<float*>(stats) + (size_t)rows * 2 * c_pad;       // the 64 spare rows behind the partials, as bn_finalize
    hipLaunchKernelGGL(bn_partial_kernel, dim3((c_pad + 31) / 32, chunks), dim3(256), 0, S(stream), stats, rows, c_pad, chunk, scratch);
    stats = scratch;
    rows = chunks;
  }
  hipLaunchKernelGGL(bn_fold_f64_kernel, dim3((c_pad + 31) / 32), dim3(256), 0, S(stream), stats, rows, c, c_pad, sums64);
  return check_launch("bn_fold_partials_f64");
}

int mi355det_bn_finalize_f64(const double* sums64, int32_t c, int32_t c_pad, int64_t count, const float* gamma, const float* beta, float eps,
                             float momentum, float* running_mean, float* running_var, float* scale_shift, void* stream) {
  if (c <= 0 || c_pad < c || count <= 0 || !sums64) return fail(MI355DET_EINVAL, "%s: bad arguments", "bn_finalize_f64");
  hipLaunchKernelGGL(bn_finalize_f64_kernel, dim3((c + 255) / 256), dim3(256), 0, S(stream), sums64, c, c_pad, (double)count, gamma, beta, eps, momentum,
                     running_mean, running_var, scale_shift);
  return check_launch("bn_finalize_f64");
}

int mi355det_bn_eval_scale_shift(int32_t c, const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                                 float* scale_shift, void* stream) {
  if (c <= 0) return fail(MI355DET_EINVAL, "%s: bad arguments", "bn_eval");
  hipLaunchKernelGGL(bn_eval_kernel, dim3((c + 255) / 256), dim3(256), 0, S(stream), c, gamma, beta, running_mean, running_var, eps, scale_shift);
  return check_launch("bn_eval");
}

#endif

int mi355det_bn_act_fwd(const void* z, int32_t z_ld, const float* scale_shift, int32_t c, int64_t pixels, float slope, const void* residual,
                        int32_t res_ld, void* out, int32_t out_ld, void* stream) {
  if (c <= 0 || c % 8 != 0 || pixels <= 0) return fail(MI355DET_EINVAL, "%s: channels must be a multiple of 8", "bn_act_fwd");
  int gshift, ppb, blocks;
  if (rows_form(c, pixels, &gshift, &ppb, &blocks)) {
    hipLaunchKernelGGL(bn_act_fwd_rows_kernel<4>, dim3(blocks), dim3(256), 0, S(stream), (const bf16_t*)z, z_ld, scale_shift, c, (long long)pixels,
                       slope, (const bf16_t*)residual, res_ld, (bf16_t*)out, out_ld, ppb, gshift);
    return check_launch("bn_act_fwd");
  }
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(grid_for(pixels * (c / 8))), dim3(256), 0, S(stream), (const bf16_t*)z, z_ld, scale_shift, c,
                     (long long)pixels, slope, (const bf16_t*)residual, res_ld, (bf16_t*)out, out_ld);
  return check_launch("bn_act_fwd");
}

struct BnReduceGeom {
  int gb, slabs, blocks;
  long long ppb;
};
static BnReduceGeom bn_reduce_geom(int c, long long pixels) {
  // a workgroup covers at most 64-128 channels (whole 128-byte lines per pixel) and more pixels instead: every workgroup ends with
  // 2 * (its channels) partial sums (one row of the workspace, or float atomics in the legacy form: ~24 G atomics/s device-wide measured),
  // so wide layers are cut into channel slabs (blockIdx.y) and large tensors get no more than ~2 workgroups per CU
  // (any multiple of 8 channels: the last slab of a channel count that is not gb * 8 * k has idle channel groups)
  const int groups = c / 8;
  BnReduceGeom g;
  g.gb = groups >= 32 ? 16 : (groups > 8 ? 8 : 1);
  while (g.gb * 2 <= groups && g.gb < 8) g.gb *= 2;
  g.slabs = (groups + g.gb - 1) / g.gb;
  const int npl = 256 / g.gb;
  g.ppb = (long long)npl * 32;   // 32 pixels per pixel-lane
  long long blocks = (pixels + g.ppb - 1) / g.ppb;
  const long long cap = 512 / g.slabs > 64 ? 512 / g.slabs : 64;
  if (blocks > cap) {
    g.ppb = ((pixels + cap - 1) / cap + npl * 4 - 1) / (npl * 4) * (npl * 4);
    blocks = (pixels + g.ppb - 1) / g.ppb;
  }
  g.blocks = (int)blocks;
  return g;
}

size_t mi355det_bn_act_bwd_reduce_workspace(int32_t c, int64_t pixels) {
  if (c <= 0 || c % 8 != 0 || pixels <= 0) return 0;
  const BnReduceGeom g = bn_reduce_geom(c, pixels);
  return (size_t)g.slabs * g.blocks * (size_t)(2 * g.gb * 8) * sizeof(float);
}

static int bn_reduce_launch(const void* g1, int32_t g1_ld, const void* g2, int32_t g2_ld, const void* z, int32_t z_ld, const float* scale_shift, int32_t c,
                            int64_t pixels, float slope, float* sums, void* workspace, size_t workspace_bytes, void* stream, const char* what) {
  if (c <= 0 || c % 8 != 0) return fail(MI355DET_EINVAL, "%s: channels must be a multiple of 8 (got c=%lld)", what, c);
  const BnReduceGeom g = bn_reduce_geom(c, pixels);
  float* part = nullptr;
  if (workspace) {
    if (g.blocks > 512 || workspace_bytes < mi355det_bn_act_bwd_reduce_workspace(c, pixels) || ((uintptr_t)workspace & 15))
      return fail(MI355DET_EINVAL, "%s: workspace too small or misaligned (%lld bytes given)", what, (long long)workspace_bytes);
    part = (float*)workspace;
  }
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(g.blocks, g.slabs), dim3(256), 0, S(stream), (const bf16_t*)g1, g1_ld, (const bf16_t*)g2, g2_ld,
                     (const bf16_t*)z, z_ld, scale_shift, c, (long long)pixels, slope, sums, (int)g.ppb, g.gb, part);
  if (part) {
    const int cb = g.gb * 8;
    hipLaunchKernelGGL(bn_bwd_fold_rows_kernel, dim3((2 * cb + 31) / 32, g.slabs), dim3(256), 0, S(stream), (const float*)part, g.blocks, cb, c, sums);
  }
  return check_launch(what);
}

int mi355det_bn_act_bwd_reduce(const void* g1, int32_t g1_ld, const void* g2, int32_t g2_ld, const void* z, int32_t z_ld, const float* scale_shift,
                               int32_t c, int64_t pixels, float slope, float* sums, void* stream) {
  return bn_reduce_launch(g1, g1_ld, g2, g2_ld, z, z_ld, scale_shift, c, pixels, slope, sums, nullptr, 0, stream, "bn_act_bwd_reduce");
}

int mi355det_bn_act_bwd_reduce_det(const void* g1, int32_t g1_ld, const void* g2, int32_t g2_ld, const void* z, int32_t z_ld, const float* scale_shift,
                                   int32_t c, int64_t pixels, float slope, float* sums, void* workspace, size_t workspace_bytes, void* stream) {
  if (!workspace) return fail(MI355DET_EINVAL, "%s: null workspace", "bn_act_bwd_reduce_det");
  return bn_reduce_launch(g1, g1_ld, g2, g2_ld, z, z_ld, scale_shift, c, pixels, slope, sums, workspace, workspace_bytes, stream, "bn_act_bwd_reduce_det");
}

int mi355det_bn_act_bwd_apply(const void* g1, int32_t g1_ld, const void* g2, int32_t g2_ld, const void* z, int32_t z_ld, const float* scale_shift,
                              const float* sums, const float* gamma, int32_t c, int64_t pixels, float slope, void* dz, int32_t dz_ld, float* dgamma,
                              float* dbeta, void* stream) {
  (void)gamma;
  if (c <= 0 || c % 8 != 0 || pixels <= 0) return fail(MI355DET_EINVAL, "%s: channels must be a multiple of 8", "bn_act_bwd_apply");
  int gshift, ppb, blocks;
  if (rows_form(c, pixels, &gshift, &ppb, &blocks)) {
    hipLaunchKernelGGL(bn_bwd_apply_rows_kernel<4>, dim3(blocks), dim3(256), 0, S(stream), (const bf16_t*)g1, g1_ld, (const bf16_t*)g2, g2_ld,
                       (const bf16_t*)z, z_ld, scale_shift, sums, c, (long long)pixels, slope, (bf16_t*)dz, dz_ld, dgamma, dbeta, ppb, gshift);
    return check_launch("bn_act_bwd_apply");
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(pixels * (c / 8))), dim3(256), 0, S(stream), (const bf16_t*)g1, g1_ld, (const bf16_t*)g2,
                     g2_ld, (const bf16_t*)z, z_ld, scale_shift, sums, c, (long long)pixels, slope, (bf16_t*)dz, dz_ld, dgamma, dbeta);
  return check_launch("bn_act_bwd_apply");
}

int mi355det_upsample2x_fwd(const void* x, int32_t x_ld, int32_t n, int32_t h, int32_t w, int32_t c, void* out, int32_t out_ld, void* stream) {
  if (c % 8 != 0) return fail(MI355DET_EINVAL, "%s: channels must be a multiple of 8", "upsample2x_fwd");
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(grid_for((long long)n * 4 * h * w * (c / 8))), dim3(256), 0, S(stream), (const bf16_t*)x, x_ld, n, h,
                     w, c, (bf16_t*)out, out_ld);
  return check_launch("upsample2x_fwd");
}

int mi355det_upsample2x_bwd(const void* g, int32_t g_ld, int32_t n, int32_t h, int32_t w, int32_t c, void* out, int32_t out_ld, void* stream) {
  if (c % 8 != 0) return fail(MI355DET_EINVAL, "%s: channels must be a multiple of 8", "upsample2x_bwd");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for((long long)n * h * w * (c / 8))), dim3(256), 0, S(stream), (const bf16_t*)g, g_ld, n, h, w,
                     c, (bf16_t*)out, out_ld);
  return check_launch("upsample2x_bwd");
}

int mi355det_add_bf16(const void* a, int32_t a_ld, const void* b, int32_t b_ld, int32_t c, int64_t pixels, void* out, int32_t out_ld, void* stream) {
  if (c % 8 != 0) return fail(MI355DET_EINVAL, "%s: channels must be a multiple of 8", "add_bf16");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(pixels * (c / 8))), dim3(256), 0, S(stream), (const bf16_t*)a, a_ld, (const bf16_t*)b, b_ld, c,
                     (long long)pixels, (bf16_t*)out, out_ld);
  return check_launch("add_bf16");
}

int mi355det_nchw_f32_to_nhwc(const float* x, int32_t n, int32_t c, int32_t h, int32_t w, void* out, int out_is_bf16, int32_t out_ld, void* stream) {
  const long long total = (long long)n * h * w;
  if (out_is_bf16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<true>, dim3(grid_for(total)), dim3(256), 0, S(stream), x, n, c, h * w, out, out_ld);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<false>, dim3(grid_for(total)), dim3(256), 0, S(stream), x, n, c, h * w, out, out_ld);
  return check_launch("nchw_f32_to_nhwc");
}

int mi355det_nhwc_to_nchw_f32(const void* x, int x_is_bf16, int32_t x_ld, int32_t n, int32_t c, int32_t h, int32_t w, float* out, void* stream) {
  const long long total = (long long)n * c * h * w;
  if (x_is_bf16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<true>, dim3(grid_for(total)), dim3(256), 0, S(stream), x, x_ld, n, c, h * w, out);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<false>, dim3(grid_for(total)), dim3(256), 0, S(stream), x, x_ld, n, c, h * w, out);
  return check_launch("nhwc_to_nchw_f32");
}

#if !MI355_F16
int mi355det_sgd_step_guarded(float* w, float* g, float* momentum_buf, int64_t n, float lr, float momentum, float dampening, float weight_decay,
                              float grad_scale, int nesterov, int first_step, int zero_grad, const int32_t* skip_flag, void* stream) {
  if (n <= 0) return MI355DET_OK;
  if (((uintptr_t)w | (uintptr_t)g | (uintptr_t)momentum_buf) & 15) return fail(MI355DET_EINVAL, "%s: buffers must be 16-byte aligned", "sgd_step");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, S(stream), w, g, momentum_buf, (long long)(n / 4), (long long)n, lr, momentum,
                     dampening, weight_decay, grad_scale, nesterov, first_step, zero_grad, (const int*)skip_flag);
  return check_launch("sgd_step");
}

int mi355det_sgd_step(float* w, float* g, float* momentum_buf, int64_t n, float lr, float momentum, float dampening, float weight_decay,
                      float grad_scale, int nesterov, int first_step, int zero_grad, void* stream) {
  return mi355det_sgd_step_guarded(w, g, momentum_buf, n, lr, momentum, dampening, weight_decay, grad_scale, nesterov, first_step, zero_grad, nullptr,
                                   stream);
}

int mi355det_adam_step_guarded(float* w, float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, float grad_scale, int32_t step, int zero_grad, const int32_t* skip_flag, void* stream) {
  if (n <= 0) return MI355DET_OK;
  if (step < 1) return fail(MI355DET_EINVAL, "%s: step counts from 1", "adam_step");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), w, g, exp_avg, exp_avg_sq, (long long)n, lr, beta1, beta2, eps,
                     weight_decay, grad_scale, bc1, bc2s, zero_grad, (const int*)skip_flag);
  return check_launch("adam_step");
}

int mi355det_adam_step(float* w, float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float grad_scale, int32_t step, int zero_grad, void* stream) {
  return mi355det_adam_step_guarded(w, g, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, grad_scale, step, zero_grad, nullptr, stream);
}

int mi355det_grad_nonfinite(const float* g, int64_t n, int32_t* flag, void* stream) {
  if (n <= 0 || !g || !flag) return fail(MI355DET_EINVAL, "%s: bad arguments", "grad_nonfinite");
  if ((uintptr_t)g & 15) return fail(MI355DET_EINVAL, "%s: buffer must be 16-byte aligned", "grad_nonfinite");
  hipLaunchKernelGGL(nonfinite_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, S(stream), g, (long long)(n / 4), (long long)n, (int*)flag);
  return check_launch("grad_nonfinite");
}

#endif

}  // extern "C"
