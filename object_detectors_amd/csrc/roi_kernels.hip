// RoIAlign / MultiScaleRoIAlign (torchvision.ops.roi_align semantics, call sites tvision/frcnn.py:208-211,
// tvision/roi_heads.py:818) and per-row top-k selection (tvision/rpn.py:215-228, retinanet.py:437-445).
// HBM/latency-bound gather kernels; -ffp-contract=off like the other box kernels.
#include <stdlib.h>

#include "common.h"

using namespace mi355;

namespace {

struct Levels {
  const float* feat[4];
  int h[4], w[4];
  float scale[4];
  int num;
};

__device__ __forceinline__ float bilinear(const float* __restrict__ f, int H, int W, float y, float x) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) {
    yh = yl = H - 1;
    y = (float)yl;
  } else yh = yl + 1;
  if (xl >= W - 1) {
    xh = xl = W - 1;
    x = (float)xl;
  } else xh = xl + 1;
  const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
  return hy * hx * f[yl * W + xl] + hy * lx * f[yl * W + xh] + ly * hx * f[yh * W + xl] + ly * lx * f[yh * W + xh];
}

__device__ __forceinline__ void bilinear_grad(float* __restrict__ g, int H, int W, float y, float x, float v) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return;
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) {
    yh = yl = H - 1;
    y = (float)yl;
  } else yh = yl + 1;
  if (xl >= W - 1) {
    xh = xl = W - 1;
    x = (float)xl;
  } else xh = xl + 1;
  const float ly = y - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
  atomicAdd(g + yl * W + xl, v * hy * hx);
  atomicAdd(g + yl * W + xh, v * hy * lx);
  atomicAdd(g + yh * W + xl, v * ly * hx);
  atomicAdd(g + yh * W + xh, v * ly * lx);
}

// LevelMapper of MultiScaleRoIAlign: k = floor(4 + log2(sqrt(area)/224) + 1e-6) clamped to the pyramid
__device__ __forceinline__ int map_level(const float4 r, int k_min, int k_max) {
  const float s = sqrtf((r.z - r.x) * (r.w - r.y));
  int k = (int)floorf(4.0f + log2f(s / 224.0f) + 1e-6f);
  k = min(max(k, k_min), k_max);
  return k - k_min;
}

// rois [K,5] = (batch, x1,y1,x2,y2); out [K,C,ph,pw].  MULTI: level chosen per RoI, else level 0.
template <bool MULTI, bool BWD>
__global__ __launch_bounds__(256) void roi_align_kernel(Levels L, const float* __restrict__ rois, int K, int C, int ph, int pw, int sampling,
                                                        int aligned, int k_min, int k_max, float* __restrict__ out, const float* __restrict__ gout,
                                                        float* __restrict__ gfeat0, float* __restrict__ gfeat1, float* __restrict__ gfeat2,
                                                        float* __restrict__ gfeat3) {
  const long long total = (long long)K * C * ph * pw;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int px = (int)(i % pw), py = (int)((i / pw) % ph), c = (int)((i / ((long long)pw * ph)) % C), k = (int)(i / ((long long)pw * ph * C));
    const float* r = rois + 5 * (size_t)k;
    const int b = (int)r[0];
    const float4 box = make_float4(r[1], r[2], r[3], r[4]);
    int lv = 0;
    if (MULTI) lv = map_level(box, k_min, k_max);
    int H = L.h[0], W = L.w[0];
    float sc = L.scale[0];
    const float* f = L.feat[0];
    float* gf = gfeat0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (lv == q) {
        H = L.h[q]; W = L.w[q]; sc = L.scale[q]; f = L.feat[q];
        gf = q == 1 ? gfeat1 : (q == 2 ? gfeat2 : gfeat3);
      }
    const float off = aligned ? 0.5f : 0.0f;
    const float x1 = box.x * sc - off, y1 = box.y * sc - off, x2 = box.z * sc - off, y2 = box.w * sc - off;
    float rw = x2 - x1, rh = y2 - y1;
    if (!aligned) {
      rw = fmaxf(rw, 1.0f);
      rh = fmaxf(rh, 1.0f);
    }
    const float bh = rh / (float)ph, bw = rw / (float)pw;
    const int gh = sampling > 0 ? sampling : (int)ceilf(rh / (float)ph), gw = sampling > 0 ? sampling : (int)ceilf(rw / (float)pw);
    const float cnt = fmaxf((float)(gh * gw), 1.0f);
    const size_t plane = ((size_t)b * C + c) * (size_t)H * W;
    if (!BWD) {
      float acc = 0.f;
      for (int iy = 0; iy < gh; ++iy) {
        const float y = y1 + py * bh + ((float)iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float x = x1 + px * bw + ((float)ix + 0.5f) * bw / (float)gw;
          acc += bilinear(f + plane, H, W, y, x);
        }
      }
      out[i] = acc / cnt;
    } else {
      const float g = gout[i] / cnt;
      for (int iy = 0; iy < gh; ++iy) {
        const float y = y1 + py * bh + ((float)iy + 0.5f) * bh / (float)gh;
        for (int ix = 0; ix < gw; ++ix) {
          const float x = x1 + px * bw + ((float)ix + 0.5f) * bw / (float)gw;
          bilinear_grad(gf + plane, H, W, y, x, g);
        }
      }
    }
  }
}

// ---- channels-last form: bf16 NHWC features (the engines' native layout), lanes over channels ---------------------------
// The NCHW kernel above spends its backward in scattered fp32 atomics (64 lanes -> 64 different rows: measured 5.3 ms for
// 2048 RoIs x 256 channels).  With channels innermost a wave's atomics fall into one or two contiguous 256-byte runs (the
// full-rate shape of the memory-side atomic unit) and the forward reads 128-byte runs; the pooled tensor keeps the
// reference's [K, C, ph, pw] order (box_head.fc6 expects it), at the price of 4-byte strided accesses on that (small) side.
struct LevelsCL {
  const bf16_t* feat[4];
  float* grad[4];
  int h[4], w[4], ld[4];
  float scale[4];
};

template <bool BWD>
__global__ __launch_bounds__(256) void roi_align_nhwc_kernel(LevelsCL L, int num_levels, const float* __restrict__ rois, int K, int C, int ph, int pw,
                                                             int sampling, int aligned, int k_min, int k_max, float* __restrict__ out,
                                                             const float* __restrict__ gout) {
  const long long total = (long long)K * ph * pw * C;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int px = (int)(t % pw);
    t /= pw;
    const int py = (int)(t % ph), k = (int)(t / ph);
    const float* r = rois + 5 * (size_t)k;
    const int b = (int)r[0];
    const float4 box = make_float4(r[1], r[2], r[3], r[4]);
    int lv = num_levels > 1 ? map_level(box, k_min, k_max) : 0;
    int H = L.h[0], W = L.w[0], ld = L.ld[0];
    float sc = L.scale[0];
    const bf16_t* f = L.feat[0];
    float* gf = L.grad[0];
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (lv == q) {
        H = L.h[q]; W = L.w[q]; ld = L.ld[q]; sc = L.scale[q]; f = L.feat[q]; gf = L.grad[q];
      }
    const float off = aligned ? 0.5f : 0.0f;
    const float x1 = box.x * sc - off, y1 = box.y * sc - off, x2 = box.z * sc - off, y2 = box.w * sc - off;
    float rw = x2 - x1, rh = y2 - y1;
    if (!aligned) {
      rw = fmaxf(rw, 1.0f);
      rh = fmaxf(rh, 1.0f);
    }
    const float bh = rh / (float)ph, bw = rw / (float)pw;
    const int gh = sampling > 0 ? sampling : (int)ceilf(rh / (float)ph), gw = sampling > 0 ? sampling : (int)ceilf(rw / (float)pw);
    const float cnt = fmaxf((float)(gh * gw), 1.0f);
    const long long oidx = (((long long)k * C + c) * ph + py) * pw + px;
    const size_t img = (size_t)b * H * W;
    float acc = 0.f;
    const float g = BWD ? gout[oidx] / cnt : 0.f;
    for (int iy = 0; iy < gh; ++iy) {
      float y = y1 + py * bh + ((float)iy + 0.5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        float x = x1 + px * bw + ((float)ix + 0.5f) * bw / (float)gw;
        float yy = y;
        if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;
        if (yy <= 0.f) yy = 0.f;
        if (x <= 0.f) x = 0.f;
        int yl = (int)yy, xl = (int)x, yh, xh;
        if (yl >= H - 1) {
          yh = yl = H - 1;
          yy = (float)yl;
        } else yh = yl + 1;
        if (xl >= W - 1) {
          xh = xl = W - 1;
          x = (float)xl;
        } else xh = xl + 1;
        const float ly = yy - yl, lx = x - xl, hy = 1.f - ly, hx = 1.f - lx;
        const size_t p00 = (img + (size_t)yl * W + xl), p01 = (img + (size_t)yl * W + xh), p10 = (img + (size_t)yh * W + xl),
                     p11 = (img + (size_t)yh * W + xh);
        if (!BWD) {
          acc += hy * hx * bf2f(f[p00 * ld + c]) + hy * lx * bf2f(f[p01 * ld + c]) + ly * hx * bf2f(f[p10 * ld + c]) + ly * lx * bf2f(f[p11 * ld + c]);
        } else {
          atomicAdd(gf + p00 * C + c, g * hy * hx);
          atomicAdd(gf + p01 * C + c, g * hy * lx);
          atomicAdd(gf + p10 * C + c, g * ly * hx);
          atomicAdd(gf + p11 * C + c, g * ly * lx);
        }
      }
    }
    if (!BWD) out[oidx] = acc / cnt;
  }
}

// ------------------------------------------------------------------------------------------
// The same RoIAlign in SEPARABLE form for the 7x7 box head (one workgroup per RoI, lanes over channels).  A bilinear sample weights its
// four pixels by (1-ly | ly) x (1-lx | lx), and a sample is dropped when its y OR its x lies outside the map: both factor per axis, so
//   forward   out[i][j]  = sum_y sum_x Ay[y][i] * Ax[x][j] * F[y][x] / count
//   backward  dF[y][x]  += sum_i sum_j Ay[y][i] * Ax[x][j] * g[i][j] / count
// with Ay [rows of the RoI's footprint x 7] the summed y-weights of the samples of bin i (Ax alike), built once per RoI in LDS.  The
// per-sample form touches 7*7*gh*gw*4 pixels per channel (784 at sampling_ratio 2) - one atomic each in the backward -, this one the
// touched rows x columns of the footprint (a 14-pixel RoI: ~225): 2048 RoIs x 256 channels went from 1.28 ms to the time below.
#define RSEP_BINS 7
#define RSEP_CAP 512           // footprint rows / columns held in LDS (= the largest feature map side this form accepts)
struct RsepAxis {
  int lo, n;                   // first touched pixel and extent of the footprint along the axis
};

// weights of one axis: thread `bin` (< 7) walks the samples of its bin.  start / bin_size / grid as in roi_align_nhwc_kernel.
__device__ __forceinline__ void rsep_sample(float s, int size, bool& ok, int& lo, int& hi, float& wl, float& wh) {
  ok = !(s < -1.0f || s > (float)size);
  if (s <= 0.f) s = 0.f;
  lo = (int)s;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    s = (float)lo;
  } else hi = lo + 1;
  wh = s - lo;
  wl = 1.f - wh;
}

template <bool BWD>
__global__ __launch_bounds__(256) void roi_align_sep_kernel(LevelsCL L, int num_levels, const float* __restrict__ rois, int C, int sampling,
                                                            int aligned, int k_min, int k_max, float* __restrict__ out,
                                                            const float* __restrict__ gout) {
  __shared__ float A[2][RSEP_CAP * RSEP_BINS];        // [axis][pixel - lo][bin]
  __shared__ unsigned char touched[2][RSEP_CAP];
  __shared__ int s_lo[2], s_hi[2];
  const int k = blockIdx.x;
  const float* r = rois + 5 * (size_t)k;
  const int b = (int)r[0];
  const float4 box = make_float4(r[1], r[2], r[3], r[4]);
  const int lv = num_levels > 1 ? map_level(box, k_min, k_max) : 0;
  int H = L.h[0], W = L.w[0], ld = L.ld[0];
  float sc = L.scale[0];
  const bf16_t* f = L.feat[0];
  float* gf = L.grad[0];
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (lv == q) {
      H = L.h[q]; W = L.w[q]; ld = L.ld[q]; sc = L.scale[q]; f = L.feat[q]; gf = L.grad[q];
    }
  const float off = aligned ? 0.5f : 0.0f;
  const float x1 = box.x * sc - off, y1 = box.y * sc - off, x2 = box.z * sc - off, y2 = box.w * sc - off;
  float rw = x2 - x1, rh = y2 - y1;
  if (!aligned) {
    rw = fmaxf(rw, 1.0f);
    rh = fmaxf(rh, 1.0f);
  }
  const float bh = rh / (float)RSEP_BINS, bw = rw / (float)RSEP_BINS;
  const int gh = sampling > 0 ? sampling : (int)ceilf(rh / (float)RSEP_BINS), gw = sampling > 0 ? sampling : (int)ceilf(rw / (float)RSEP_BINS);
  const float cnt = fmaxf((float)(gh * gw), 1.0f);
  if (threadIdx.x < 2) {
    s_lo[threadIdx.x] = 0x7fffffff;
    s_hi[threadIdx.x] = -1;
  }
  __syncthreads();
  // threads 0..6: the y axis, 64..70: the x axis (one wave each); pass 1 = extent of the footprint
  const int axis = threadIdx.x >> 6, bin = threadIdx.x & 63;
  const bool worker = axis < 2 && bin < RSEP_BINS;
  const float a0 = axis ? x1 : y1, bs = axis ? bw : bh;
  const int gn = axis ? gw : gh, size = axis ? W : H;
  if (worker) {
    int mn = 0x7fffffff, mx = -1;
    for (int i = 0; i < gn; ++i) {
      const float sp = a0 + bin * bs + ((float)i + 0.5f) * bs / (float)gn;
      bool ok;
      int lo, hi;
      float wl, wh;
      rsep_sample(sp, size, ok, lo, hi, wl, wh);
      if (ok) mn = min(mn, lo), mx = max(mx, hi);
    }
    if (mx >= 0) {
      atomicMin(&s_lo[axis], mn);
      atomicMax(&s_hi[axis], mx);
    }
  }
  __syncthreads();
  const int ylo = s_lo[0], xlo = s_lo[1];
  const int FH = s_hi[0] - ylo + 1, FW = s_hi[1] - xlo + 1;
  const bool empty = s_hi[0] < 0 || s_hi[1] < 0;           // every sample outside the map: zero output, no gradient
  if (!empty) {
    for (int i = threadIdx.x; i < FH * RSEP_BINS; i += 256) A[0][i] = 0.f;
    for (int i = threadIdx.x; i < FW * RSEP_BINS; i += 256) A[1][i] = 0.f;
    for (int i = threadIdx.x; i < FH; i += 256) touched[0][i] = 0;
    for (int i = threadIdx.x; i < FW; i += 256) touched[1][i] = 0;
  }
  __syncthreads();
  if (worker && !empty) {                                    // pass 2: column `bin` of the axis' weight matrix (no other thread writes it)
    const int base = axis ? xlo : ylo;
    for (int i = 0; i < gn; ++i) {
      const float sp = a0 + bin * bs + ((float)i + 0.5f) * bs / (float)gn;
      bool ok;
      int lo, hi;
      float wl, wh;
      rsep_sample(sp, size, ok, lo, hi, wl, wh);
      if (!ok) continue;
      A[axis][(lo - base) * RSEP_BINS + bin] += wl;
      A[axis][(hi - base) * RSEP_BINS + bin] += wh;
      touched[axis][lo - base] = 1;                          // (several bins may set the same flag: same value)
      touched[axis][hi - base] = 1;
    }
  }
  __syncthreads();
  const size_t img = (size_t)b * H * W;
  for (int c = threadIdx.x; c < C; c += 256) {
    float g[RSEP_BINS][RSEP_BINS];
    const size_t obase = ((size_t)k * C + c) * (RSEP_BINS * RSEP_BINS);
    if (BWD) {
#pragma unroll
      for (int i = 0; i < RSEP_BINS; ++i)
#pragma unroll
        for (int j = 0; j < RSEP_BINS; ++j) g[i][j] = gout[obase + i * RSEP_BINS + j] / cnt;
    } else {
#pragma unroll
      for (int i = 0; i < RSEP_BINS; ++i)
#pragma unroll
        for (int j = 0; j < RSEP_BINS; ++j) g[i][j] = 0.f;
    }
    if (!empty) {
      for (int y = 0; y < FH; ++y) {
        if (!touched[0][y]) continue;
        const float* ay = &A[0][y * RSEP_BINS];
        const size_t row = img + (size_t)(ylo + y) * W + xlo;
        float t[RSEP_BINS];
        if (BWD) {
#pragma unroll
          for (int j = 0; j < RSEP_BINS; ++j) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < RSEP_BINS; ++i) a += ay[i] * g[i][j];
            t[j] = a;
          }
          for (int x = 0; x < FW; ++x) {
            if (!touched[1][x]) continue;
            const float* ax = &A[1][x * RSEP_BINS];
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < RSEP_BINS; ++j) v += ax[j] * t[j];
            atomicAdd(gf + (row + x) * C + c, v);
          }
        } else {
#pragma unroll
          for (int j = 0; j < RSEP_BINS; ++j) t[j] = 0.f;
          // untouched columns carry zero weights: no test, so that the loads of four columns are in flight together
          const bf16_t* fr = f + row * ld + c;
          int x = 0;
          for (; x + 3 < FW; x += 4) {
            const float v0 = bf2f(fr[(size_t)x * ld]), v1 = bf2f(fr[(size_t)(x + 1) * ld]), v2 = bf2f(fr[(size_t)(x + 2) * ld]),
                        v3 = bf2f(fr[(size_t)(x + 3) * ld]);
            const float* ax = &A[1][x * RSEP_BINS];
#pragma unroll
            for (int j = 0; j < RSEP_BINS; ++j)
              t[j] += ax[j] * v0 + ax[RSEP_BINS + j] * v1 + ax[2 * RSEP_BINS + j] * v2 + ax[3 * RSEP_BINS + j] * v3;
          }
          for (; x < FW; ++x) {
            const float* ax = &A[1][x * RSEP_BINS];
            const float v = bf2f(fr[(size_t)x * ld]);
#pragma unroll
            for (int j = 0; j < RSEP_BINS; ++j) t[j] += ax[j] * v;
          }
#pragma unroll
          for (int i = 0; i < RSEP_BINS; ++i)
#pragma unroll
            for (int j = 0; j < RSEP_BINS; ++j) g[i][j] += ay[i] * t[j];
        }
      }
    }
    if (!BWD) {
#pragma unroll
      for (int i = 0; i < RSEP_BINS; ++i)
#pragma unroll
        for (int j = 0; j < RSEP_BINS; ++j) out[obase + i * RSEP_BINS + j] = g[i][j] / cnt;
    }
  }
}

// ------------------------------------------------------------------------------------------
// per-row top-k (descending, ties -> lower index first), one workgroup per row:
// 3-pass radix select (11+11+10 bits of the order-preserving key) in LDS histograms, ordered compaction of the
// selected elements, bitonic sort of the <= 16384 survivors.  Elements <= min_value are never selected.
#define TOPK_THREADS 1024
#define TOPK_MAXK 16384
__device__ __forceinline__ void topk_select_digit(const unsigned* __restrict__ hist, int nb, unsigned need, unsigned* s_out /* [3]: digit, need, all */) {
  // one wave: highest bucket b with (count of keys in buckets > b) < need <= (count in buckets >= b)
  const int lane = threadIdx.x;
  unsigned acc = 0;
  int found = -1;
  unsigned need_out = 0;
  for (int base = nb - 64; base >= 0 && found < 0; base -= 64) {
    const unsigned c = hist[base + 63 - lane];              // lane 0 = highest bucket of this group
    unsigned pre = c;                                        // inclusive prefix over lanes (descending buckets)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(pre, o, WAVE);
      if (lane >= o) pre += t;
    }
    const unsigned long long hit = __ballot(acc + pre >= need);
    if (hit) {
      const int l = __ffsll((long long)hit) - 1;
      found = base + 63 - l;
      const unsigned before = __shfl(pre, l, WAVE) - __shfl(c, l, WAVE);
      need_out = need - (acc + before);
    } else {
      acc += __shfl(pre, 63, WAVE);
    }
  }
  if (lane == 0) {
    s_out[0] = found < 0 ? 0u : (unsigned)found;
    s_out[1] = found < 0 ? 0u : need_out;
    s_out[2] = found < 0 ? 1u : 0u;                          // fewer than `need` valid keys: take everything valid
  }
}

// one row by one workgroup: xr [n] -> idx_row / val_row [k], *count_ptr; keys = TOPK_MAXK x 8 bytes of LDS (the histogram aliases the front)
__device__ __forceinline__ void topk_row_ordered(const float* __restrict__ xr, long long n, int k, float min_value, long long* __restrict__ idx_row,
                                                 float* __restrict__ val_row, int* __restrict__ count_ptr, unsigned long long* keys) {
  __shared__ unsigned s_prefix, s_need, s_digit[3];
  __shared__ int wsum[TOPK_THREADS / WAVE];
  __shared__ int s_base, s_tiebase;
  unsigned* hist = (unsigned*)keys;
  const unsigned min_key = f2ord(min_value);
  if (threadIdx.x == 0) {
    s_prefix = 0;
    s_need = (unsigned)k;
  }
  __syncthreads();
  // --- radix select of the k-th largest key among keys > min_key
  const int shifts[3] = {21, 10, 0};
  const int bits[3] = {11, 11, 10};
  unsigned mask_hi = 0;
  for (int pass = 0; pass < 3; ++pass) {
    const int nb = 1 << bits[pass];
    for (int i = threadIdx.x; i < 2048; i += TOPK_THREADS) hist[i] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    for (long long i = threadIdx.x; i < n; i += TOPK_THREADS) {
      const unsigned key = f2ord(xr[i]);
      if (key > min_key && (key & mask_hi) == prefix) atomicAdd(&hist[(key >> shifts[pass]) & (nb - 1)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < WAVE) topk_select_digit(hist, nb, s_need, s_digit);   // (a serial scan of the 2048 buckets by one thread cost 55 us per pass)
    __syncthreads();
    if (threadIdx.x == 0) {
      if (s_digit[2]) {   // fewer than k valid elements: select everything valid
        s_need = 0;
        s_prefix = 0xFFFFFFFFu;   // marker
      } else {
        s_need = s_digit[1];
        s_prefix = prefix | (s_digit[0] << shifts[pass]);
      }
    }
    __syncthreads();
    if (s_prefix == 0xFFFFFFFFu) break;
    mask_hi |= (unsigned)((1 << bits[pass]) - 1) << shifts[pass];
  }
  const bool all_valid = s_prefix == 0xFFFFFFFFu;
  const unsigned thr = all_valid ? min_key : s_prefix;      // select key > thr, plus the first s_need elements == thr
  const unsigned ties_needed = all_valid ? 0u : s_need;
  __syncthreads();
  // --- ordered compaction (index order) of the selected elements into keys[]
  if (threadIdx.x == 0) {
    s_base = 0;
    s_tiebase = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  for (long long i0 = 0; i0 < n; i0 += TOPK_THREADS) {
    const long long i = i0 + threadIdx.x;
    const unsigned key = i < n ? f2ord(xr[i]) : 0u;
    const bool gt = i < n && key > thr && key > min_key;
    const bool tie = i < n && !all_valid && key == thr && key > min_key;
    // ties: prefix count in index order
    const unsigned long long tb = __ballot(tie);
    const int tpre = __popcll(tb & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wid] = __popcll(tb);
    __syncthreads();
    int toff = s_tiebase;
    for (int w = 0; w < wid; ++w) toff += wsum[w];
    const bool take_tie = tie && (unsigned)(toff + tpre) < ties_needed;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < TOPK_THREADS / WAVE; ++w) t += wsum[w];
      s_tiebase += t;
    }
    const bool sel = gt || take_tie;
    const unsigned long long sb = __ballot(sel);
    const int spre = __popcll(sb & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) wsum[wid] = __popcll(sb);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wid; ++w) off += wsum[w];
    if (sel && off + spre < TOPK_MAXK) keys[off + spre] = ((unsigned long long)key << 32) | (0xFFFFFFFFu - (unsigned)i);
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < TOPK_THREADS / WAVE; ++w) t += wsum[w];
      s_base += t;
    }
    __syncthreads();
  }
  const int m = min(s_base, TOPK_MAXK);
  int npad = 64;
  while (npad < m) npad <<= 1;
  for (int i = m + threadIdx.x; i < npad; i += TOPK_THREADS) keys[i] = 0;
  __syncthreads();
  for (int kk = 2; kk <= npad; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npad; i += TOPK_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], c = keys[ixj];
          if (((i & kk) == 0) ? a < c : a > c) {
            keys[i] = c;
            keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < m; i += TOPK_THREADS) {
    const unsigned long long kv = keys[i];
    idx_row[i] = (long long)(0xFFFFFFFFu - (unsigned)(kv & 0xFFFFFFFFull));
    if (val_row) val_row[i] = ord2f((unsigned)(kv >> 32));
  }
  if (threadIdx.x == 0) *count_ptr = m;
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(const float* __restrict__ x, long long n, long long row_stride, int k,
                                                            float min_value, long long* __restrict__ idx_out, float* __restrict__ val_out,
                                                            int* __restrict__ count_out, const unsigned* __restrict__ run_flags = nullptr,
                                                            int flag_stride = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // TOPK_MAXK keys
  if (run_flags && run_flags[(size_t)blockIdx.x * flag_stride] == 0) return;   // fallback launch of the multi-workgroup form: nothing to redo
  const int row = blockIdx.x;
  topk_row_ordered(x + (size_t)row * row_stride, n, k, min_value, idx_out + (size_t)row * k, val_out ? val_out + (size_t)row * k : nullptr,
                   count_out + row, keys);
}


// ------------------------------------------------------------------------------------------
// The same selection for LONG rows (RetinaNet post-processing flattens HWA x K scores per level: 0.1 - 100 M elements) with many
// workgroups per row.  The one-workgroup form above walks the row four times with 1024 threads and synchronises seven times per 1024
// elements in its ordered compaction: 0.9 ms at n = 360 000.  Here:
//   3 histogram launches (11 + 11 + 10 bits; each workgroup bins its slice in LDS and adds the non-empty bins to the row's global
//     histogram; from the second launch on every workgroup first derives the digit chosen by the previous level from that histogram),
//   1 collect launch (keys above the exact 32-bit threshold and the keys equal to it are appended UNORDERED, one reservation per
//     workgroup: the final sort orders by (key desc, index asc) anyway, which is the required output order),
//   1 finish launch (one workgroup per row: bitonic sort of the <= 16384 candidates in LDS, first k out).
// If more elements equal the threshold than the candidate buffer holds (degenerate rows of identical scores) a flag is raised and the
// one-workgroup form, launched behind with that flag, redoes the row exactly.
struct TopkState {           // per row, zeroed by the launch function
  unsigned hist[3][2048];
  unsigned n_gt, n_tie, overflow, pad;
};
#define TOPK_SLICES 64

// state after `levels` histogram levels: prefix bits, remaining need, all_valid
__device__ __forceinline__ void topk_replay(const TopkState* st, int levels, unsigned k, unsigned* s_sel /* LDS [4] */) {
  const int shifts[3] = {21, 10, 0};
  const int bits[3] = {11, 11, 10};
  unsigned prefix = 0, need = k, all = 0, stop = 0;
  __shared__ unsigned s_tmp[3];
  for (int l = 0; l < levels && !all && !stop; ++l) {
    if (threadIdx.x < WAVE) topk_select_digit(st->hist[l], 1 << bits[l], need, s_tmp);
    __syncthreads();
    all = s_tmp[2];
    if (!all) {
      prefix |= s_tmp[0] << shifts[l];
      need = s_tmp[1];
      // Early stop: the keys above this bucket (k - need of them) plus the whole bucket fit the candidate buffer - the final sort picks the
      // k best of them exactly, so the remaining histogram levels (a full pass over the row each) are skipped.  With a score threshold
      // (RetinaNet post-processing) the first level usually decides: two passes over a 108 M-element row instead of four.
      if ((k - need) + st->hist[l][s_tmp[0]] <= (unsigned)TOPK_MAXK - 64u) stop = (unsigned)l + 1u;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    s_sel[0] = prefix;
    s_sel[1] = need;
    s_sel[2] = all;
    s_sel[3] = stop;      // > 0: `prefix` holds only the digits of levels < stop; every key >= prefix is a candidate
  }
  __syncthreads();
}

template <int LEVEL>
__global__ __launch_bounds__(1024) void topk_hist_kernel(const float* __restrict__ x, long long n, long long row_stride, int k, float min_value,
                                                          TopkState* __restrict__ states) {
  __shared__ unsigned hist[2048];
  __shared__ unsigned s_sel[4];
  const int shifts[3] = {21, 10, 0};
  const int bits[3] = {11, 11, 10};
  const int row = blockIdx.y;
  TopkState* st = states + row;
  const float* xr = x + (size_t)row * row_stride;
  const unsigned min_key = f2ord(min_value);
  for (int i = threadIdx.x; i < 2048; i += 1024) hist[i] = 0;
  topk_replay(st, LEVEL, (unsigned)k, s_sel);               // (ends with a barrier: hist zeroed too)
  if (s_sel[2] || s_sel[3]) return;                          // fewer than k valid keys in the row, or the candidates already fit: nothing to refine
  const unsigned prefix = s_sel[0];
  unsigned mask_hi = 0;
  for (int l = 0; l < LEVEL; ++l) mask_hi |= (unsigned)((1 << bits[l]) - 1) << shifts[l];
  const long long per = (n + gridDim.x - 1) / gridDim.x;
  const long long lo = per * blockIdx.x, hi = min(n, lo + per);
  const int nb = 1 << bits[LEVEL];
  auto count = [&](float v) {
    const unsigned key = f2ord(v);
    if (key > min_key && (key & mask_hi) == prefix) atomicAdd(&hist[(key >> shifts[LEVEL]) & (nb - 1)], 1u);
  };
  // 16-byte loads over the aligned body of this workgroup's range (the scalar form read 108 M-element rows at 2.3 TB/s)
  if (lo < hi) {
    const float* p0 = xr + lo;
    const long long cnt = hi - lo;
    const long long head = min(cnt, (long long)(((16 - ((unsigned long long)p0 & 15)) & 15) >> 2));
    for (long long i = threadIdx.x; i < head; i += 1024) count(p0[i]);
    const long long nvec = (cnt - head) >> 2;
    const float4* pv = (const float4*)(p0 + head);
    long long v = threadIdx.x;
    for (; v + 1024 < nvec; v += 2048) {                      // two independent loads in flight
      const float4 a = pv[v], b = pv[v + 1024];
      count(a.x); count(a.y); count(a.z); count(a.w);
      count(b.x); count(b.y); count(b.z); count(b.w);
    }
    for (; v < nvec; v += 1024) {
      const float4 a = pv[v];
      count(a.x); count(a.y); count(a.z); count(a.w);
    }
    for (long long i = head + 4 * nvec + threadIdx.x; i < cnt; i += 1024) count(p0[i]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += 1024)
    if (hist[i]) atomicAdd(&st->hist[LEVEL][i], hist[i]);
}

__global__ __launch_bounds__(1024) void topk_collect_kernel(const float* __restrict__ x, long long n, long long row_stride, int k, float min_value,
                                                             TopkState* __restrict__ states, unsigned long long* __restrict__ cand) {
  __shared__ unsigned s_sel[4];
  __shared__ unsigned s_cnt, s_base;
  __shared__ unsigned long long s_keys[4096];                // this workgroup's hits before they get their place in the row's list
  const int row = blockIdx.y;
  TopkState* st = states + row;
  const float* xr = x + (size_t)row * row_stride;
  unsigned long long* out = cand + (size_t)row * TOPK_MAXK;
  const unsigned min_key = f2ord(min_value);
  topk_replay(st, 3, (unsigned)k, s_sel);
  const bool all = s_sel[2] != 0;
  const bool coarse = s_sel[3] != 0;                         // early stop: keys >= the coarse prefix are all candidates
  const unsigned thr = all ? min_key : (coarse ? (s_sel[0] ? s_sel[0] - 1u : 0u) : s_sel[0]);   // take key > thr, and (exact form) keys == thr as ties
  const bool take_zero = coarse && s_sel[0] == 0;           // prefix 0: key 0 itself is a candidate too (key > thr cannot express it)
  const long long per = (n + gridDim.x - 1) / gridDim.x;
  const long long lo = per * blockIdx.x, hi = min(n, lo + per);
  // 4096 elements per round (the LDS list cannot overflow): one 16-byte load per thread over the aligned body of the range
  const float* p0 = xr + lo;
  const long long cnt_all = hi > lo ? hi - lo : 0;
  const long long head = min(cnt_all, (long long)(((16 - ((unsigned long long)p0 & 15)) & 15) >> 2));
  const long long nvec = (cnt_all - head) >> 2;
  const float4* pv = (const float4*)(p0 + head);
  const long long tail0 = head + 4 * nvec;
  // rounds: [0] = the unaligned head + tail (< 8 elements), then the vector body
  const long long rounds = cnt_all ? 1 + (nvec + 1023) / 1024 : 0;
  for (long long r = 0; r < rounds; ++r) {
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    auto take = [&](long long i, float v) {
      const unsigned key = f2ord(v);
      const bool gt = (key > thr || take_zero) && key > min_key;
      const bool tie = !all && !coarse && key == thr && key > min_key;
      if (gt || tie) {
        const unsigned p = atomicAdd(&s_cnt, 1u);
        s_keys[p] = ((unsigned long long)key << 32) | (0xFFFFFFFFu - (unsigned)(lo + i));
      }
    };
    if (r == 0) {
      if ((long long)threadIdx.x < head) take(threadIdx.x, p0[threadIdx.x]);
      const long long t = tail0 + threadIdx.x;
      if (t < cnt_all && (long long)threadIdx.x < 4) take(t, p0[t]);
    } else {
      const long long v = (r - 1) * 1024 + threadIdx.x;
      if (v < nvec) {
        const float4 a = pv[v];
        const long long i = head + 4 * v;
        take(i, a.x); take(i + 1, a.y); take(i + 2, a.z); take(i + 3, a.w);
      }
    }
    __syncthreads();
    const unsigned cnt = s_cnt;
    if (cnt) {
      if (threadIdx.x == 0) s_base = atomicAdd(&st->n_gt, cnt);          // gt and ties share one list: the sort separates them
      __syncthreads();
      const unsigned base = s_base;
      for (unsigned j = threadIdx.x; j < cnt; j += 1024) {
        if (base + j < TOPK_MAXK) out[base + j] = s_keys[j];
        else st->overflow = 1u;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_finish_kernel(int k, TopkState* __restrict__ states, const unsigned long long* __restrict__ cand,
                                                                    long long* __restrict__ idx_out, float* __restrict__ val_out,
                                                                    int* __restrict__ count_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const int row = blockIdx.x;
  const TopkState* st = states + row;
  if (st->overflow) return;                                   // the one-workgroup launch behind this one redoes the row
  const int m = (int)min(st->n_gt, (unsigned)TOPK_MAXK);
  int npad = 64;
  while (npad < m) npad <<= 1;
  const unsigned long long* in = cand + (size_t)row * TOPK_MAXK;
  for (int i = threadIdx.x; i < npad; i += TOPK_THREADS) keys[i] = i < m ? in[i] : 0ull;
  __syncthreads();
  for (int kk = 2; kk <= npad; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npad; i += TOPK_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], c = keys[ixj];
          if (((i & kk) == 0) ? a < c : a > c) {
            keys[i] = c;
            keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  const int take = min(m, k);
  for (int i = threadIdx.x; i < take; i += TOPK_THREADS) {
    const unsigned long long kv = keys[i];
    idx_out[(size_t)row * k + i] = (long long)(0xFFFFFFFFu - (unsigned)(kv & 0xFFFFFFFFull));
    if (val_out) val_out[(size_t)row * k + i] = ord2f((unsigned)(kv >> 32));
  }
  if (threadIdx.x == 0) count_out[row] = take;
}

// ------------------------------------------------------------------------------------------
// Short rows cut into SEGMENTS (the pyramid levels of RegionProposalNetwork._get_top_n_idx, rpn.py:215-228: top-k per level of every
// image): one workgroup per (row, segment), all of them in ONE launch - the per-level calls of the one-workgroup form ran 4 workgroups
// at a time, five times in a row (0.9 ms of a 1.4 ms proposal filter).  Per workgroup: histogram levels with 16-byte loads until the
// candidates (keys above the chosen digit + its whole bucket) fit a short sort, unordered append into LDS, bitonic sort by (key desc,
// index asc), first k out - the same selection as the ordered form; rows of mostly identical keys fall back to it in place.
#define TOPK_SEG_MAX 8
struct TopkSegs {
  long long start[TOPK_SEG_MAX];     // first column of the segment in a row of x
  int n[TOPK_SEG_MAX], k[TOPK_SEG_MAX];
  long long* idx[TOPK_SEG_MAX];      // [rows, k]
  float* val[TOPK_SEG_MAX];          // [rows, k] or null
  int* cnt[TOPK_SEG_MAX];            // [rows]
};

template <class F>
__device__ __forceinline__ void topk_row_scan(const float* __restrict__ p0, int n, F f) {
  const int head = min(n, (int)(((16 - ((unsigned long long)p0 & 15)) & 15) >> 2));
  const int nvec = (n - head) >> 2;
  const float4* pv = (const float4*)(p0 + head);
  if ((int)threadIdx.x < head) f((int)threadIdx.x, p0[threadIdx.x]);
  for (int v = threadIdx.x; v < nvec; v += TOPK_THREADS) {
    const float4 a = pv[v];
    const int i = head + 4 * v;
    f(i, a.x); f(i + 1, a.y); f(i + 2, a.z); f(i + 3, a.w);
  }
  const int t = head + 4 * nvec + (int)threadIdx.x;
  if (t < n) f(t, p0[t]);
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_seg_kernel(const float* __restrict__ x, long long row_stride, TopkSegs G, float min_value) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // TOPK_MAXK keys
  __shared__ unsigned hist[2048];
  __shared__ unsigned s_sel[3], s_cnt;
  const int row = blockIdx.x, sg = blockIdx.y;
  const int n = G.n[sg], k = G.k[sg];
  const float* xr = x + (size_t)row * row_stride + G.start[sg];
  long long* idx_row = G.idx[sg] + (size_t)row * k;
  float* val_row = G.val[sg] ? G.val[sg] + (size_t)row * k : nullptr;
  const unsigned min_key = f2ord(min_value);
  const int shifts[3] = {21, 10, 0};
  const int bits[3] = {11, 11, 10};
  const unsigned target = (unsigned)min(TOPK_MAXK, max(4096, 2 * k));        // candidates worth sorting rather than another pass over the row
  unsigned prefix = 0, mask_hi = 0, need = (unsigned)k, cand = 0;
  bool all = false, done = false;
  for (int l = 0; l < 3 && !done; ++l) {
    for (int i = threadIdx.x; i < 2048; i += TOPK_THREADS) hist[i] = 0;
    __syncthreads();
    const int nb = 1 << bits[l];
    topk_row_scan(xr, n, [&](int, float v) {
      const unsigned key = f2ord(v);
      if (key > min_key && (key & mask_hi) == prefix) atomicAdd(&hist[(key >> shifts[l]) & (nb - 1)], 1u);
    });
    __syncthreads();
    if (threadIdx.x < WAVE) topk_select_digit(hist, nb, need, s_sel);
    __syncthreads();
    if (s_sel[2]) {                       // fewer than `need` valid keys: everything valid is selected
      all = done = true;
    } else {
      cand = ((unsigned)k - s_sel[1]) + hist[s_sel[0]];
      prefix |= s_sel[0] << shifts[l];
      mask_hi |= (unsigned)(nb - 1) << shifts[l];
      need = s_sel[1];
      done = cand <= target;
    }
    __syncthreads();                      // hist is zeroed again by the next level
  }
  if (!all && cand > (unsigned)TOPK_MAXK) {   // exact threshold, and more keys equal to it than the list holds: the ordered form takes the first ones
    topk_row_ordered(xr, n, k, min_value, idx_row, val_row, G.cnt[sg] + row, keys);
    return;
  }
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  topk_row_scan(xr, n, [&](int i, float v) {
    const unsigned key = f2ord(v);
    if (key > min_key && (all || (key & mask_hi) >= prefix)) {
      const unsigned p = atomicAdd(&s_cnt, 1u);
      if (p < (unsigned)TOPK_MAXK) keys[p] = ((unsigned long long)key << 32) | (0xFFFFFFFFu - (unsigned)i);
    }
  });
  __syncthreads();
  const int m = (int)min(s_cnt, (unsigned)TOPK_MAXK);
  int npad = 64;
  while (npad < m) npad <<= 1;
  for (int i = m + threadIdx.x; i < npad; i += TOPK_THREADS) keys[i] = 0;
  __syncthreads();
  for (int kk = 2; kk <= npad; kk <<= 1)
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npad; i += TOPK_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], c = keys[ixj];
          if (((i & kk) == 0) ? a < c : a > c) {
            keys[i] = c;
            keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  const int take = min(m, k);
  for (int i = threadIdx.x; i < take; i += TOPK_THREADS) {
    const unsigned long long kv = keys[i];
    idx_row[i] = (long long)(0xFFFFFFFFu - (unsigned)(kv & 0xFFFFFFFFull));
    if (val_row) val_row[i] = ord2f((unsigned)(kv >> 32));
  }
  if (threadIdx.x == 0) G.cnt[sg][row] = take;
}

}  // namespace

extern "C" {

int mi355det_roi_align(const float* const* feats, const int32_t* hs, const int32_t* ws, const float* scales, int32_t num_levels,
                       const float* rois, int32_t num_rois, int32_t channels, int32_t pooled_h, int32_t pooled_w, int32_t sampling_ratio,
                       int aligned, int32_t k_min, int32_t k_max, float* out, const float* grad_out, float* const* grad_feats, void* stream) {
  if (num_levels < 1 || num_levels > 4 || num_rois < 0 || channels <= 0 || pooled_h <= 0 || pooled_w <= 0)
    return fail(MI355DET_EINVAL, "%s: bad arguments", "roi_align");
  if (num_rois == 0) return 0;
  Levels L{};
  L.num = num_levels;
  float* gf[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int q = 0; q < num_levels; ++q) {
    L.feat[q] = feats ? feats[q] : nullptr;
    L.h[q] = hs[q];
    L.w[q] = ws[q];
    L.scale[q] = scales[q];
    if (grad_feats) gf[q] = grad_feats[q];
  }
  const long long total = (long long)num_rois * channels * pooled_h * pooled_w;
  const int blocks = (int)min((long long)256 * 16, (total + 255) / 256);
  const bool multi = num_levels > 1, bwd = grad_out != nullptr;
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, S(stream), L, rois, num_rois, channels, pooled_h, pooled_w, sampling_ratio, aligned, k_min,
                       k_max, out, grad_out, gf[0], gf[1], gf[2], gf[3]);
  };
  if (multi && bwd) launch(roi_align_kernel<true, true>);
  else if (multi) launch(roi_align_kernel<true, false>);
  else if (bwd) launch(roi_align_kernel<false, true>);
  else launch(roi_align_kernel<false, false>);
  return check_launch("roi_align");
}

int mi355det_roi_align_nhwc(const void* const* feats, const int32_t* hs, const int32_t* ws, const int32_t* lds, const float* scales, int32_t num_levels,
                            const float* rois, int32_t num_rois, int32_t channels, int32_t pooled_h, int32_t pooled_w, int32_t sampling_ratio,
                            int aligned, int32_t k_min, int32_t k_max, float* out, const float* grad_out, float* const* grad_feats, void* stream) {
  if (num_levels < 1 || num_levels > 4 || num_rois < 0 || channels <= 0 || pooled_h <= 0 || pooled_w <= 0)
    return fail(MI355DET_EINVAL, "%s: bad arguments", "roi_align_nhwc");
  if ((grad_out != nullptr) != (grad_feats != nullptr)) return fail(MI355DET_EINVAL, "%s: grad_out and grad_feats go together", "roi_align_nhwc");
  if (num_rois == 0) return 0;
  LevelsCL L{};
  for (int q = 0; q < num_levels; ++q) {
    L.feat[q] = feats ? (const bf16_t*)feats[q] : nullptr;
    L.grad[q] = grad_feats ? grad_feats[q] : nullptr;
    L.h[q] = hs[q];
    L.w[q] = ws[q];
    L.ld[q] = lds ? lds[q] : channels;
    L.scale[q] = scales[q];
  }
  bool separable = pooled_h == RSEP_BINS && pooled_w == RSEP_BINS;
  for (int q = 0; q < num_levels; ++q) separable = separable && hs[q] <= RSEP_CAP && ws[q] <= RSEP_CAP;
  if (separable) {          // the 7x7 box head: one workgroup per RoI, footprint weights in LDS
    if (grad_out)
      hipLaunchKernelGGL(roi_align_sep_kernel<true>, dim3(num_rois), dim3(256), 0, S(stream), L, num_levels, rois, channels, sampling_ratio, aligned,
                         k_min, k_max, out, grad_out);
    else
      hipLaunchKernelGGL(roi_align_sep_kernel<false>, dim3(num_rois), dim3(256), 0, S(stream), L, num_levels, rois, channels, sampling_ratio, aligned,
                         k_min, k_max, out, grad_out);
    return check_launch("roi_align_nhwc");
  }
  const long long total = (long long)num_rois * channels * pooled_h * pooled_w;
  const int blocks = (int)min((long long)256 * 32, (total + 255) / 256);
  if (grad_out)
    hipLaunchKernelGGL(roi_align_nhwc_kernel<true>, dim3(blocks), dim3(256), 0, S(stream), L, num_levels, rois, num_rois, channels, pooled_h, pooled_w,
                       sampling_ratio, aligned, k_min, k_max, out, grad_out);
  else
    hipLaunchKernelGGL(roi_align_nhwc_kernel<false>, dim3(blocks), dim3(256), 0, S(stream), L, num_levels, rois, num_rois, channels, pooled_h, pooled_w,
                       sampling_ratio, aligned, k_min, k_max, out, grad_out);
  return check_launch("roi_align_nhwc");
}

int mi355det_topk(const float* x, int32_t rows, int64_t n, int64_t row_stride, int32_t k, float min_value, int64_t* idx_out, float* val_out,
                  int32_t* count_out, void* stream) {
  if (rows <= 0 || n <= 0 || k <= 0 || k > TOPK_MAXK || n >= (1ll << 32)) return fail(MI355DET_EINVAL, "%s: need 1 <= k <= 16384 and n < 2^32", "topk");
  const int lds = TOPK_MAXK * 8;
  static DeviceOnce attr_done;
  attr_done.once([&] {
    (void)hipFuncSetAttribute((const void*)topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  hipLaunchKernelGGL(topk_kernel, dim3(rows), dim3(TOPK_THREADS), lds, S(stream), x, (long long)n, (long long)row_stride, k, min_value,
                     (long long*)idx_out, val_out, count_out);
  return check_launch("topk");
}

size_t mi355det_topk_workspace(int32_t rows) { return (size_t)(rows > 0 ? rows : 0) * (sizeof(TopkState) + (size_t)TOPK_MAXK * 8); }

int mi355det_topk_ws(const float* x, int32_t rows, int64_t n, int64_t row_stride, int32_t k, float min_value, int64_t* idx_out, float* val_out,
                     int32_t* count_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (rows <= 0 || n <= 0 || k <= 0 || k > TOPK_MAXK || n >= (1ll << 32)) return fail(MI355DET_EINVAL, "%s: need 1 <= k <= 16384 and n < 2^32", "topk");
  if (n < 65536) return mi355det_topk(x, rows, n, row_stride, k, min_value, idx_out, val_out, count_out, stream);     // short rows: one workgroup each
  if (workspace_bytes < mi355det_topk_workspace(rows)) return fail(MI355DET_EWORKSPACE, "%s: workspace too small", "topk");
  TopkState* states = (TopkState*)workspace;
  unsigned long long* cand = (unsigned long long*)((char*)workspace + (size_t)rows * sizeof(TopkState));
  hipStream_t st = S(stream);
  (void)hipMemsetAsync(states, 0, (size_t)rows * sizeof(TopkState), st);
  const int slices = (int)min((long long)TOPK_SLICES * 4, max((long long)1, (long long)(n / 16384)));
  const dim3 grid(slices, rows);
  hipLaunchKernelGGL(topk_hist_kernel<0>, grid, dim3(1024), 0, st, x, (long long)n, (long long)row_stride, k, min_value, states);
  hipLaunchKernelGGL(topk_hist_kernel<1>, grid, dim3(1024), 0, st, x, (long long)n, (long long)row_stride, k, min_value, states);
  hipLaunchKernelGGL(topk_hist_kernel<2>, grid, dim3(1024), 0, st, x, (long long)n, (long long)row_stride, k, min_value, states);
  hipLaunchKernelGGL(topk_collect_kernel, grid, dim3(1024), 0, st, x, (long long)n, (long long)row_stride, k, min_value, states, cand);
  const int lds = TOPK_MAXK * 8;
  static DeviceOnce attr_done;
  attr_done.once([&] {
    (void)hipFuncSetAttribute((const void*)topk_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  hipLaunchKernelGGL(topk_finish_kernel, dim3(rows), dim3(TOPK_THREADS), lds, st, k, states, cand, (long long*)idx_out, val_out, count_out);
  // rows whose threshold value repeats more often than the candidate list holds: exact redo by the one-workgroup form (exits at once otherwise)
  hipLaunchKernelGGL(topk_kernel, dim3(rows), dim3(TOPK_THREADS), lds, st, x, (long long)n, (long long)row_stride, k, min_value, (long long*)idx_out,
                     val_out, count_out, (const unsigned*)&states[0].overflow, (int)(sizeof(TopkState) / sizeof(unsigned)));
  return check_launch("topk_ws");
}

int mi355det_topk_segments(const float* x, int32_t rows, int64_t row_stride, int32_t nseg, const int64_t* seg_start, const int64_t* seg_n,
                           const int32_t* seg_k, float min_value, int64_t* const* idx_out, float* const* val_out, int32_t* const* count_out,
                           void* workspace, size_t workspace_bytes, void* stream) {
  if (rows <= 0 || nseg <= 0 || nseg > TOPK_SEG_MAX || !seg_start || !seg_n || !seg_k || !idx_out || !count_out)
    return fail(MI355DET_EINVAL, "%s: need rows > 0 and 1..8 segments", "topk_segments");
  TopkSegs G{};
  int ns = 0;
  for (int s = 0; s < nseg; ++s) {
    if (seg_n[s] <= 0 || seg_k[s] <= 0 || seg_k[s] > TOPK_MAXK || seg_k[s] > seg_n[s] || seg_n[s] >= (1ll << 31) || !idx_out[s] || !count_out[s])
      return fail(MI355DET_EINVAL, "%s: need 1 <= k <= min(n, 16384) per segment", "topk_segments");
    if (seg_n[s] >= 65536) {             // long segment: many workgroups per row
      if (int e = mi355det_topk_ws(x + seg_start[s], rows, seg_n[s], row_stride, seg_k[s], min_value, idx_out[s], val_out ? val_out[s] : nullptr,
                                   count_out[s], workspace, workspace_bytes, stream))
        return e;
      continue;
    }
    G.start[ns] = seg_start[s];
    G.n[ns] = (int)seg_n[s];
    G.k[ns] = seg_k[s];
    G.idx[ns] = (long long*)idx_out[s];
    G.val[ns] = val_out ? val_out[s] : nullptr;
    G.cnt[ns] = count_out[s];
    ++ns;
  }
  if (ns == 0) return 0;
  const int lds = TOPK_MAXK * 8;
  static DeviceOnce attr_done;
  attr_done.once([&] { (void)hipFuncSetAttribute((const void*)topk_seg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
  hipLaunchKernelGGL(topk_seg_kernel, dim3(rows, ns), dim3(TOPK_THREADS), lds, S(stream), x, (long long)row_stride, G, min_value);
  return check_launch("topk_segments");
}

}  // extern "C"
