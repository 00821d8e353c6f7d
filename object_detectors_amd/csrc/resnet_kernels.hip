// HBM-bound kernels of the ResNet-FPN / RetinaNet training path (torchvision_models/utilities/resnet.py:87-143,230-240,
// tvision/backbone_utils.py:33-63, tvision/retinanet.py:150-223): everything around the MFMA convolutions that is not a
// convolution.  NHWC bf16 activations, 16-byte accesses, grid-stride loops.
//
//   im2col_nchw        7x7/2 stem: NCHW fp32 image (+ GeneralizedRCNNTransform normalisation, transform.py:120-124)
//                      -> im2col rows [n*ho*wo][kpad] bf16, consumed by the MFMA GEMM as a 1x1 convolution
//   maxpool3x3s2       nn.MaxPool2d(3, 2, 1) (resnet.py:176)
//   relu_affine_bwd    backward of  a = relu(conv * scale + shift [+ identity])  with a frozen affine (FrozenBatchNorm2d):
//                      gm = (g1 [+ g2]) * [a > 0]  (gradient of the pre-activation sum = gradient of the identity branch),
//                      dz = gm * scale             (gradient of the raw convolution output)
//   upsample_nearest   FPN top-down path: out = lateral + nearest(top -> lateral size) and its adjoint
//   cast_rows          fp32 [N, rows, C] slice of the level-concatenated head gradient -> bf16 NHWC level buffer
#include "common.h"

using namespace mi355;

namespace {

struct bf8 {
  float v[8];
};
__device__ __forceinline__ bf8 unpack8(const uint4 u) {
  bf8 r;
  r.v[0] = bf2f((bf16_t)(u.x & 0xFFFF)); r.v[1] = bf2f((bf16_t)(u.x >> 16));
  r.v[2] = bf2f((bf16_t)(u.y & 0xFFFF)); r.v[3] = bf2f((bf16_t)(u.y >> 16));
  r.v[4] = bf2f((bf16_t)(u.z & 0xFFFF)); r.v[5] = bf2f((bf16_t)(u.z >> 16));
  r.v[6] = bf2f((bf16_t)(u.w & 0xFFFF)); r.v[7] = bf2f((bf16_t)(u.w >> 16));
  return r;
}
__device__ __forceinline__ uint4 pack8(const bf8& r) {
  uint4 u;
  u.x = (unsigned)f2bf(r.v[0]) | ((unsigned)f2bf(r.v[1]) << 16);
  u.y = (unsigned)f2bf(r.v[2]) | ((unsigned)f2bf(r.v[3]) << 16);
  u.z = (unsigned)f2bf(r.v[4]) | ((unsigned)f2bf(r.v[5]) << 16);
  u.w = (unsigned)f2bf(r.v[6]) | ((unsigned)f2bf(r.v[7]) << 16);
  return u;
}

inline int grid_for(long long total) { return (int)min((long long)256 * 16, max(1ll, (total + 255) / 256)); }

// one thread per (output pixel, 8-wide k chunk): k = (kh*ks + kw)*c + ch.  The k -> (kh, kw, ch) decode is a per-block LDS table
// and the pixel decode is 32-bit: with runtime divisors every element cost three integer divisions and the kernel was ALU
// bound (0.84 ms for 16 x 800 x 800; the writes alone are 0.41 GB).
__global__ __launch_bounds__(256) void im2col_nchw_kernel(const float* __restrict__ img, const float* __restrict__ mean, const float* __restrict__ istd,
                                                           bf16_t* __restrict__ out, int n, int c, int h, int w, int ho, int wo, int ks, int stride,
                                                           int pad, int kpad) {
  extern __shared__ int ktab[];      // [kpad]: kh | kw << 8 | ch << 16, or -1 beyond ks*ks*c
  const int kvalid = ks * ks * c;
  for (int k = threadIdx.x; k < kpad; k += 256) {
    int v = -1;
    if (k < kvalid) {
      const int ch = k % c, t = k / c, kw = t % ks, kh = t / ks;
      v = kh | (kw << 8) | (ch << 16);
    }
    ktab[k] = v;
  }
  __syncthreads();
  const int chunks = kpad / 8;
  const unsigned total = (unsigned)n * ho * wo * chunks;            // < 2^31 checked on the host
  const unsigned hw = (unsigned)ho * wo;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned q = i % chunks, p = i / chunks;
    const unsigned b = p / hw, r = p - b * hw;
    const int oy = (int)(r / wo), ox = (int)(r - (r / wo) * wo);
    const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
    const float* ib = img + (size_t)b * c * h * w;
    bf8 rr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = ktab[q * 8 + j];
      float v = 0.f;
      if (e >= 0) {
        const int kh = e & 0xFF, kw = (e >> 8) & 0xFF, ch = e >> 16;
        const int iy = iy0 + kh, ix = ix0 + kw;
        if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) {
          v = ib[((size_t)ch * h + iy) * w + ix];
          if (mean) v = (v - mean[ch]) * istd[ch];
        }
      }
      rr.v[j] = v;
    }
    *(uint4*)(out + (size_t)p * kpad + q * 8) = pack8(rr);
  }
}

__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const bf16_t* __restrict__ x, int x_ld, int n, int h, int w, int c, int ho, int wo,
                                                            bf16_t* __restrict__ out, int out_ld) {
  const int groups = c >> 3;
  const long long total = (long long)n * ho * wo * groups;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int g = (int)(i % groups);
    const long long p = i / groups;
    const int ox = (int)(p % wo), oy = (int)((p / wo) % ho), b = (int)(p / ((long long)wo * ho));
    bf8 m;
#pragma unroll
    for (int j = 0; j < 8; ++j) m.v[j] = -__builtin_inff();
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = 2 * oy - 1 + kh;
      if (iy < 0 || iy >= h) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = 2 * ox - 1 + kw;
        if (ix < 0 || ix >= w) continue;
        const bf8 v = unpack8(*(const uint4*)(x + ((long long)(b * h + iy) * w + ix) * x_ld + g * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) m.v[j] = fmaxf(m.v[j], v.v[j]);
      }
    }
    *(uint4*)(out + p * out_ld + g * 8) = pack8(m);
  }
}

// Backward of nn.MaxPool2d(3, 2, 1) as a GATHER (deterministic, no atomics): an input element receives the gradient of every output window
// whose maximum it is - the FIRST maximum in the window's (kh, kw) scan, torch's index rule (the forward keeps `val > max`, so ties stay with
// the earlier position).  An input pixel lies in at most 2 x 2 windows; each window's arg-max is re-derived from the forward input.
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const bf16_t* __restrict__ x, int x_ld, const bf16_t* __restrict__ g, int g_ld, int n, int h,
                                                                int w, int c, int ho, int wo, bf16_t* __restrict__ dx, int dx_ld) {
  const int groups = c >> 3;
  const long long total = (long long)n * h * w * groups;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int gq = (int)(i % groups);
    const long long p = i / groups;
    const int ix = (int)(p % w), iy = (int)((p / w) % h), b = (int)(p / ((long long)w * h));
    const bf8 me = unpack8(*(const uint4*)(x + p * x_ld + gq * 8));
    bf8 acc;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc.v[j] = 0.f;
    const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1, ox0 = ix >> 1, ox1 = (ix + 1) >> 1;      // windows with 2*o - 1 <= i <= 2*o + 1
    for (int oy = oy0; oy <= oy1; ++oy) {
      if (oy >= ho) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        if (ox >= wo) continue;
        const int my = iy - (2 * oy - 1), mx = ix - (2 * ox - 1);                          // this element's position in the window
        bool win[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) win[j] = true;
        for (int kh = 0; kh < 3; ++kh) {
          const int yy = 2 * oy - 1 + kh;
          if (yy < 0 || yy >= h) continue;
          for (int kw = 0; kw < 3; ++kw) {
            const int xx = 2 * ox - 1 + kw;
            if (xx < 0 || xx >= w || (kh == my && kw == mx)) continue;
            const bf8 v = unpack8(*(const uint4*)(x + ((long long)(b * h + yy) * w + xx) * x_ld + gq * 8));
            const bool earlier = kh < my || (kh == my && kw < mx);
#pragma unroll
            for (int j = 0; j < 8; ++j) win[j] = win[j] && (earlier ? v.v[j] < me.v[j] : v.v[j] <= me.v[j]) ;
          }
        }
        const bf8 gv = unpack8(*(const uint4*)(g + ((long long)(b * ho + oy) * wo + ox) * g_ld + gq * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[j] += win[j] ? gv.v[j] : 0.f;
      }
    }
    *(uint4*)(dx + p * dx_ld + gq * 8) = pack8(acc);
  }
}

__global__ __launch_bounds__(256) void relu_affine_bwd_kernel(const bf16_t* __restrict__ g1, int g1_ld, const bf16_t* __restrict__ g2, int g2_ld,
                                                               const bf16_t* __restrict__ a, int a_ld, const float* __restrict__ scale, int c,
                                                               long long pixels, int relu, bf16_t* __restrict__ dz, int dz_ld,
                                                               bf16_t* __restrict__ gm, int gm_ld) {
  const int groups = c >> 3;
  const long long total = pixels * groups;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int g = (int)(i % groups);
    const long long p = i / groups;
    bf8 d = unpack8(*(const uint4*)(g1 + p * g1_ld + g * 8));
    if (g2) {
      const bf8 e = unpack8(*(const uint4*)(g2 + p * g2_ld + g * 8));
#pragma unroll
      for (int j = 0; j < 8; ++j) d.v[j] = bf2f(f2bf(d.v[j] + e.v[j]));   // the sum is what an eager framework would have materialised
    }
    if (relu) {
      const bf8 av = unpack8(*(const uint4*)(a + p * a_ld + g * 8));
#pragma unroll
      for (int j = 0; j < 8; ++j) d.v[j] = av.v[j] > 0.f ? d.v[j] : 0.f;
    }
    if (gm) *(uint4*)(gm + p * gm_ld + g * 8) = pack8(d);
    if (dz) {
      if (scale) {
        const float4 s0 = *(const float4*)(scale + g * 8), s1 = *(const float4*)(scale + g * 8 + 4);
        d.v[0] *= s0.x; d.v[1] *= s0.y; d.v[2] *= s0.z; d.v[3] *= s0.w;
        d.v[4] *= s1.x; d.v[5] *= s1.y; d.v[6] *= s1.z; d.v[7] *= s1.w;
      }
      *(uint4*)(dz + p * dz_ld + g * 8) = pack8(d);
    }
  }
}

// out[b,Y,X,:] = (lat ? lat[b,Y,X,:] : 0) + x[b, Y*h/H, X*w/W, :]     (F.interpolate(mode="nearest", size=(H,W)))
__global__ __launch_bounds__(256) void upsample_nearest_add_kernel(const bf16_t* __restrict__ x, int x_ld, int n, int h, int w, int c,
                                                                    const bf16_t* __restrict__ lat, int lat_ld, int H, int W,
                                                                    bf16_t* __restrict__ out, int out_ld) {
  const int groups = c >> 3;
  const long long total = (long long)n * H * W * groups;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int g = (int)(i % groups);
    const long long p = i / groups;
    const int X = (int)(p % W), Y = (int)((p / W) % H), b = (int)(p / ((long long)W * H));
    const int sy = min((int)((long long)Y * h / H), h - 1), sx = min((int)((long long)X * w / W), w - 1);
    bf8 v = unpack8(*(const uint4*)(x + ((long long)(b * h + sy) * w + sx) * x_ld + g * 8));
    if (lat) {
      const bf8 l = unpack8(*(const uint4*)(lat + p * lat_ld + g * 8));
#pragma unroll
      for (int j = 0; j < 8; ++j) v.v[j] += l.v[j];
    }
    *(uint4*)(out + p * out_ld + g * 8) = pack8(v);
  }
}

// adjoint: out[b,y,x,:] (+)= sum of g over the destination pixels that read source pixel (y,x)
__global__ __launch_bounds__(256) void upsample_nearest_bwd_kernel(const bf16_t* __restrict__ gq, int g_ld, int n, int h, int w, int c, int H, int W,
                                                                    const bf16_t* __restrict__ acc, int acc_ld, bf16_t* __restrict__ out,
                                                                    int out_ld) {
  const int groups = c >> 3;
  const long long total = (long long)n * h * w * groups;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int g = (int)(i % groups);
    const long long p = i / groups;
    const int x = (int)(p % w), y = (int)((p / w) % h), b = (int)(p / ((long long)w * h));
    // destination rows Y with floor(Y*h/H) == y  <=>  Y in [ceil(y*H/h), ceil((y+1)*H/h))
    const int Y0 = (int)(((long long)y * H + h - 1) / h), Y1 = min(H, (int)(((long long)(y + 1) * H + h - 1) / h));
    const int X0 = (int)(((long long)x * W + w - 1) / w), X1 = min(W, (int)(((long long)(x + 1) * W + w - 1) / w));
    bf8 s;
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] = 0.f;
    if (acc) s = unpack8(*(const uint4*)(acc + p * acc_ld + g * 8));
    for (int Y = Y0; Y < Y1; ++Y)
      for (int X = X0; X < X1; ++X) {
        const bf8 v = unpack8(*(const uint4*)(gq + ((long long)(b * H + Y) * W + X) * g_ld + g * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) s.v[j] += v.v[j];
      }
    *(uint4*)(out + p * out_ld + g * 8) = pack8(s);
  }
}

// dst[b, r, 0..cols) (bf16, pitch dst_ld, cols..dst_ld zero filled) = src[b*src_img + r*src_row + 0..cols) * mul
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ src, long long src_img, long long src_row, int n, long long rows,
                                                         int cols, float mul, bf16_t* __restrict__ dst, int dst_ld) {
  const long long total = (long long)n * rows * dst_ld;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cc = (int)(i % dst_ld);
    const long long pr = i / dst_ld;
    const long long r = pr % rows;
    const int b = (int)(pr / rows);
    dst[i] = f2bf(cc < cols ? src[b * src_img + r * src_row + cc] * mul : 0.f);
  }
}

}  // namespace

extern "C" {

int mi355det_im2col_nchw(const float* img, const float* mean, const float* inv_std, void* out, int32_t n, int32_t c, int32_t h, int32_t w,
                         int32_t ksize, int32_t stride, int32_t pad, int32_t kpad, void* stream) {
  if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || ksize <= 0 || stride <= 0 || pad < 0) return fail(MI355DET_EINVAL, "%s: bad shape", "im2col_nchw");
  if (kpad % 8 != 0 || kpad < ksize * ksize * c) return fail(MI355DET_EINVAL, "%s: kpad must be a multiple of 8 and >= k*k*c", "im2col_nchw");
  if ((mean == nullptr) != (inv_std == nullptr)) return fail(MI355DET_EINVAL, "%s: mean and inv_std go together", "im2col_nchw");
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
  if ((long long)n * ho * wo * (kpad / 8) >= (1ll << 31) || ksize > 255 || c > 32767) return fail(MI355DET_EINVAL, "%s: problem too large", "im2col_nchw");
  hipLaunchKernelGGL(im2col_nchw_kernel, dim3(grid_for((long long)n * ho * wo * (kpad / 8))), dim3(256), sizeof(int) * kpad, S(stream), img, mean, inv_std,
                     (bf16_t*)out, n, c, h, w, ho, wo, ksize, stride, pad, kpad);
  return check_launch("im2col_nchw");
}

int mi355det_maxpool3x3s2(const void* x, int32_t x_ld, int32_t n, int32_t h, int32_t w, int32_t c, void* out, int32_t out_ld, void* stream) {
  if (c % 8 != 0) return fail(MI355DET_EINVAL, "%s: channels must be a multiple of 8", "maxpool3x3s2");
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for((long long)n * ho * wo * (c / 8))), dim3(256), 0, S(stream), (const bf16_t*)x, x_ld, n, h, w, c,
                     ho, wo, (bf16_t*)out, out_ld);
  return check_launch("maxpool3x3s2");
}

int mi355det_maxpool3x3s2_bwd(const void* x, int32_t x_ld, const void* g, int32_t g_ld, int32_t n, int32_t h, int32_t w, int32_t c, void* dx,
                              int32_t dx_ld, void* stream) {
  if (c % 8 != 0 || !x || !g || !dx) return fail(MI355DET_EINVAL, "%s: channels must be a multiple of 8", "maxpool3x3s2_bwd");
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(grid_for((long long)n * h * w * (c / 8))), dim3(256), 0, S(stream), (const bf16_t*)x, x_ld,
                     (const bf16_t*)g, g_ld, n, h, w, c, ho, wo, (bf16_t*)dx, dx_ld);
  return check_launch("maxpool3x3s2_bwd");
}

int mi355det_relu_affine_bwd(const void* g1, int32_t g1_ld, const void* g2, int32_t g2_ld, const void* a, int32_t a_ld, const float* scale,
                             int32_t c, int64_t pixels, int relu, void* dz, int32_t dz_ld, void* gm, int32_t gm_ld, void* stream) {
  if (c % 8 != 0) return fail(MI355DET_EINVAL, "%s: channels must be a multiple of 8", "relu_affine_bwd");
  if (relu && !a) return fail(MI355DET_EINVAL, "%s: relu backward needs the forward activation", "relu_affine_bwd");
  hipLaunchKernelGGL(relu_affine_bwd_kernel, dim3(grid_for(pixels * (c / 8))), dim3(256), 0, S(stream), (const bf16_t*)g1, g1_ld, (const bf16_t*)g2,
                     g2_ld, (const bf16_t*)a, a_ld, scale, c, (long long)pixels, relu, (bf16_t*)dz, dz_ld, (bf16_t*)gm, gm_ld);
  return check_launch("relu_affine_bwd");
}

int mi355det_upsample_nearest_add(const void* x, int32_t x_ld, int32_t n, int32_t h, int32_t w, int32_t c, const void* lateral, int32_t lateral_ld,
                                  int32_t out_h, int32_t out_w, void* out, int32_t out_ld, void* stream) {
  if (c % 8 != 0) return fail(MI355DET_EINVAL, "%s: channels must be a multiple of 8", "upsample_nearest_add");
  hipLaunchKernelGGL(upsample_nearest_add_kernel, dim3(grid_for((long long)n * out_h * out_w * (c / 8))), dim3(256), 0, S(stream), (const bf16_t*)x,
                     x_ld, n, h, w, c, (const bf16_t*)lateral, lateral_ld, out_h, out_w, (bf16_t*)out, out_ld);
  return check_launch("upsample_nearest_add");
}

int mi355det_upsample_nearest_bwd(const void* g, int32_t g_ld, int32_t n, int32_t h, int32_t w, int32_t c, int32_t g_h, int32_t g_w,
                                  const void* accumulate, int32_t accumulate_ld, void* out, int32_t out_ld, void* stream) {
  if (c % 8 != 0) return fail(MI355DET_EINVAL, "%s: channels must be a multiple of 8", "upsample_nearest_bwd");
  hipLaunchKernelGGL(upsample_nearest_bwd_kernel, dim3(grid_for((long long)n * h * w * (c / 8))), dim3(256), 0, S(stream), (const bf16_t*)g, g_ld, n, h,
                     w, c, g_h, g_w, (const bf16_t*)accumulate, accumulate_ld, (bf16_t*)out, out_ld);
  return check_launch("upsample_nearest_bwd");
}

int mi355det_cast_rows_bf16(const float* src, int64_t src_image_stride, int64_t src_row_stride, int32_t n, int64_t rows, int32_t cols, float mul,
                            void* dst, int32_t dst_ld, void* stream) {
  if (cols > dst_ld) return fail(MI355DET_EINVAL, "%s: dst pitch smaller than the row", "cast_rows_bf16");
  hipLaunchKernelGGL(cast_rows_kernel, dim3(grid_for((long long)n * rows * dst_ld)), dim3(256), 0, S(stream), src, (long long)src_image_stride,
                     (long long)src_row_stride, n, (long long)rows, cols, mul, (bf16_t*)dst, dst_ld);
  return check_launch("cast_rows_bf16");
}

}  // extern "C"
