// ResNet stem (utilities/resnet.py:173-176,232-234: conv1 7x7 / stride 2 / pad 3, 3 -> 64, FrozenBatchNorm2d, ReLU) as ONE direct
// convolution kernel for the frozen case (backbone_utils.py:89-104: conv1 is trained only with trainable_backbone_layers = 5).
//
// The im2col route (mi355det_im2col_nchw + a 160-deep GEMM) writes and re-reads an [n*400*400, 160] bf16 matrix - 819 MB each way at batch
// 16 / 800 px, 0.73 + 0.26 ms - for a convolution whose input is 123 MB and whose output is 328 MB.  Here, as in the Darknet stem
// (stem_kernels.hip), the image halo of a 16 x 16 output tile is staged in LDS as bf16 (normalised with the ImageNet mean / std of
// tvision/transform.py:120-124 when asked, padding pixels exact zeros), the im2col fragments are built in registers
// (k = (kh*7 + kw)*3 + c, 147 of 160 valid: five 32-deep MFMA steps), the 64 x 160 weights stay in registers as 20 A fragments, and the
// FrozenBN scale / shift + ReLU epilogue writes bf16 NHWC with 16-byte stores.
// Channel order trick (stem_kernels.hip): MFMA row r of weight fragment i stands for channel (r/4)*16 + i*4 + r%4, so that a lane's four
// accumulator quads are the 16 consecutive channels fq*16 .. fq*16+15 of its pixel.
#include "common.h"

#include <type_traits>

using namespace mi355;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace {

constexpr int KS = 7, STRIDE = 2, PAD = 3, KPAD = 160, KVALID = 147, NQ = KPAD / 32;
constexpr int TH = 16, TW = 16;                          // output tile
constexpr int HR = STRIDE * (TH - 1) + KS;               // 37 halo rows
constexpr int HC = STRIDE * (TW - 1) + KS;               // 37 halo columns
constexpr int P = 38;                                    // LDS row pitch (bf16 elements)
constexpr int IMG_ELEMS = 3 * HR * P;                    // 4218
constexpr int ZBASE = (IMG_ELEMS * 2 + 15) / 16 * 16;    // zero slot: the padded k columns 147..159 read it at every fragment-row offset
constexpr int ZSLOT = (STRIDE * 3 * P * 2 + 2 + 15) / 16 * 16;            // fragment rows 0..3 of a wave add up to STRIDE*3*P elements
constexpr int DUMP = ZBASE + ZSLOT;                      // where the surplus lanes of the halo fetch put their value
constexpr int IMG_BYTES = DUMP + 16;
constexpr int HALO = 3 * HR * HC;                        // 4107 values per tile
constexpr int PER_T = (HALO + 255) / 256;                // 17 per thread

struct RStemParams {
  const float* img;       // [n,3,H,W] fp32
  const float* mean;      // [3] or null
  const float* istd;      // [3] or null
  const bf16_t* w;        // packed forward weights [64][160] bf16, k = (kh*7+kw)*3 + c
  const float* scale;     // [64] or null
  const float* shift;     // [64] or null
  bf16_t* out;            // [n, Ho, Wo, 64] pitch ld
  int ld, relu;
  int n, H, W, Ho, Wo, tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(256, 2) void rstem_kernel(const RStemParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 image buffers]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  // ---- weight fragments (A operand): row fr of fragment i = channel (fr/4)*16 + i*4 + fr%4, k chunk q*32 + fq*8
  bf16x8_t wf[4][NQ];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = (fr >> 2) * 16 + i * 4 + (fr & 3);
#pragma unroll
    for (int q = 0; q < NQ; ++q) wf[i][q] = *(const bf16x8_t*)(p.w + ch * KPAD + q * 32 + fq * 8);
  }
  // ---- im2col fragment (B operand): this lane's 8 k values of step q = 8 LDS byte offsets relative to the pixel's halo position
  int kaddr[NQ][8];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = q * 32 + fq * 8 + e;
      const int t = k / 3, c = k - t * 3, kh = t / KS, kw = t - kh * KS;
      kaddr[q][e] = k < KVALID ? ((c * HR + kh) * P + kw) * 2 : -1;
    }
  // pixel of fragment f (tile row 4*wid + f, column fr): halo byte offset
  const int pix_base = (STRIDE * (4 * wid) * P + STRIDE * fr) * 2;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) kaddr[q][e] = kaddr[q][e] >= 0 ? pix_base + kaddr[q][e] : ZBASE;

  // ---- epilogue constants of this lane's 16 channels
  float sc[16], sh[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    sc[k] = p.scale ? p.scale[fq * 16 + k] : 1.f;
    sh[k] = p.shift ? p.shift[fq * 16 + k] : 0.f;
  }

  // ---- halo fetch roles (as stem_kernels.hip): element e of the [3][HR][HC] halo, packed (channel << 24 | row << 12 | column); validity is
  //      computed per tile from its origin (the image border cuts up to PAD rows / columns of halo)
  int h_rc[PER_T];
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    const int e = tid + i * 256;
    const int cr = e / HC, x = e - cr * HC, c = cr / HR, r = cr - c * HR;
    h_rc[i] = e < HALO ? (c << 24) | (r << 12) | x : -1;
  }
  const float m0 = p.mean ? p.mean[0] : 0.f, m1 = p.mean ? p.mean[1] : 0.f, m2 = p.mean ? p.mean[2] : 0.f;
  const float i0 = p.istd ? p.istd[0] : 1.f, i1 = p.istd ? p.istd[1] : 1.f, i2 = p.istd ? p.istd[2] : 1.f;
  // base one PAD row / column in front of the image: scalar offset = tile origin (never negative)
  const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc((void*)(p.img - (PAD * p.W + PAD)), 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x7FFFFFF0, 0x00020000);
  auto tile_origin = [&](int tile, int& b, int& oy0, int& ox0) {
    const int tx = tile % p.tiles_x, t2 = tile / p.tiles_x;
    const int ty = t2 % p.tiles_y;
    b = t2 / p.tiles_y;
    oy0 = ty * TH;
    ox0 = tx * TW;
  };
  float hv[PER_T];
  auto fetch_halo = [&](int tile) {
    int b, oy0, ox0;
    tile_origin(tile, b, oy0, ox0);
    const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;       // image coordinates of halo element (0, 0)
    const int soff = ((b * 3 * p.H + oy0 * STRIDE) * p.W + ox0 * STRIDE) * 4;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int c = h_rc[i] >> 24, r = (h_rc[i] >> 12) & 0xFFF, x = h_rc[i] & 0xFFF;
      const bool ok = h_rc[i] >= 0 && (unsigned)(iy0 + r) < (unsigned)p.H && (unsigned)(ix0 + x) < (unsigned)p.W;
      const int rel = ((c * p.H + r) * p.W + x) * 4;
      const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_img, ok ? rel : (int)0x80000000, soff, 0));
      const float mu = c == 0 ? m0 : (c == 1 ? m1 : m2), is = c == 0 ? i0 : (c == 1 ? i1 : i2);
      hv[i] = ok ? (v - mu) * is : 0.f;                  // padding is zero AFTER normalisation (transform.py pads the normalised image)
    }
  };
  auto store_halo = [&](int buf) {
    bf16_t* s = (bf16_t*)(smem + buf * IMG_BYTES);
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      const int c = h_rc[i] >> 24, r = (h_rc[i] >> 12) & 0xFFF, x = h_rc[i] & 0xFFF;
      s[h_rc[i] >= 0 ? (c * HR + r) * P + x : DUMP / 2] = f2bf(hv[i]);
    }
  };
  // zero slots behind both image buffers
  for (int i = tid; i < 2 * ((DUMP - ZBASE) / 4); i += 256) {
    const int b = i / ((DUMP - ZBASE) / 4), o = i - b * ((DUMP - ZBASE) / 4);
    *(unsigned*)(smem + b * IMG_BYTES + ZBASE + o * 4) = 0u;
  }

  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const int G = gridDim.x;
  int tile = blockIdx.x;
  if (tile < p.ntiles) {
    fetch_halo(tile);
    store_halo(0);
  }
  __syncthreads();
  int buf = 0;
#pragma nounroll
  for (; tile < p.ntiles; tile += G) {
    const int nxt = tile + G;
    const bool has_next = nxt < p.ntiles;
    if (has_next) fetch_halo(nxt);                       // lands while this tile is computed
    const char* simg = smem + buf * IMG_BYTES;
    int b, oy0, ox0;
    tile_origin(tile, b, oy0, ox0);
#pragma unroll 1
    for (int f = 0; f < 4; ++f) {
      const int foff = (STRIDE * f * P) * 2;
      f32x4_t acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *(const unsigned short*)(simg + kaddr[q][e] + foff);
        uint4 u;
        u.x = v[0] | ((unsigned)v[1] << 16);
        u.y = v[2] | ((unsigned)v[3] << 16);
        u.z = v[4] | ((unsigned)v[5] << 16);
        u.w = v[6] | ((unsigned)v[7] << 16);
        const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, u);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][q], xf, acc[i], 0, 0, 0);
      }
      // lane = pixel (row 4*wid + f, column fr), channels fq*16 + i*4 + r
      unsigned short o[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        float y = acc[k >> 2][k & 3] * sc[k] + sh[k];
        if (p.relu) y = fmaxf(y, 0.f);
        o[k] = f2bf(y);
      }
      const int oy = oy0 + 4 * wid + f, ox = ox0 + fr;
      const int off = (((b * p.Ho + oy) * p.Wo + ox) * p.ld + fq * 16) * 2;
      uint4 w0, w1;
      w0.x = o[0] | ((unsigned)o[1] << 16); w0.y = o[2] | ((unsigned)o[3] << 16); w0.z = o[4] | ((unsigned)o[5] << 16); w0.w = o[6] | ((unsigned)o[7] << 16);
      w1.x = o[8] | ((unsigned)o[9] << 16); w1.y = o[10] | ((unsigned)o[11] << 16); w1.z = o[12] | ((unsigned)o[13] << 16); w1.w = o[14] | ((unsigned)o[15] << 16);
      // vector offset only (no SGPR soffset): see the store hazard note in stem_kernels.hip
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, w0), rs_out, off, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, w1), rs_out, off + 16, 0, 0);
    }
    if (has_next) store_halo(buf ^ 1);
    lds_barrier();
    buf ^= 1;
  }
}

int rstem_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

}  // namespace

extern "C" {

int mi355det_resnet_stem_fwd(const float* img, const float* mean, const float* inv_std, const void* w, const float* scale, const float* shift,
                             int32_t relu, void* out, int32_t out_ld, int32_t n, int32_t h, int32_t wd, void* stream) {
  if (n <= 0 || h <= 0 || wd <= 0 || h % 32 != 0 || wd % 32 != 0)
    return fail(MI355DET_EINVAL, "%s: needs h %% 32 == 0 and w %% 32 == 0 (got %lld x %lld)", "resnet_stem_fwd", h, wd);
  if (!img || !w || !out || out_ld < 64 || out_ld % 8 != 0 || (!mean) != (!inv_std)) return fail(MI355DET_EINVAL, "%s: bad argument", "resnet_stem_fwd");
  if ((long long)n * h * wd * 12 >= 0x7FFFFFF0ll || (long long)n * (h / 2) * (wd / 2) * out_ld * 2 >= 0x7FFFFFF0ll)
    return fail(MI355DET_EINVAL, "%s: tensor too large (32-bit byte offsets)", "resnet_stem_fwd");
  RStemParams p{};
  p.img = img; p.mean = mean; p.istd = inv_std; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift;
  p.out = (bf16_t*)out; p.ld = out_ld; p.relu = relu;
  p.n = n; p.H = h; p.W = wd; p.Ho = h / 2; p.Wo = wd / 2;
  p.tiles_x = p.Wo / TW; p.tiles_y = p.Ho / TH;
  p.ntiles = n * p.tiles_x * p.tiles_y;
  constexpr int lds = 2 * IMG_BYTES;
  const int g = rstem_cus() * 2;
  hipLaunchKernelGGL(rstem_kernel, dim3(p.ntiles < g ? p.ntiles : g), dim3(256), lds, S(stream), p);
  return check_launch("resnet_stem_fwd");
}

}  // extern "C"
