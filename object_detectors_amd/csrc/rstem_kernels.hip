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
constexpr int ZSLOT = ((STRIDE * 15 * P + STRIDE * 15) * 2 + 2 + 15) / 16 * 16;      // the padded k columns read ZBASE + the pixel's own offset (rows 0..15, columns 0..15)
constexpr int DUMP = ZBASE + ZSLOT;                      // where the surplus lanes of the halo fetch put their value
constexpr int IMG_BYTES = DUMP + 16;
constexpr int HALO = 3 * HR * HC;                        // 4107 values per tile
constexpr int PER_T = (HALO + 255) / 256;                // 17 per thread
constexpr int WP = 168;                                  // LDS row pitch of the weights (bf16 elements)

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
  // the fields the loop needs, as plain scalars (nothing below takes the argument struct by address)
  const float* const img = p.img;
  bf16_t* const outp = p.out;
  const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo, ld = p.ld, relu = p.relu, tiles_x = p.tiles_x, tiles_y = p.tiles_y, ntiles = p.ntiles;

  // ---- weights in LDS ([64][WP] bf16, row pitch 336 B: the 16 rows a fragment read touches fall on distinct banks); as 20 register-resident
  //      A fragments per lane they left no room for the halo prefetch (the kernel spilled).  A operand: row fr of fragment i = channel
  //      (fr/4)*16 + i*4 + fr%4, k chunk q*32 + fq*8
  bf16_t* const wl = (bf16_t*)(smem + 2 * IMG_BYTES + 128 * 4 + KPAD * 4);
  for (int i = tid; i < 64 * (KPAD / 8); i += 256) {
    const int ch = i / (KPAD / 8), kc = i - ch * (KPAD / 8);
    *(uint4*)(wl + ch * WP + kc * 8) = *(const uint4*)(p.w + ch * KPAD + kc * 8);
  }
  int wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wrow[i] = (((fr >> 2) * 16 + i * 4 + (fr & 3)) * WP + fq * 8) * 2;
  // ---- im2col fragment (B operand): byte offset of k column k relative to the pixel's halo position, as a 160-entry LDS table (40 per-lane
  //      registers otherwise: the kernel spilled); the padded columns point at the zero slot
  int* const ktab = (int*)(smem + 2 * IMG_BYTES + 128 * 4);
  if (tid < KPAD) {
    const int k = tid;
    const int t = k / 3, c = k - t * 3, kh = t / KS, kw = t - kh * KS;
    ktab[k] = k < KVALID ? ((c * HR + kh) * P + kw) * 2 : ZBASE;
  }
  // pixel of fragment f (tile row 4*wid + f, column fr): halo byte offset
  const int pix_base = (STRIDE * (4 * wid) * P + STRIDE * fr) * 2;
  // ---- epilogue constants: scale | shift of the 64 channels behind the two image buffers (read back 16 + 16 per fragment: keeping them in
  //      registers spilled)
  float* const aff = (float*)(smem + 2 * IMG_BYTES);
  if (tid < 64) {
    aff[tid] = p.scale ? p.scale[tid] : 1.f;
    aff[64 + tid] = p.shift ? p.shift[tid] : 0.f;
  }

  // ---- halo fetch roles: element e = tid + i*256 of the [3][HR][HC] halo; (channel, row, column) are recomputed from e where they are needed
  //      (constant divisors: a few multiplies) - as a 17-entry per-thread array they lived in scratch, and every scratch access waits on
  //      the same counter as the halo loads in flight (the prefetch then stalled the compute: 610 us instead of ~200)
  auto halo_elem = [&](int i, int& c, int& r, int& x) __attribute__((always_inline)) {
    const int e = tid + i * 256;
    const int cr = e / HC;
    x = e - cr * HC;
    c = cr / HR;
    r = cr - c * HR;
    return e < HALO;
  };
  const float m0 = p.mean ? p.mean[0] : 0.f, m1 = p.mean ? p.mean[1] : 0.f, m2 = p.mean ? p.mean[2] : 0.f;
  const float i0 = p.istd ? p.istd[0] : 1.f, i1 = p.istd ? p.istd[1] : 1.f, i2 = p.istd ? p.istd[2] : 1.f;
  // Plain global loads / stores here, no buffer descriptors: in this kernel (249 VGPRs, long unrolled prologue) the compiler kept the
  // descriptors in VECTOR registers and turned every buffer access into a readfirstlane "waterfall" loop (636 us instead of ~200).
  // Image base one PAD row / column in front of the image: scalar offset = tile origin (never negative).
  auto tile_origin = [&](int tile, int& b, int& oy0, int& ox0) __attribute__((always_inline)) {
    const int tx = tile % tiles_x, t2 = tile / tiles_x;
    const int ty = t2 % tiles_y;
    b = t2 / tiles_y;
    oy0 = ty * TH;
    ox0 = tx * TW;
  };
  // fetch = 17 UNCONDITIONAL loads (clamped address) + a validity mask; normalisation, zero padding and the LDS store happen after the
  // tile's compute (store_halo).  Written as `ok ? (load - mean)*istd : 0` in one place the compiler made every load a branch of its own
  // with a full wait behind it: 17 serialised round trips per tile (134 of 299 us).
  float hv[PER_T];
  unsigned hok = 0;
  auto fetch_halo = [&](int tile) __attribute__((always_inline)) {
    int b, oy0, ox0;
    tile_origin(tile, b, oy0, ox0);
    const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;       // image coordinates of halo element (0, 0)
    const float* base = img + (long long)b * 3 * H * W;
    hok = 0;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      int c, r, x;
      const bool in = halo_elem(i, c, r, x);
      const int iy = iy0 + r, ix = ix0 + x;
      const bool ok = in && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      int idx = ok ? (c * H + iy) * W + ix : 0;
      asm volatile("" : "+v"(idx));                        // opaque: the load below stays unconditional
      hv[i] = base[idx];
      hok |= ok ? (1u << i) : 0u;
    }
  };
  auto store_halo = [&](int buf) __attribute__((always_inline)) {
    bf16_t* s = (bf16_t*)(smem + buf * IMG_BYTES);
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      int c, r, x;
      const bool in = halo_elem(i, c, r, x);
      // mean / 1/std of channel c without a lookup table (the compiler turns a select chain into one, in scratch)
      const float f1 = c >= 1 ? 1.f : 0.f, f2 = c >= 2 ? 1.f : 0.f;
      const float mu = m0 + f1 * (m1 - m0) + f2 * (m2 - m1), is = i0 + f1 * (i1 - i0) + f2 * (i2 - i1);
      const float v = ((hok >> i) & 1u) ? (hv[i] - mu) * is : 0.f;     // padding is zero AFTER normalisation (transform.py pads the normalised image)
      s[in ? (c * HR + r) * P + x : DUMP / 2] = f2bf(v);
    }
  };
  // zero slots behind both image buffers
  for (int i = tid; i < 2 * ((DUMP - ZBASE) / 4); i += 256) {
    const int b = i / ((DUMP - ZBASE) / 4), o = i - b * ((DUMP - ZBASE) / 4);
    *(unsigned*)(smem + b * IMG_BYTES + ZBASE + o * 4) = 0u;
  }

  const int G = gridDim.x;
  int tile = blockIdx.x;
  if (tile < ntiles) {
    fetch_halo(tile);
    store_halo(0);
  }
  __syncthreads();
  int buf = 0;
#pragma nounroll
  for (; tile < ntiles; tile += G) {
    const int nxt = tile + G;
    const bool has_next = nxt < ntiles;
    if (has_next && !(relu & 4)) fetch_halo(nxt);                       // lands while this tile is computed
    const char* simg = smem + buf * IMG_BYTES;
    int b, oy0, ox0;
    tile_origin(tile, b, oy0, ox0);
#pragma unroll 1
    for (int f2 = 0; f2 < ((relu & 2) ? 0 : 2); ++f2) {
      // two fragments (tile rows 4*wid + 2*f2 and + 1) per weight read
      f32x4_t acc[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[u][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const char* spix = simg + pix_base + (STRIDE * (2 * f2) * P) * 2;
      int wofs = 0;
      asm volatile("" : "+v"(wofs));          // opaque zero: keeps the weight reads inside the loop (hoisted, they are 80 live registers again)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        typedef __attribute__((ext_vector_type(4))) int i32x4_t;
        const i32x4_t t0 = *(const i32x4_t*)(ktab + q * 32 + fq * 8), t1 = *(const i32x4_t*)(ktab + q * 32 + fq * 8 + 4);
        const int ko[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
        bf16x8_t xf[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          unsigned short v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = *(const unsigned short*)(spix + u * (STRIDE * P * 2) + ko[e]);
          uint4 uu;
          uu.x = v[0] | ((unsigned)v[1] << 16);
          uu.y = v[2] | ((unsigned)v[3] << 16);
          uu.z = v[4] | ((unsigned)v[5] << 16);
          uu.w = v[6] | ((unsigned)v[7] << 16);
          xf[u] = __builtin_bit_cast(bf16x8_t, uu);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16x8_t wfr = *(const bf16x8_t*)((const char*)wl + wrow[i] + wofs + q * 64);
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, xf[u], acc[u][i], 0, 0, 0);
        }
      }
      // lane = pixel (row 4*wid + 2*f2 + u, column fr), channels fq*16 + i*4 + r
      float sc[16], sh[16];
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const f32x4_t a = *(const f32x4_t*)(aff + fq * 16 + k4 * 4), c = *(const f32x4_t*)(aff + 64 + fq * 16 + k4 * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc[k4 * 4 + r] = a[r];
          sh[k4 * 4 + r] = c[r];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        unsigned short o[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          float y = acc[u][k >> 2][k & 3] * sc[k] + sh[k];
          if (relu & 1) y = fmaxf(y, 0.f);
          o[k] = f2bf(y);
        }
        const int oy = oy0 + 4 * wid + 2 * f2 + u, ox = ox0 + fr;
        uint4 w0, w1;
        w0.x = o[0] | ((unsigned)o[1] << 16); w0.y = o[2] | ((unsigned)o[3] << 16); w0.z = o[4] | ((unsigned)o[5] << 16); w0.w = o[6] | ((unsigned)o[7] << 16);
        w1.x = o[8] | ((unsigned)o[9] << 16); w1.y = o[10] | ((unsigned)o[11] << 16); w1.z = o[12] | ((unsigned)o[13] << 16); w1.w = o[14] | ((unsigned)o[15] << 16);
        bf16_t* dst = outp + (long long)(((b * Ho + oy) * Wo + ox) * ld + fq * 16);
        *(uint4*)dst = w0;
        *(uint4*)(dst + 8) = w1;
      }
    }
    if (has_next && !(relu & 8)) store_halo(buf ^ 1);
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
  constexpr int lds = 2 * IMG_BYTES + 128 * 4 + KPAD * 4 + 64 * WP * 2;
  const int g = rstem_cus() * 2;
  hipLaunchKernelGGL(rstem_kernel, dim3(p.ntiles < g ? p.ntiles : g), dim3(256), lds, S(stream), p);
  return check_launch("resnet_stem_fwd");
}

}  // extern "C"
