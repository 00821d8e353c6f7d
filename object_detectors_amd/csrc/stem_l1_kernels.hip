// Darknet stem activation + first down-sampling convolution in ONE kernel (gfx950):
//
//   a0 = lrelu(bn1(conv1(img)))            darknet.py:41-43,74-76   (3 -> 32, 3x3 s1; recomputed from the fp32 image as in stem_kernels.hip)
//   z1 = layer1.ds_conv(a0)                darknet.py:64-66         (32 -> 64, 3x3 s2 p1) + its BatchNorm partial statistics
//
// Unfused, stem_fwd_apply writes a0 (839 MB at batch 32 / 640 px) and the implicit-GEMM forward of the 32 -> 64 stride-2 layer reads it back
// nine taps at a time: 221 + 379 us in the step against 225 us of HBM time for everything the pair has to move.  Here a workgroup owns an
// 8 x 16 tile of z1: it computes the 17 x 33 activation pixels that tile reads (one 16x16x32 MFMA pair per 16 pixels, BatchNorm + LeakyReLU
// in registers), parks them in LDS as bf16 - even and odd columns in separate planes, so that the stride-2 taps read CONSECUTIVE 64-byte
// pixels, 16-byte chunks XOR-swizzled by (index >> 1) & 3: conflict-free for both tap alignments (checked exhaustively) - and runs the
// 3x3 / stride-2 convolution from there: 9 taps x 2 channel fragments of weights stay in REGISTERS for the whole kernel (each wave owns 32
// output channels), so a tap costs one LDS read per 2 MFMAs.  The activation is written to HBM once, on the side, for the weight gradient
// of the layer (each workgroup stores the 16 x 32 pixels only it owns); z1 leaves as 16-byte stores of 8 consecutive channels per lane.
// Persistent workgroups, image halo of the next tile prefetched into registers; fixed-order statistics (one partial row per workgroup).
#include "common.h"

using namespace mi355;

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

namespace {

constexpr int OT_H = 8, OT_W = 16;                 // z1 tile
constexpr int AR = 2 * OT_H + 1, AC = 2 * OT_W + 1;   // activation tile 17 x 33 (rows / columns 2*o - 1 .. 2*o + 1)
constexpr int IR = AR + 2, IC = AC + 2;            // image halo 19 x 35
constexpr int IP = 36;                             // LDS pitch of an image row (bf16 elements)
constexpr int IMG_ELEMS = 3 * IR * IP;
constexpr int ZBASE = (IMG_ELEMS * 2 + 15) / 16 * 16;   // 128 zero bytes: k columns 27..31 of the im2col fragment
constexpr int DUMP = ZBASE + 128;
constexpr int IMG_BYTES = (DUMP + 16 + 63) / 64 * 64;
constexpr int HALO = 3 * IR * IC;                  // 1995 values per tile
constexpr int PER_T = (HALO + 255) / 256;          // 8 per thread
constexpr int A_IDX = OT_W + 1;                    // pixels per plane row (17 even-plane entries; the odd plane uses 16 of them)
constexpr int A_ROW = A_IDX * 64;                  // bytes per plane row
constexpr int A_PLANE = AR * A_ROW;
constexpr int A_BYTES = 2 * A_PLANE;
constexpr int NFRAG = 2 * AR + 2;                  // stem fragments per tile: 2 per activation row (columns 0-15, 16-31) + 2 for column 32
static_assert(NFRAG % 12 == 0, "whole groups of three fragments per wave");
static_assert(ZBASE % 16 == 0 && A_ROW % 64 == 0 && A_PLANE % 64 == 0, "alignment");

struct StemL1Params {
  const float* img;
  const bf16_t* w0;       // stem forward pack [32][32]
  const float* ss0;       // stem scale / shift [4*32]
  const bf16_t* w1;       // 32 -> 64 forward pack [64][9*32]
  bf16_t* a0;             // activation out or null, pitch a0_ld
  bf16_t* z1;             // pre-BN output, pitch z1_ld
  float* stats;           // [grid][2][64]
  const float* ss1;       // inference form: scale | shift [2*64] of layer 1's folded BatchNorm -> z1 receives lrelu(z*scale + shift), no statistics
  int a0_ld, z1_ld;
  int n, H, W, tiles_x, tiles_y, ntiles;
  float slope;
};

// workgroup barrier for LDS hand-offs only: this wave's LDS traffic has completed and the compiler moves no memory access across it.
// __syncthreads() also drains vmcnt, i.e. it waits for the prefetched image halo and for every output store of the tile (2-4 us of HBM
// latency per barrier: the persistent loops ran at half their speed with it).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

__global__ __launch_bounds__(256, 2) void stem_l1_kernel(const StemL1Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const a0s = smem + 2 * IMG_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int ch_half = wid & 1, row_half = wid >> 1;          // convolution phase: 32 output channels x 4 output rows per wave
  float* const aff1 = (float*)(smem + 2 * IMG_BYTES + A_BYTES + 64 * 16);      // inference form: scale | shift of layer 1 (128 floats)
  if (p.ss1 && tid < 128) aff1[tid] = p.ss1[tid];

  // ---- stem weights (row fr of fragment i = channel (fr/4)*8 + i*4 + fr%4: a lane's accumulators are 8 consecutive channels)
  st16x8_t wf0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) wf0[i] = *(const st16x8_t*)(p.w0 + ((fr >> 2) * 8 + i * 4 + (fr & 3)) * 32 + fq * 8);
  // ---- 32 -> 64 weights of this wave's 32 channels, all nine taps, resident in registers
  st16x8_t wf1[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      wf1[t][i] = *(const st16x8_t*)(p.w1 + (ch_half * 32 + (fr >> 2) * 8 + i * 4 + (fr & 3)) * 288 + t * 32 + fq * 8);
  // ---- im2col offsets of the stem: k = fq*8 + e -> (tap, c)
  int koff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = fq * 8 + e;
    const int t = k / 3, c = k - t * 3, kh = t / 3, kw = t - kh * 3;
    koff[e] = k < 27 ? ((c * IR + kh) * IP + kw) * 2 : -1;
  }
  float sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = p.ss0[fq * 8 + k];
    sh[k] = p.ss0[32 + fq * 8 + k];
  }
  float s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;

  // ---- image halo fetch roles (buffer loads: scalar tile offset, per-lane constant offset, out-of-image lanes read zero)
  int h_lds[PER_T], h_rel[PER_T];
  unsigned h_edges = 0;
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    const int e = tid + i * 256;
    const int cr = e / IC, x = e - cr * IC, c = cr / IR, r = cr - c * IR;
    h_lds[i] = e < HALO ? (c * IR + r) * IP + x : DUMP / 2;
    h_rel[i] = e < HALO ? ((c * p.H + r) * p.W + x) * 4 : (int)0x80000000;
    // rows 0, 1 lie above the image when the tile is at the top (origin - 2), columns 0, 1 left of it; the bottom / right halo never leaves
    // the image (the last activation row / column a stride-2, pad-1 layer reads is H-1 / W-1, plus one image row / column for the stem)
    h_edges |= (unsigned)((r < 2 ? 1 : 0) | (r == IR - 1 ? 2 : 0) | (x < 2 ? 4 : 0) | (x == IC - 1 ? 8 : 0)) << (4 * i);
  }
  const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc((void*)(p.img - 2 * (p.W + 1)), 0, 0x7FFFFFF0, 0x00020000);
  const bool write_a0 = p.a0 != nullptr;
  const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)(write_a0 ? p.a0 : p.z1), 0, write_a0 ? 0x7FFFFFF0 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_z1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.z1, 0, 0x7FFFFFF0, 0x00020000);
  auto tile_origin = [&](int tile, int& b, int& oy0, int& ox0) {
    const int tx = tile % p.tiles_x, t2 = tile / p.tiles_x;
    const int ty = t2 % p.tiles_y;
    b = t2 / p.tiles_y;
    oy0 = ty * OT_H;
    ox0 = tx * OT_W;
  };
  float hv[PER_T];
  auto fetch_halo = [&](int tile) {
    int b, oy0, ox0;
    tile_origin(tile, b, oy0, ox0);
    // image tile origin = (2*oy0 - 2, 2*ox0 - 2); the descriptor base sits two rows and two pixels in front of the image
    const int y0 = 2 * oy0, x0 = 2 * ox0;
    const unsigned edges = (y0 == 0 ? 1u : 0u) | (y0 + 2 * OT_H == p.H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + 2 * OT_W == p.W ? 8u : 0u);
    const int soff = ((b * 3 * p.H + y0) * p.W + x0) * 4;
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
      // at the top / left border only ONE of the two halo rows / columns is outside when the tile origin is 0: row 0 <-> image row -2,
      // row 1 <-> image row -1: both outside; at the bottom row IR-1 <-> image row H: outside
      hv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_img, ((h_edges >> (4 * i)) & edges) ? (int)0x80000000 : h_rel[i], soff, 0));
    }
  };
  auto store_halo = [&](int buf) {
    bf16_t* s = (bf16_t*)(smem + buf * IMG_BYTES);
#pragma unroll
    for (int i = 0; i < PER_T; ++i) s[h_lds[i]] = f2s(hv[i]);
  };
  if (tid < 64) *(unsigned*)(smem + (tid >> 5) * IMG_BYTES + ZBASE + (tid & 31) * 4) = 0u;

  const int G = gridDim.x;
  int tile = blockIdx.x;
  if (tile < p.ntiles) {
    fetch_halo(tile);
    store_halo(0);
  }
  __syncthreads();
  int buf = 0;
  for (; tile < p.ntiles; tile += G) {
    const int nxt = tile + G;
    const bool has_next = nxt < p.ntiles;
    if (has_next) fetch_halo(nxt);
    const char* simg = smem + buf * IMG_BYTES;
    int b, oy0, ox0;
    tile_origin(tile, b, oy0, ox0);
    // ================= stem phase: activation pixel (r, lc) of the 17 x 33 tile = image-local (r + kh, lc + kw).  36 fragments of 16
    // pixels: two per row (columns 0-15, 16-31) and two for column 32 (rows 0-15, row 16); nine per wave, unrolled so that the LDS
    // read -> MFMA -> BatchNorm -> LDS write chains of different fragments overlap
#pragma unroll 1
    for (int it0 = 0; it0 < NFRAG / 4; it0 += 3)
#pragma unroll
    for (int it = it0; it < it0 + 3; ++it) {
      const int f = wid + 4 * it;
      // branch-free on purpose: with branches the three fragments of a group run one after the other
      const bool rowf = f < 2 * AR;                                  // (wave-uniform) row fragment or column-32 fragment
      const int rr = rowf ? f >> 1 : (f - 2 * AR) * 16 + fr;
      const int lc = rowf ? (f & 1) * 16 + fr : AC - 1;
      const bool live = rr < AR;
      const int r = min(rr, AR - 1);
      const int pix = (r * IP + lc) * 2;
      unsigned short v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = *(const unsigned short*)(simg + (koff[e] >= 0 ? pix + koff[e] : ZBASE));
      uint4 u;
      u.x = v[0] | ((unsigned)v[1] << 16);
      u.y = v[2] | ((unsigned)v[3] << 16);
      u.z = v[4] | ((unsigned)v[5] << 16);
      u.w = v[6] | ((unsigned)v[7] << 16);
      const st16x8_t xf = __builtin_bit_cast(st16x8_t, u);
      f32x4_t acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = MI355_MFMA_16x16x32(wf0[i], xf, f32x4_t{0.f, 0.f, 0.f, 0.f});
      const int ay = 2 * oy0 - 1 + r, ax = 2 * ox0 - 1 + lc;        // activation pixel in the image frame
      const bool inside = ay >= 0 && ax >= 0;                       // row / column -1 is the convolution's zero padding, not a stem output
      unsigned short o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float y = acc[k >> 2][k & 3] * sc[k] + sh[k];
        o[k] = inside ? f2s(fmaxf(y, y * p.slope)) : (unsigned short)0;
      }
      uint4 w4;
      w4.x = o[0] | ((unsigned)o[1] << 16);
      w4.y = o[2] | ((unsigned)o[3] << 16);
      w4.z = o[4] | ((unsigned)o[5] << 16);
      w4.w = o[6] | ((unsigned)o[7] << 16);
      const int idx = lc >> 1;
      // dead lanes (rows 17.. of the last column fragment) park their value in a scratch slot behind the planes
      *(uint4*)(a0s + (live ? (lc & 1) * A_PLANE + r * A_ROW + idx * 64 + ((fq ^ ((idx >> 1) & 3)) << 4) : A_BYTES + lane * 16)) = w4;
      // side output: the 16 x 32 activation pixels only this tile owns; every other lane gets an out-of-range offset (the store is dropped)
      const bool own = write_a0 && live && r >= 1 && lc >= 1;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, w4), rs_a0,
                                             own ? (((b * p.H + ay) * p.W + ax) * p.a0_ld + fq * 8) * 2 : (int)0x80000000, 0, 0);
    }
    lds_barrier();
    // ================= convolution phase: z1 rows row_half*4 .. +3, 16 columns, channels ch_half*32 .. +31
    f32x4_t acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[j][i] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int idx = fr + (kw == 2 ? 1 : 0);
        const int col = (kw & 1) * A_PLANE + idx * 64 + ((fq ^ ((idx >> 1) & 3)) << 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const st16x8_t xf = *(const st16x8_t*)(a0s + (2 * (row_half * 4 + j) + kh) * A_ROW + col);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[j][i] = MI355_MFMA_16x16x32(wf1[kh * 3 + kw][i], xf, acc[j][i]);
        }
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned short o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float zz = acc[j][k >> 2][k & 3];
        if (p.ss1) {                                   // inference: folded BN + LeakyReLU, the activation itself is stored
          const int ch = ch_half * 32 + fq * 8 + k;
          zz = zz * aff1[ch] + aff1[64 + ch];
          zz = fmaxf(zz, zz * p.slope);
        }
        o[k] = f2s(zz);
        const float v = s2f(o[k]);                 // statistics of the STORED tensor (igemm_common.h, EPI_STATS)
        s1[k] += v;
        s2[k] += v * v;
      }
      uint4 w4;
      w4.x = o[0] | ((unsigned)o[1] << 16);
      w4.y = o[2] | ((unsigned)o[3] << 16);
      w4.z = o[4] | ((unsigned)o[5] << 16);
      w4.w = o[6] | ((unsigned)o[7] << 16);
      const int oy = oy0 + row_half * 4 + j, ox = ox0 + fr;
      // vector offset only: a 16-byte buffer store with an SGPR soffset gets no wait state before its data registers are reused (stem_kernels.hip)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, w4), rs_z1,
                                             (((b * (p.H / 2) + oy) * (p.W / 2) + ox) * p.z1_ld + ch_half * 32 + fq * 8) * 2, 0, 0);
    }
    if (has_next) store_halo(buf ^ 1);
    lds_barrier();
    buf ^= 1;
  }

  // ---- statistics: lanes of a 16-lane row hold different pixels of the same 8 channels; the two waves of a channel half add up
  __syncthreads();
  float* red = (float*)smem;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    s1[k] = row16_sum(s1[k]);
    s2[k] = row16_sum(s2[k]);
  }
  if (fr == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      red[(wid * 32 + fq * 8 + k) * 2 + 0] = s1[k];
      red[(wid * 32 + fq * 8 + k) * 2 + 1] = s2[k];
    }
  }
  __syncthreads();
  if (tid < 128) {
    const int ch = tid & 63, which = tid >> 6;
    const int hf = ch >> 5, c = ch & 31;
    if (p.stats) p.stats[(long long)blockIdx.x * 128 + which * 64 + ch] = red[((hf) * 32 + c) * 2 + which] + red[((hf + 2) * 32 + c) * 2 + which];
  }
}

int l1_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}
int l1_grid(int ntiles) {
  const int g = l1_cus() * 2;
  return ntiles < g ? ntiles : g;
}
bool l1_ok(int n, int h, int w) {
  return n > 0 && h > 0 && w > 0 && h % (2 * OT_H) == 0 && w % (2 * OT_W) == 0 && (long long)n * h * w * 64 < 0x7FFFFFF0ll;
}

}  // namespace

extern "C" {

int mi355det_stem_l1_rows(int32_t n, int32_t h, int32_t w) {
  if (!l1_ok(n, h, w)) return 0;
  return l1_grid(n * (h / (2 * OT_H)) * (w / (2 * OT_W)));
}

static int stem_l1_impl(const float* img, const void* w0, const float* scale_shift0, float slope, const void* w1, void* a0, int32_t a0_ld, void* z1,
                        int32_t z1_ld, float* stats, const float* ss1, int32_t n, int32_t h, int32_t w, void* stream) {
  if (!l1_ok(n, h, w)) return fail(MI355DET_EINVAL, "%s: needs h %% 16 == 0 and w %% 32 == 0 (got %lld x %lld)", "stem_l1_fwd", h, w);
  if (!img || !w0 || !scale_shift0 || !w1 || !z1 || (!stats && !ss1) || z1_ld < 64 || z1_ld % 8 || (a0 && (a0_ld < 32 || a0_ld % 8)))
    return fail(MI355DET_EINVAL, "%s: bad argument", "stem_l1_fwd");
  if (!(slope > 0.f && slope < 1.f)) return fail(MI355DET_EINVAL, "%s: LeakyReLU slope must be in (0, 1)", "stem_l1_fwd");
  if ((long long)n * h * w * (a0 ? a0_ld : 1) * 2 >= 0x7FFFFFF0ll || (long long)n * (h / 2) * (w / 2) * z1_ld * 2 >= 0x7FFFFFF0ll)
    return fail(MI355DET_EINVAL, "%s: tensor too large (32-bit byte offsets)", "stem_l1_fwd");
  StemL1Params p{};
  p.img = img;
  p.w0 = (const bf16_t*)w0;
  p.ss0 = scale_shift0;
  p.w1 = (const bf16_t*)w1;
  p.a0 = (bf16_t*)a0;
  p.z1 = (bf16_t*)z1;
  p.stats = stats;
  p.ss1 = ss1;
  p.a0_ld = a0_ld;
  p.z1_ld = z1_ld;
  p.n = n;
  p.H = h;
  p.W = w;
  p.tiles_x = w / (2 * OT_W);
  p.tiles_y = h / (2 * OT_H);
  p.ntiles = n * p.tiles_x * p.tiles_y;
  p.slope = slope;
  constexpr int lds = 2 * IMG_BYTES + A_BYTES + 64 * 16 + 128 * 4;
  static DeviceOnce once;
  once.once([&] { (void)hipFuncSetAttribute((const void*)stem_l1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
  hipLaunchKernelGGL(stem_l1_kernel, dim3(l1_grid(p.ntiles)), dim3(256), lds, S(stream), p);
  return check_launch("stem_l1_fwd");
}

int mi355det_stem_l1_fwd(const float* img, const void* w0, const float* scale_shift0, float slope, const void* w1, void* a0, int32_t a0_ld, void* z1,
                         int32_t z1_ld, float* stats, int32_t n, int32_t h, int32_t w, void* stream) {
  if (!stats) return fail(MI355DET_EINVAL, "%s: null statistics buffer", "stem_l1_fwd");
  return stem_l1_impl(img, w0, scale_shift0, slope, w1, a0, a0_ld, z1, z1_ld, stats, nullptr, n, h, w, stream);
}

int mi355det_stem_l1_fwd_eval(const float* img, const void* w0, const float* scale_shift0, float slope, const void* w1, const float* scale_shift1,
                              void* a1, int32_t a1_ld, int32_t n, int32_t h, int32_t w, void* stream) {
  if (!scale_shift1) return fail(MI355DET_EINVAL, "%s: null scale / shift", "stem_l1_fwd_eval");
  return stem_l1_impl(img, w0, scale_shift0, slope, w1, nullptr, 0, a1, a1_ld, nullptr, scale_shift1, n, h, w, stream);
}

}  // extern "C"
