// Darknet stem (darknet.py:41-43, 74-76: conv1 3->32 3x3 s1 p1 -> bn1 -> LeakyReLU) as ONE family of recompute kernels (gfx950).
//
// The stem's output is the largest tensor of the network (n*H*W*32: 839 MB of bf16 at batch 32 / 640 px) and its convolution the
// cheapest (K = 27): every pass over a stored pre-BN tensor z costs more HBM time than recomputing z from the 157 MB fp32 image.  So z is
// never stored.  All four kernels share one core - stage the image halo of an 8 x 32 pixel tile in LDS as bf16, build the im2col
// fragments in registers (k = (kh*3+kw)*3 + c, 27 of 32 valid), one 16x16x32 MFMA per 16 pixels x 16 channels - and differ in what they
// do with the fp32 z tile that is then in registers (lane = one pixel, 8 consecutive channels):
//
//   MODE 0  stem_fwd_stats        sum z, sum z^2 per channel                       -> partial rows (bn_finalize makes scale / shift)
//   MODE 1  stem_fwd_apply        a = lrelu(z*scale + shift)                       -> a (bf16 NHWC), 16-byte stores, 1 KB per wave instruction
//   MODE 2  stem_bwd_reduce       dy = da * lrelu'(.), sum dy, sum dy*xhat         -> partial rows (bn_bwd_sum_partials makes the sums)
//   MODE 3  stem_bwd_apply_wgrad  dz = scale*(dy - mean dy - xhat*mean dy*xhat)    -> dW[32][32] += dz^T * im2col (MFMA over pixels: dz goes
//                                 through a wave-private LDS tile and comes back transposed with ds_read_b64_tr_b16); dz never exists in HBM
//   MODE 4  stem_bwd_fused        ONE backward pass instead of MODE 2 + MODE 3: A[32][32] = dy^T * [im2col | 1] and the Gram matrix
//                                 G[32][32] = [im2col | 1]^T * [im2col | 1] (both on MFMA over pixels).  Everything the stem's backward
//                                 needs is linear in them (stem_bwd_finish_kernel): sum dy = A[:,27], sum dy*z = <W, A>, and
//                                 dW = scale*(A - mean(dy)*B - mean(dy*xhat)*invstd*(W*G - mean*B)) with B = G[27,:] = sum im2col
//
// Channel order trick: MFMA row r of weight fragment i stands for channel (r/4)*8 + i*4 + r%4, so that a lane's two accumulator quads
// are the 8 CONSECUTIVE channels fq*8 .. fq*8+7 of its pixel: loads and stores of the activation / gradient tensors are 16 bytes per lane
// and a wave instruction covers 16 whole pixels (1 KB contiguous) without an LDS transpose.
// Persistent workgroups (grid = 4 per CU), tiles dealt round-robin; the next tile's image halo and gradient tile are fetched into registers
// while the current one is computed.  Everything is fixed-order (bit-reproducible), unlike the atomics of bn_act_bwd_reduce.
#include "common.h"

#include <type_traits>

using namespace mi355;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace {

constexpr int TH = 8, TW = 32;             // output tile
constexpr int HR = TH + 2, HC = TW + 2;    // halo tile
constexpr int P = 36;                      // LDS row pitch of the halo tile (bf16 elements)
constexpr int IMG_ELEMS = 3 * HR * P;      // 1080
constexpr int ZBASE = IMG_ELEMS * 2;        // 128 zero bytes behind the halo tile: the padded k columns 27..31 read them at every fragment offset
constexpr int DUMP = ZBASE + 128;           // where the surplus lanes of the halo fetch (1020 values on 1024 lanes) put their value
constexpr int ONES = DUMP + 16;             // 96 bytes of bf16 1.0: the k = 27 column of MODE 4's im2col operand (sum dy and sum im2col come out of the same MFMAs)
constexpr int IMG_BYTES = ONES + 96;
static_assert(ZBASE % 16 == 0 && (P + 16) * 2 + 2 <= 128 && (P + 7) * 2 + 2 <= 128, "zero slot covers every fragment offset");
static_assert(IMG_BYTES % 16 == 0 && (P + 7) * 2 + 2 <= 96, "ones slot covers every wgrad operand offset");
constexpr int HALO = 3 * HR * HC;          // 1020 values to fetch per tile
constexpr int PER_T = (HALO + 255) / 256;  // 4 per thread
constexpr int DZ_WAVE = 2 * TW * 64;       // wave-private dz tile: 2 rows x 32 pixels x 32 channels bf16

struct StemParams {
  const float* img;       // [n,3,H,W] fp32
  const bf16_t* w;        // packed forward weights [32][32] bf16, k = (kh*3+kw)*3 + c
  const float* ss;        // [4*32] scale, shift, mean, invstd (MODE 1-3)
  const float* sums;      // [2*32] sum dy, sum dy*xhat (MODE 3)
  const bf16_t* da;       // gradient of the activation (MODE 2, 3), pitch ld
  bf16_t* a;              // activation out (MODE 1), pitch ld
  float* partial;         // MODE 0 / 2: [grid][2][32]; MODE 3: slab [grid][32][32]; MODE 4: slab [grid][2][32][32] (A, G)
  int ld;
  int n, H, W, tiles_x, tiles_y, ntiles;
  int rows_total;         // rows the caller allocated (mi355det_stem_rows): the rows beyond this launch's grid are zero-filled
  float slope, inv_count;
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

__device__ __forceinline__ st16x4_t lds_tr(const char* p) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
  return __builtin_bit_cast(st16x4_t, v);
}

template <int MODE>
__global__ __launch_bounds__(256, (MODE == 4 ? 2 : MODE >= 2 ? 3 : 4)) void stem_kernel(const StemParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [2 image buffers][MODE 3: 4 wave-private dz tiles][end-of-kernel reduction scratch aliases the front]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  // ---- weight fragments (A operand: row fr of fragment i = channel (fr/4)*8 + i*4 + fr%4, k chunk fq)
  st16x8_t wf[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = (fr >> 2) * 8 + i * 4 + (fr & 3);
    wf[i] = *(const st16x8_t*)(p.w + ch * 32 + fq * 8);
  }
  // ---- im2col fragment (B operand): this lane's 8 k values of a pixel = 8 LDS element offsets relative to the pixel's halo position
  int koff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = fq * 8 + e;
    const int t = k / 3, c = k - t * 3, kh = t / 3, kw = t - kh * 3;
    koff[e] = k < 27 ? ((c * HR + kh) * P + kw) * 2 : -1;
  }
  const int pix_base = ((2 * wid) * P + fr) * 2;      // byte offset of this lane's pixel of fragment 0 (row 2*wid, column fr)
  int kaddr[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) kaddr[e] = koff[e] >= 0 ? pix_base + koff[e] : ZBASE;

  // ---- per-channel constants of this lane's 8 channels
  float sc[8], sh[8], ca[8], cb[8], cc[8];
  if (MODE >= 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ch = fq * 8 + k;
      sc[k] = p.ss[ch];
      sh[k] = p.ss[32 + ch];
      const float mu = p.ss[64 + ch], is = p.ss[96 + ch];
      if (MODE == 2) {          // xhat = z*is - mu*is
        ca[k] = is;
        cb[k] = -mu * is;
      }
      if (MODE == 3) {          // dz = sc*dy + cb*z + cc  with  cb = -sc*m2*is,  cc = -sc*m1 + sc*m2*is*mu
        const float m1 = p.sums[ch] * p.inv_count, m2 = p.sums[32 + ch] * p.inv_count;
        ca[k] = sc[k];
        cb[k] = -sc[k] * m2 * is;
        cc[k] = -sc[k] * m1 + sc[k] * m2 * is * mu;
      }
    }
  }
  float s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;
  f32x4_t accw[2][2], accg[MODE == 4 ? 2 : 1][MODE == 4 ? 2 : 1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      accw[i][j] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (MODE == 4) accg[i][j] = {0.f, 0.f, 0.f, 0.f};
    }

  // wgrad B operand (MODE 3, 4): lane = k column fr of block jb, pixels fq*8 .. fq*8+7 of a tile row -> 8 consecutive halo elements;
  // MODE 4: column 27 reads the ones slot
  int wb_addr[2];
  if (MODE >= 3) {
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      const int k = jb * 16 + fr;
      const int t = k / 3, c = k - t * 3, kh = t / 3, kw = t - kh * 3;
      wb_addr[jb] = k < 27 ? (((c * HR + kh + 2 * wid) * P) + kw + fq * 8) * 2 : (MODE == 4 && k == 27 ? ONES : ZBASE);
    }
  }
  char* const dzt = smem + 2 * IMG_BYTES + wid * DZ_WAVE;

  // ---- halo fetch roles: element e of the [3][HR][HC] halo -> LDS element offset (or -1), its byte offset relative to the tile's first halo
  //      pixel, and which tile edges it lies on (bit 0 top, 1 bottom, 2 left, 3 right: outside the image when the tile touches that border).
  //      Loads go through a buffer descriptor whose base sits one row and one pixel in front of the image: scalar offset = tile origin
  //      (never negative), per-lane offset = loop constant, out-of-image lanes get an out-of-range offset and read 0 - no branches.
  int h_lds[PER_T], h_rel[PER_T], h_edges = 0;      // h_edges: 4 bits per fetched element
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    const int e = tid + i * 256;
    const int cr = e / HC, x = e - cr * HC, c = cr / HR, r = cr - c * HR;
    h_lds[i] = e < HALO ? (c * HR + r) * P + x : DUMP / 2;
    h_rel[i] = e < HALO ? ((c * p.H + r) * p.W + x) * 4 : (int)0x80000000;
    h_edges |= ((r == 0 ? 1 : 0) | (r == HR - 1 ? 2 : 0) | (x == 0 ? 4 : 0) | (x == HC - 1 ? 8 : 0)) << (4 * i);
  }
  const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc((void*)(p.img - (p.W + 1)), 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_act = __builtin_amdgcn_make_buffer_rsrc((void*)(MODE == 1 ? (const bf16_t*)p.a : p.da), 0, 0x7FFFFFF0, 0x00020000);
  const int act_lane = (fr * p.ld + fq * 8) * 2;      // byte offset of this lane's 16 bytes inside a 16-pixel fragment
  auto tile_origin = [&](int tile, int& b, int& y0, int& x0) {
    const int tx = tile % p.tiles_x, t2 = tile / p.tiles_x;
    const int ty = t2 % p.tiles_y;
    b = t2 / p.tiles_y;
    y0 = ty * TH;
    x0 = tx * TW;
  };
  // Prefetch ring, two tiles deep: under a store-heavy load a global load takes several microseconds, and with one tile of look-ahead
  // every iteration of a workgroup waited for one (stem_fwd_apply: 251 us with one tile of look-ahead).  Slot s = iteration parity.
  float hv[2][PER_T];
  auto fetch_halo = [&](int tile, auto SLOT) {
    constexpr int s_ = decltype(SLOT)::value;
    int b, y0, x0;
    tile_origin(tile, b, y0, x0);
    const int edges = (y0 == 0 ? 1 : 0) | (y0 + TH == p.H ? 2 : 0) | (x0 == 0 ? 4 : 0) | (x0 + TW == p.W ? 8 : 0);
    const int soff = ((b * 3 * p.H + y0) * p.W + x0) * 4;
#pragma unroll
    for (int i = 0; i < PER_T; ++i)
      hv[s_][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_img, ((h_edges >> (4 * i)) & edges) ? (int)0x80000000 : h_rel[i], soff, 0));
  };
  auto store_halo = [&](int buf, auto SLOT) {
    constexpr int s_ = decltype(SLOT)::value;
    bf16_t* s = (bf16_t*)(smem + buf * IMG_BYTES);
#pragma unroll
    for (int i = 0; i < PER_T; ++i) s[h_lds[i]] = f2s(hv[s_][i]);
  };
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  // activation / gradient tile: fragment f = 16 consecutive pixels of tile row 2*wid + f/2; scalar offset = the fragment's first pixel
  auto frag_soff = [&](int b, int y0, int x0, int f) { return (((b * p.H + y0 + 2 * wid + (f >> 1)) * p.W + x0 + (f & 1) * 16) * p.ld) * 2; };
  uint4 gv[2][4];
  auto fetch_grad = [&](int tile, auto SLOT) {
    constexpr int s_ = decltype(SLOT)::value;
    int b, y0, x0;
    tile_origin(tile, b, y0, x0);
#pragma unroll
    for (int f = 0; f < 4; ++f)
      gv[s_][f] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_act, act_lane, frag_soff(b, y0, x0, f), 0));
  };

  // the zero slots behind both image buffers (store_halo never touches them), and the ones slots
  if (tid < 64) *(unsigned*)(smem + (tid >> 5) * IMG_BYTES + ZBASE + (tid & 31) * 4) = 0u;
  if (tid >= 64 && tid < 112) *(unsigned*)(smem + ((tid - 64) / 24) * IMG_BYTES + ONES + ((tid - 64) % 24) * 4) = MI355_F16 ? 0x3C003C00u : 0x3F803F80u;      // two stored 1.0

  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;
  const int G = gridDim.x;
  int tile = blockIdx.x;
  // prologue: tile 0 of this workgroup into LDS buffer 0, tile 1 on its way into ring slot 1, gradients of both
  if (tile < p.ntiles) {
    fetch_halo(tile, S0{});
    if (MODE >= 2) fetch_grad(tile, S0{});
    if (tile + G < p.ntiles) {
      fetch_halo(tile + G, S1{});
      if (MODE >= 2) fetch_grad(tile + G, S1{});
    }
    store_halo(0, S0{});
  }
  __syncthreads();
  int buf = 0;
  // iteration with parity s: computes `tile` (LDS buffer `buf`, gradient slot s), fetches the halo of tile + 2G into slot s (free: it
  // went to LDS one iteration ago), stores the halo of tile + G (slot 1 - s, fetched one iteration ago) into the other LDS buffer
  auto iteration = [&](auto SLOT) {
    constexpr int s_ = decltype(SLOT)::value;
    typedef std::integral_constant<int, 1 - s_> OTHER;
    const int nxt = tile + G, nxt2 = tile + 2 * G;
    const bool has_next = nxt < p.ntiles;
    if (nxt2 < p.ntiles) fetch_halo(nxt2, SLOT);
    const char* simg = smem + buf * IMG_BYTES;
    int b, y0, x0;
    tile_origin(tile, b, y0, x0);
    // ---- fragment f = tile row 2*wid + f/2, columns (f%2)*16 + fr: z of the lane's pixel, channels fq*8 + k (k = i*4 + r)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int foff = ((f >> 1) * P + (f & 1) * 16) * 2;
      unsigned short v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = *(const unsigned short*)(simg + kaddr[e] + foff);
      uint4 u;
      u.x = v[0] | ((unsigned)v[1] << 16);
      u.y = v[2] | ((unsigned)v[3] << 16);
      u.z = v[4] | ((unsigned)v[5] << 16);
      u.w = v[6] | ((unsigned)v[7] << 16);
      const st16x8_t xf = __builtin_bit_cast(st16x8_t, u);
      f32x4_t acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = MI355_MFMA_16x16x32(wf[i], xf, f32x4_t{0.f, 0.f, 0.f, 0.f});
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float z = acc[k >> 2][k & 3];
          s1[k] += z;
          s2[k] += z * z;
        }
      }
      if (MODE == 1) {
        unsigned short o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float y = acc[k >> 2][k & 3] * sc[k] + sh[k];
          o[k] = f2s(fmaxf(y, y * p.slope));         // LeakyReLU, 0 < slope < 1
        }
        uint4 w4;
        w4.x = o[0] | ((unsigned)o[1] << 16);
        w4.y = o[2] | ((unsigned)o[3] << 16);
        w4.z = o[4] | ((unsigned)o[5] << 16);
        w4.w = o[6] | ((unsigned)o[7] << 16);
        // The tile offset goes into the VECTOR offset here (scalar offset 0), unlike the loads.  A 16-byte buffer store reads its data
        // registers over several cycles; with an SGPR in the soffset field the compiler assumes that overwriting them in the very next
        // instruction is safe and inserts no wait state - on gfx950 it is not: the last lanes (12-15 of every 16) of the LAST data dword
        // picked up the next fragment's value whenever another process shared the GPU (tools/debug_stem_stress.py: 4-56 wrong pixels in
        // ~2 % of the launches).  Without an soffset register the hazard recogniser adds the required wait state itself.
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, w4), rs_act, act_lane + frag_soff(b, y0, x0, f), 0, 0);
      }
      if (MODE == 2) {
        const unsigned gi[4] = {gv[s_][f].x, gv[s_][f].y, gv[s_][f].z, gv[s_][f].w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float z = acc[k >> 2][k & 3];
          const float gg = s2f((bf16_t)((k & 1) ? gi[k >> 1] >> 16 : gi[k >> 1] & 0xFFFFu));
          const float y = z * sc[k] + sh[k];
          const float dy = y > 0.f ? gg : gg * p.slope;
          s1[k] += dy;
          s2[k] += dy * (z * ca[k] + cb[k]);
        }
      }
      if (MODE >= 3) {
        // dz (MODE 4: dy) -> wave-private LDS tile [pixel row 0..63][32 channels], 32-byte blocks XOR-swizzled by bit 3 of the pixel row
        const unsigned gi[4] = {gv[s_][f].x, gv[s_][f].y, gv[s_][f].z, gv[s_][f].w};
        unsigned short o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float z = acc[k >> 2][k & 3];
          const float gg = s2f((bf16_t)((k & 1) ? gi[k >> 1] >> 16 : gi[k >> 1] & 0xFFFFu));
          const float y = z * sc[k] + sh[k];
          const float dy = y > 0.f ? gg : gg * p.slope;
          o[k] = f2s(MODE == 4 ? dy : ca[k] * dy + (cb[k] * z + cc[k]));
        }
        uint4 w4;
        w4.x = o[0] | ((unsigned)o[1] << 16);
        w4.y = o[2] | ((unsigned)o[3] << 16);
        w4.z = o[4] | ((unsigned)o[5] << 16);
        w4.w = o[6] | ((unsigned)o[7] << 16);
        const int prow = f * 16 + fr;                                  // pixel row of the wave tile: (f/2)*32 + (f%2)*16 + fr
        const int blk = (fq >> 1) ^ ((prow >> 3) & 1);
        *(uint4*)(dzt + prow * 64 + blk * 32 + (fq & 1) * 16) = w4;
      }
    }
    if (MODE >= 2 && nxt2 < p.ntiles) fetch_grad(nxt2, SLOT);      // this slot's gradient registers are consumed: refill them two tiles ahead
    if (MODE >= 3) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      // dW[co][k] += sum over the 32 pixels of each of the wave's two tile rows
      const int g4 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        st16x8_t af[2], bfr[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int prow = rr * 32 + 8 * g4 + 4 * h + q;
          const int sw = (prow >> 3) & 1;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const st16x4_t t4 = lds_tr(dzt + prow * 64 + ((i ^ sw) << 5) + pp * 8);
            af[i][4 * h + 0] = t4[0];
            af[i][4 * h + 1] = t4[1];
            af[i][4 * h + 2] = t4[2];
            af[i][4 * h + 3] = t4[3];
          }
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
          unsigned short v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = *(const unsigned short*)(simg + wb_addr[jb] + (rr * P + e) * 2);
          uint4 u;
          u.x = v[0] | ((unsigned)v[1] << 16);
          u.y = v[2] | ((unsigned)v[3] << 16);
          u.z = v[4] | ((unsigned)v[5] << 16);
          u.w = v[6] | ((unsigned)v[7] << 16);
          bfr[jb] = __builtin_bit_cast(st16x8_t, u);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jb = 0; jb < 2; ++jb) accw[i][jb] = MI355_MFMA_16x16x32(af[i], bfr[jb], accw[i][jb]);
        if constexpr (MODE == 4) {
          // Gram matrix of the (im2col | 1) operand: the B fragment of columns ja*16 + fr IS the A fragment of rows ja*16 + fr of its transpose
#pragma unroll
          for (int ja = 0; ja < 2; ++ja)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) accg[ja][jb] = MI355_MFMA_16x16x32(bfr[ja], bfr[jb], accg[ja][jb]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (has_next) store_halo(buf ^ 1, OTHER{});
    lds_barrier();
    buf ^= 1;
    tile = nxt;
  };
#pragma nounroll
  while (tile < p.ntiles) {
    iteration(S0{});
    if (tile >= p.ntiles) break;
    iteration(S1{});
  }

  // ---- end of the persistent loop: fold the per-lane sums (fixed order) and write this workgroup's row
  __syncthreads();
  float* red = (float*)smem;
  {
    const int rowf = MODE == 4 ? 2048 : MODE == 3 ? 1024 : 64;
    for (int r = blockIdx.x + gridDim.x; r < p.rows_total; r += gridDim.x)
      for (int i = tid; i < rowf; i += 256) p.partial[(long long)r * rowf + i] = 0.f;
  }
  if (MODE == 0 || MODE == 2) {
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
    if (tid < 64) {
      const int ch = tid & 31, which = tid >> 5;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[(w * 32 + ch) * 2 + which];
      p.partial[(long long)blockIdx.x * 64 + which * 32 + ch] = s;
    }
  }
  if (MODE == 3) {
    // accw[i][jb][r] = dW[co = (fq_row...)] - rows are MFMA rows of the TRANSPOSED dz fragment i: plain channels i*16 + fq*4 + r; column = jb*16 + fr
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wid * 1024 + (i * 16 + fq * 4 + r) * 32 + jb * 16 + fr] = accw[i][jb][r];
    __syncthreads();
    float* dst = p.partial + (long long)blockIdx.x * 1024;
    for (int i = tid; i < 1024; i += 256) dst[i] = (red[i] + red[1024 + i]) + (red[2048 + i] + red[3072 + i]);
  }
  if constexpr (MODE == 4) {
    // per wave [A 32x32 | G 32x32]; A rows = plain channels i*16 + fq*4 + r, G rows = k index ja*16 + fq*4 + r; columns = k index jb*16 + fr
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          red[wid * 2048 + (i * 16 + fq * 4 + r) * 32 + jb * 16 + fr] = accw[i][jb][r];
          red[wid * 2048 + 1024 + (i * 16 + fq * 4 + r) * 32 + jb * 16 + fr] = accg[i][jb][r];
        }
    __syncthreads();
    float* dst = p.partial + (long long)blockIdx.x * 2048;
    for (int i = tid; i < 2048; i += 256) dst[i] = (red[i] + red[2048 + i]) + (red[4096 + i] + red[6144 + i]);
  }
}

// ---- MODE 4 tail, part 1: AG[2][32][32] = sum of the workgroup slabs (fixed order: 8 row parts per column, LDS tree), 32 columns per block
__global__ __launch_bounds__(256) void stem_bwd_fold_kernel(const float* __restrict__ slab, int rows, float* __restrict__ ag) {
  __shared__ float red[8][32];
  const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + col;
  float s0 = 0.f, s1 = 0.f;
  int r = part;
  for (; r + 8 < rows; r += 16) {
    s0 += slab[(long long)r * 2048 + i];
    s1 += slab[(long long)(r + 8) * 2048 + i];
  }
  if (r < rows) s0 += slab[(long long)r * 2048 + i];
  red[part][col] = s0 + s1;
  __syncthreads();
  if (part == 0) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][col];
    ag[i] = t;
  }
}
// the BatchNorm-backward sums A implies: sums[c] = sum dy_c = A[c][27];  sums[32 + c] = sum dy_c * xhat_c = invstd_c * (<W[c], A[c]> - mean_c * A[c][27])
__global__ __launch_bounds__(64) void stem_bwd_sums_kernel(const float* __restrict__ ag, const bf16_t* __restrict__ w, const float* __restrict__ ss,
                                                           float* __restrict__ sums) {
  const int c = threadIdx.x;
  if (c >= 32) return;
  float dot = 0.f;
  for (int j = 0; j < 27; ++j) dot += s2f(w[c * 32 + j]) * ag[c * 32 + j];
  const float sdy = ag[c * 32 + 27];
  sums[c] = sdy;
  sums[32 + c] = ss[96 + c] * (dot - ss[64 + c] * sdy);
}

// ---- MODE 4 tail, part 2 (after the optional cross-rank average of `sums`): with m1 = sums[c] / count, m2 = sums[32 + c] / count,
//      B[j] = G[27][j] = sum im2col_j and sum xhat_c * im2col_j = invstd_c * ((W G)[c][j] - mean_c * B[j]):
//      dW[c][j] += scale_c * (A[c][j] - m1 * B[j] - m2 * invstd_c * ((W G)[c][j] - mean_c * B[j]));  dgamma += sums[32 + c];  dbeta += sums[c]
__global__ __launch_bounds__(1024) void stem_bwd_finish_kernel(const float* __restrict__ ag, const float* __restrict__ sums, const bf16_t* __restrict__ w,
                                                               const float* __restrict__ ss, float inv_count, float* __restrict__ dw,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float g_s[1024];
  __shared__ float w_s[1024];
  const int t = threadIdx.x;
  g_s[t] = ag[1024 + t];
  w_s[t] = s2f(w[t]);
  __syncthreads();
  const int c = t >> 5, j = t & 31;
  if (j < 27) {
    float wg = 0.f;
    for (int q = 0; q < 27; ++q) wg += w_s[c * 32 + q] * g_s[q * 32 + j];
    const float bj = g_s[27 * 32 + j];
    const float m1 = sums[c] * inv_count, m2 = sums[32 + c] * inv_count;
    const float sc = ss[c], mu = ss[64 + c], is = ss[96 + c];
    dw[t] += sc * (ag[t] - m1 * bj - m2 * is * (wg - mu * bj));
  }
  if (t < 32 && dgamma) {
    dbeta[t] += sums[t];
    dgamma[t] += sums[32 + t];
  }
}

// dW[32][32] += sum over the workgroup slabs (fixed order); block 0 also adds the BatchNorm parameter gradients
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ slab, int rows, float* __restrict__ dw,
                                                                const float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[8][32];
  const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + col;
  float s = 0.f;
  for (int r = part; r < rows; r += 8) s += slab[(long long)r * 1024 + i];
  red[part][col] = s;
  __syncthreads();
  if (part == 0) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][col];
    dw[i] += t;
  }
  if (blockIdx.x == 0 && threadIdx.x < 32 && dgamma) {
    dbeta[threadIdx.x] += sums[threadIdx.x];
    dgamma[threadIdx.x] += sums[32 + threadIdx.x];
  }
}

int stem_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}
// persistent grid = what is resident at once (a workgroup that has to queue behind the resident ones would run its tiles alone at the end):
// four workgroups per CU for the forward kernels, three for the backward ones (168 VGPRs)
int stem_grid(int ntiles, int per_cu = 4) {
  const int g = stem_cus() * per_cu;
  return ntiles < g ? ntiles : g;
}

int stem_check(const char* what, int n, int h, int w) {
  if (n <= 0 || h <= 0 || w <= 0 || h % TH != 0 || w % TW != 0)
    return fail(MI355DET_EINVAL, "%s: needs h %% 8 == 0 and w %% 32 == 0 (got %lld x %lld)", what, h, w);
  if ((long long)n * h * w * 12 >= 0x7FFFFFF0ll) return fail(MI355DET_EINVAL, "%s: image batch too large (32-bit byte offsets)", what);
  return 0;
}

StemParams stem_params(const float* img, const void* w, int n, int h, int wd) {
  StemParams p{};
  p.img = img;
  p.w = (const bf16_t*)w;
  p.n = n;
  p.H = h;
  p.W = wd;
  p.tiles_x = wd / TW;
  p.tiles_y = h / TH;
  p.ntiles = n * p.tiles_x * p.tiles_y;
  p.inv_count = 1.0f / (float)((long long)n * h * wd);
  p.rows_total = stem_grid(p.ntiles);
  return p;
}

template <int MODE>
int stem_launch(const char* what, const StemParams& p, hipStream_t st) {
  constexpr int lds_loop = 2 * IMG_BYTES + (MODE >= 3 ? 4 * DZ_WAVE : 0);
  constexpr int lds_end = MODE == 4 ? 4 * 2048 * 4 : MODE == 3 ? 4 * 1024 * 4 : 4 * 32 * 2 * 4;
  constexpr int lds = lds_loop > lds_end ? lds_loop : lds_end;
  hipLaunchKernelGGL(stem_kernel<MODE>, dim3(stem_grid(p.ntiles, MODE == 4 ? 2 : MODE >= 2 ? 3 : 4)), dim3(256), lds, st, p);
  return check_launch(what);
}

}  // namespace

extern "C" {

int mi355det_stem_rows(int32_t n, int32_t h, int32_t w) {
  if (n <= 0 || h <= 0 || w <= 0 || h % TH != 0 || w % TW != 0) return 0;
  return stem_grid(n * (h / TH) * (w / TW));
}

int mi355det_stem_fwd_stats(const float* img, const void* w, float* partial, int32_t n, int32_t h, int32_t wd, void* stream) {
  if (int e = stem_check("stem_fwd_stats", n, h, wd)) return e;
  if (!img || !w || !partial) return fail(MI355DET_EINVAL, "%s: null argument", "stem_fwd_stats");
  StemParams p = stem_params(img, w, n, h, wd);
  p.partial = partial;
  return stem_launch<0>("stem_fwd_stats", p, S(stream));
}

int mi355det_stem_fwd_apply(const float* img, const void* w, const float* scale_shift, float slope, void* a, int32_t a_ld, int32_t n, int32_t h,
                            int32_t wd, void* stream) {
  if (int e = stem_check("stem_fwd_apply", n, h, wd)) return e;
  if (!img || !w || !scale_shift || !a || a_ld < 32 || a_ld % 8) return fail(MI355DET_EINVAL, "%s: bad argument", "stem_fwd_apply");
  if (!(slope > 0.f && slope < 1.f)) return fail(MI355DET_EINVAL, "%s: LeakyReLU slope must be in (0, 1)", "stem_fwd_apply");
  if ((long long)n * h * wd * a_ld * 2 >= 0x7FFFFFF0ll) return fail(MI355DET_EINVAL, "%s: tensor too large (32-bit byte offsets)", "stem_fwd_apply");
  StemParams p = stem_params(img, w, n, h, wd);
  p.ss = scale_shift;
  p.slope = slope;
  p.a = (bf16_t*)a;
  p.ld = a_ld;
  return stem_launch<1>("stem_fwd_apply", p, S(stream));
}

int mi355det_stem_bwd_reduce(const float* img, const void* w, const float* scale_shift, float slope, const void* da, int32_t da_ld, float* partial,
                             int32_t n, int32_t h, int32_t wd, void* stream) {
  if (int e = stem_check("stem_bwd_reduce", n, h, wd)) return e;
  if (!img || !w || !scale_shift || !da || !partial || da_ld < 32 || da_ld % 8) return fail(MI355DET_EINVAL, "%s: bad argument", "stem_bwd_reduce");
  if ((long long)n * h * wd * da_ld * 2 >= 0x7FFFFFF0ll) return fail(MI355DET_EINVAL, "%s: tensor too large (32-bit byte offsets)", "stem_bwd_reduce");
  StemParams p = stem_params(img, w, n, h, wd);
  p.ss = scale_shift;
  p.slope = slope;
  p.da = (const bf16_t*)da;
  p.ld = da_ld;
  p.partial = partial;
  return stem_launch<2>("stem_bwd_reduce", p, S(stream));
}

int mi355det_stem_bwd_apply_wgrad(const float* img, const void* w, const float* scale_shift, const float* sums, float slope, const void* da,
                                  int32_t da_ld, float* slab, float* dw, float* dgamma, float* dbeta, int32_t n, int32_t h, int32_t wd,
                                  void* stream) {
  if (int e = stem_check("stem_bwd_apply_wgrad", n, h, wd)) return e;
  if (!img || !w || !scale_shift || !sums || !da || !slab || !dw || da_ld < 32 || da_ld % 8)
    return fail(MI355DET_EINVAL, "%s: bad argument", "stem_bwd_apply_wgrad");
  if ((long long)n * h * wd * da_ld * 2 >= 0x7FFFFFF0ll) return fail(MI355DET_EINVAL, "%s: tensor too large (32-bit byte offsets)", "stem_bwd_apply_wgrad");
  StemParams p = stem_params(img, w, n, h, wd);
  p.ss = scale_shift;
  p.sums = sums;
  p.slope = slope;
  p.da = (const bf16_t*)da;
  p.ld = da_ld;
  p.partial = slab;
  if (int e = stem_launch<3>("stem_bwd_apply_wgrad", p, S(stream))) return e;
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(32), dim3(256), 0, S(stream), slab, stem_grid(p.ntiles), dw, sums, dgamma, dbeta);
  return check_launch("stem_wgrad_reduce");
}

int mi355det_stem_bwd_fused(const float* img, const void* w, const float* scale_shift, float slope, const void* da, int32_t da_ld, float* slab,
                            float* ag, float* sums, int32_t n, int32_t h, int32_t wd, void* stream) {
  if (int e = stem_check("stem_bwd_fused", n, h, wd)) return e;
  if (!img || !w || !scale_shift || !da || !slab || !ag || !sums || da_ld < 32 || da_ld % 8) return fail(MI355DET_EINVAL, "%s: bad argument", "stem_bwd_fused");
  if (!(slope > 0.f && slope < 1.f)) return fail(MI355DET_EINVAL, "%s: LeakyReLU slope must be in (0, 1)", "stem_bwd_fused");
  if ((long long)n * h * wd * da_ld * 2 >= 0x7FFFFFF0ll) return fail(MI355DET_EINVAL, "%s: tensor too large (32-bit byte offsets)", "stem_bwd_fused");
  StemParams p = stem_params(img, w, n, h, wd);
  p.ss = scale_shift;
  p.slope = slope;
  p.da = (const bf16_t*)da;
  p.ld = da_ld;
  p.partial = slab;
  if (int e = stem_launch<4>("stem_bwd_fused", p, S(stream))) return e;
  hipLaunchKernelGGL(stem_bwd_fold_kernel, dim3(64), dim3(256), 0, S(stream), slab, stem_grid(p.ntiles), ag);
  hipLaunchKernelGGL(stem_bwd_sums_kernel, dim3(1), dim3(64), 0, S(stream), ag, (const bf16_t*)w, scale_shift, sums);
  return check_launch("stem_bwd_fold");
}

int mi355det_stem_bwd_finish(const void* w, const float* scale_shift, const float* ag, const float* sums, int64_t count, float* dw, float* dgamma,
                             float* dbeta, void* stream) {
  if (!w || !scale_shift || !ag || !sums || !dw || count <= 0 || (!dgamma) != (!dbeta)) return fail(MI355DET_EINVAL, "%s: bad argument", "stem_bwd_finish");
  hipLaunchKernelGGL(stem_bwd_finish_kernel, dim3(1), dim3(1024), 0, S(stream), ag, sums, (const bf16_t*)w, scale_shift, 1.0f / (float)count, dw, dgamma, dbeta);
  return check_launch("stem_bwd_finish");
}

}  // extern "C"
