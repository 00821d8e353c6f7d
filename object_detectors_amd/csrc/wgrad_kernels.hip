// Weight-gradient GEMM on MFMA (gfx950):
//
//   dW[co][n'] += sum_{m in pixel range} dY[m][co] * Xcol[m][n'],   n' = tap*Cin + ci
//
// Both operands are pixel-major in HBM (NHWC), i.e. the reduction index is the slow axis, so the MFMA
// fragments are produced with the gfx950 transposed LDS read (ds_read_b64_tr_b16): tiles are staged
// [pixel][channel] with LDS-DMA (coalesced 256-B rows) and read column-wise without any shuffle.
// 32-byte blocks of every LDS row are XOR-swizzled (on the source side) so the 8 rows a half-wave
// touches in one transposed read land on distinct banks.
// Split over the pixel axis; every split writes its partial tile to an fp32 slab (plain stores) and a second launch adds the slabs
// in a fixed order (bit-reproducible).  Two kernels: wgrad_kernel (128 x 128 tile, 4 waves, two workgroups per CU) and wgrad8_kernel
// (256 x 256 tile, 8 waves, the phase-staggered schedule of igemm8_kernel); same products in the same order for the same split count.
#include "common.h"

#include <cstdlib>
#include <unordered_map>

#include "tune_record.h"

using namespace mi355;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct FastDiv {
  unsigned mul, shift, d;
};

static FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.shift = l;
  f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv f) {
  const unsigned t = __umulhi(f.mul, n);
  return (t + n) >> f.shift;   // valid for n < 2^31
}

struct WgradParams {
  const bf16_t* x;      // [n,h,w,cin] pitch ldx
  const bf16_t* dy;     // [n,ho,wo,cout] pitch lddy
  float* dw;            // [cout][T*cin] fp32 (+=)
  float* slab;          // split-K partial tiles [split][tile][128*128] fp32, or nullptr (single split: direct +=)
  const bf16_t* zero;
  int M, Ho, Wo, H, W, ldx, lddy, Cin, Cout, stride, pad, ks, T, NP;   // NP = T*Cin
  int co_tiles, np_tiles, splits, chunk;                                // chunk = pixels per split (multiple of 64)
  int step_q, step_r;                                                   // 64 = step_q*Wo + step_r
  FastDiv dWo, dHo, dCin;
  int ablate;   // diagnostic (mi355det_debug_set(6, v), timing only, results are garbage): 1 = no X-tile LDS-DMA after the first k-step, 2 = no dY-tile LDS-DMA, 4 = no MFMAs
};

int g_wgrad_ablate = 0;    // diagnostic (mi355det_debug_set(6, v)): WgradParams::ablate
int g_wgrad_general = 0;   // diagnostic (mi355det_debug_set(1, v)): 1 = always the per-lane bookkeeping form (tests compare the two)
int g_wgrad_force_dbg = 0; // diagnostic (mi355det_debug_set(7, v)): split count (+ 65536: the 256 x 256 phase-staggered kernel) for every launch; 0 = tuned
int g_wgrad8_off = 0;      // diagnostic (mi355det_debug_set(8, v)): 1 = the tuner does not time the 256 x 256 phase-staggered kernel (same-box A/B of the two kernels)

namespace {

#define WG_BKP 64      // pixels per k-step
#define WG_TILE 128    // co and n' tile
#define WG_ROWB 256    // LDS row bytes (128 bf16)

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}

// LDS-DMA through a buffer descriptor (per-lane 32-bit byte offset + SCALAR offset); a lane whose offset is out of range
// gets zeros written to LDS.  Issued as inline assembly ON PURPOSE: for the builtin the compiler's wait-count pass assumes every
// later ds_read may alias the DMA's LDS destination and puts s_waitcnt vmcnt(0) in front of the next fragment read, which
// serialises the prefetch of stage s+1 with the reads of stage s (no load/compute overlap inside a workgroup).  The kernels
// order DMA and reads themselves (counted vmcnt + barrier), so the compiler must not know.
typedef __attribute__((ext_vector_type(4))) int srd_t;
__device__ __forceinline__ srd_t make_srd(const void* base, unsigned num_records) {
  const unsigned long long a = (unsigned long long)base;
  srd_t r;
  r[0] = (int)(unsigned)a;
  r[1] = (int)((unsigned)(a >> 32) & 0xFFFFu);
  r[2] = (int)num_records;
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void bufld16(srd_t rsrc, const void* lds_dst_uniform, int voffset, int soffset) {
  const unsigned lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)lds_dst_uniform;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}
#define OOB_VOFF ((int)0x80000000)

__device__ __forceinline__ int rowf(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

__device__ __forceinline__ st16x4_t lds_tr(const char* p) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
  return __builtin_bit_cast(st16x4_t, v);
}

// CI x CJ = 16-wide fragments per wave along cout / n': 4 x 4 is the full 128x128 tile; layers whose Cout or k*k*Cin is 32 / 64 (the
// first Darknet layers) use 1 or 2 so that the four waves split the REAL channels instead of multiplying zero fragments
// (50-94 % of the MFMA issue with the full tile).  Staging and the slab layout are unchanged.
// GRP4: the (n, ho, wo) bookkeeping is kept per LDS-DMA piece (4 consecutive pixels) in SCALAR registers; when the piece lies in one
// image row (always, for map widths that are multiples of 4) the dY offsets are loop constants and the X offsets a lane constant
// plus a scalar: the per-step vector work drops from ~146 to ~25 VALU instructions (the general form spends more issue cycles on
// addresses than on the 32 MFMAs of a step).  A piece that straddles a row end adds a per-lane select of the next row's scalars.
template <int CI, int CJ, int GRP4>   // 0: per-lane bookkeeping, 1: scalar, pieces never straddle (Wo % 4 == 0), 2: scalar with straddling pieces,
                                      // 3: scalar, one tracker per wave (Wo % 16 == 0)
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
  constexpr int STAGE = 2 * WG_BKP * WG_ROWB;   // dy tile + x tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid & 1, wc = wid >> 1;   // wave tile: co rows wr*64.., n' cols wc*64..

  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tiles = p.co_tiles * p.np_tiles;
  const int split = bid / tiles, tile = bid - split * tiles;
  const int ct = tile % p.co_tiles, nt = tile / p.co_tiles;
  const int co0 = ct * WG_TILE, np0 = nt * WG_TILE;
  const int mA = split * p.chunk, mB = min(p.M, mA + p.chunk);
  if (mA >= mB) return;

  // staging roles: one LDS-DMA wave instruction = 4 rows x 256 B; 16 instr per tile per k-step, 4 per wave
  const int lrow = lane >> 4, cpos = lane & 15;
  // this lane's column block for the two possible swizzle phases (row bit 3 = 0 / 1)
  int x_off[2], x_tapdy[2], x_tapdx[2], dy_off[2];
  bool x_ok[2];
#pragma unroll
  for (int ph = 0; ph < 2; ++ph) {
    const int f = (lrow & 3) | (ph << 2);
    const int gchunk = (((cpos >> 1) ^ f) << 1) | (cpos & 1);
    dy_off[ph] = co0 + gchunk * 8;
    const int np = np0 + gchunk * 8;
    x_ok[ph] = np < p.NP;
    const int t = x_ok[ph] ? (int)fdiv((unsigned)np, p.dCin) : 0;
    const int ci = np - t * p.Cin;
    const int kh = t / p.ks, kw = t - kh * p.ks;
    x_tapdy[ph] = kh - p.pad;
    x_tapdx[ph] = kw - p.pad;
    x_off[ph] = ci;
  }
  const bool dy_ok0 = dy_off[0] < p.Cout, dy_ok1 = dy_off[1] < p.Cout;

  // each lane stages 4 fixed tile rows; their output pixels advance by 64 per k-step, so (n, ho, wo) is
  // kept incrementally (one conditional wrap per axis) instead of two divisions per row per step
  int r_n[4], r_ho[4], r_wo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (GRP4) break;
    const unsigned um = (unsigned)(mA + (wid * 4 + i) * 4 + lrow);
    const unsigned q1 = fdiv(um, p.dWo);
    r_wo[i] = (int)(um - q1 * p.Wo);
    const unsigned n = fdiv(q1, p.dHo);
    r_ho[i] = (int)(q1 - n * p.Ho);
    r_n[i] = (int)n;
  }

  // GRP4 state: per LDS-DMA piece (4 consecutive pixels) the SCALED scalar coordinates of its first pixel (hs = ho*stride,
  // ws = wo*stride) and the byte offset sb of that input pixel, all advanced by constants (no multiply in the loop); per lane and
  // swizzle phase the constant parts of the X offset and of the bounds tests
  int g_hs[4], g_ws[4], g_sb[4], dyv[4], lc[2], ty[2], cx[2];
  srd_t rsrc_x;
  const int WoS = p.Wo * p.stride, HoS = p.Ho * p.stride;
  const int c_ws = p.step_r * p.stride, c_hs = p.step_q * p.stride;
  const int c_sb = (c_hs * p.W + c_ws) * p.ldx * 2;                 // 64 pixels ahead without wrapping
  const int c_row = (p.stride * p.W - WoS) * p.ldx * 2;             // extra bytes when wo wraps into the next output row
  const int c_img = (p.H - HoS) * p.W * p.ldx * 2;                  // extra bytes when ho wraps into the next image
  const int c_piece = 4 * p.stride * p.ldx * 2;                     // bytes between the pieces of one wave inside a row (GRP4 == 3)
  if constexpr (GRP4 != 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned um = (unsigned)(mA + (wid * 4 + i) * 4);
      const unsigned q1 = fdiv(um, p.dWo);
      const int wo = (int)(um - q1 * p.Wo);
      const unsigned n = fdiv(q1, p.dHo);
      const int ho = (int)(q1 - n * p.Ho);
      g_hs[i] = ho * p.stride;
      g_ws[i] = wo * p.stride;
      g_sb[i] = (((int)n * p.H + g_hs[i]) * p.W + g_ws[i]) * p.ldx * 2;
      const int ph = (i >> 1) & 1;
      dyv[i] = (ph ? dy_ok1 : dy_ok0) ? (((wid * 4 + i) * 4 + lrow) * p.lddy + dy_off[ph]) * 2 : OOB_VOFF;
    }
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      // address = x - bias + [((n*H + ho*s)*W + wo*s) * ldx] + [((kh*W + lrow*s + kw) * ldx + ci)], bias = (pad*W + pad) * ldx
      lc[ph] = (((x_tapdy[ph] + p.pad) * p.W + lrow * p.stride + x_tapdx[ph] + p.pad) * p.ldx + x_off[ph]) * 2;
      ty[ph] = x_ok[ph] ? x_tapdy[ph] : -(1 << 24);
      cx[ph] = lrow * p.stride + x_tapdx[ph];
    }
    rsrc_x = make_srd(p.x - (p.pad * p.W + p.pad) * p.ldx, 0x7FFFFFF0u);
  }
  auto stage_grp4 = [&](int m_base, int buf) {
    char* sd = smem + buf * STAGE;
    char* sx = sd + WG_BKP * WG_ROWB;
    const int rem = mB - m_base;   // rows past it read zeros through the descriptor's range check
    const srd_t rsrc_dy = make_srd(p.dy + (long long)m_base * p.lddy, (unsigned)(rem * p.lddy * 2));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int instr = wid * 4 + i;
      const int ph = (i >> 1) & 1;
      if (!(p.ablate & 2) || m_base == mA) bufld16(rsrc_dy, sd + instr * 1024, dyv[i], 0);
      if (GRP4 == 3) {   // Wo % 16 == 0: the wave's four pieces are 16 consecutive pixels of ONE image row, one tracker serves them
        const int hs = g_hs[0], ws = g_ws[0] + i * 4 * p.stride;
        const bool ok = instr * 4 < rem && (unsigned)(hs + ty[ph]) < (unsigned)p.H && (unsigned)(ws + cx[ph]) < (unsigned)p.W;
        if (!(p.ablate & 1) || m_base == mA) bufld16(rsrc_x, sx + instr * 1024, ok ? lc[ph] : OOB_VOFF, g_sb[0] + i * c_piece);
        if (i < 3) continue;
      }
      const int ti = GRP4 == 3 ? 0 : i;
      const int hs = g_hs[ti], ws = g_ws[ti];
      if (GRP4 == 3) {
      } else if (GRP4 == 1 || (ws + 4 * p.stride <= WoS && instr * 4 + 4 <= rem)) {   // wave-uniform: the piece's four pixels lie in one image row
        const bool ok = instr * 4 < rem && (unsigned)(hs + ty[ph]) < (unsigned)p.H && (unsigned)(ws + cx[ph]) < (unsigned)p.W;
        if (!(p.ablate & 1) || m_base == mA) bufld16(rsrc_x, sx + instr * 1024, ok ? lc[ph] : OOB_VOFF, g_sb[i]);
      } else {
        // the piece straddles a row end (map widths that are not multiples of 4) or the end of the pixel range: lanes past the
        // row end move to the next row / image by a scalar byte delta
        const bool last_row = hs + p.stride == HoS;
        const int hs1 = last_row ? 0 : hs + p.stride;
        const int ex1 = c_row + (last_row ? c_img : 0);
        const bool wr = ws + lrow * p.stride >= WoS;
        const int iy = (wr ? hs1 : hs) + ty[ph];
        const int ix = ws + cx[ph] - (wr ? WoS : 0);
        const bool ok = instr * 4 + lrow < rem && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        bufld16(rsrc_x, sx + instr * 1024, ok ? lc[ph] + (wr ? ex1 : 0) : OOB_VOFF, g_sb[i]);
      }
      g_ws[ti] += c_ws;
      g_hs[ti] += c_hs;
      g_sb[ti] += c_sb;
      if (g_ws[ti] >= WoS) {
        g_ws[ti] -= WoS;
        g_hs[ti] += p.stride;
        g_sb[ti] += c_row;
      }
      if (g_hs[ti] >= HoS) {   // a single wrap: the host only picks this form when 64 pixels span at most Ho - 1 rows
        g_hs[ti] -= HoS;
        g_sb[ti] += c_img;
      }
    }
  };

  auto stage_any = [&](int m_base, int buf) {
    char* sd = smem + buf * STAGE;
    char* sx = sd + WG_BKP * WG_ROWB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int instr = wid * 4 + i;
      const int row = instr * 4 + lrow;
      const int ph = (row >> 3) & 1;
      const int m = m_base + row;
      const bool mv = m < mB;
      // dY tile: row m, channels co0 + swizzled chunk
      const bf16_t* sdy = (mv && (ph ? dy_ok1 : dy_ok0)) ? p.dy + (long long)m * p.lddy + (ph ? dy_off[1] : dy_off[0]) : p.zero;
      glds16(sdy, sd + instr * 1024);
      // X tile: gather the input pixel of tap(t) for output pixel m
      const int iy = r_ho[i] * p.stride + (ph ? x_tapdy[1] : x_tapdy[0]);
      const int ix = r_wo[i] * p.stride + (ph ? x_tapdx[1] : x_tapdx[0]);
      const bool ok = mv && (ph ? x_ok[1] : x_ok[0]) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const bf16_t* sxp = ok ? p.x + ((long long)(r_n[i] * p.H + iy) * p.W + ix) * p.ldx + (ph ? x_off[1] : x_off[0]) : p.zero;
      glds16(sxp, sx + instr * 1024);
      // advance this row by 64 pixels
      r_wo[i] += p.step_r;
      r_ho[i] += p.step_q;
      if (r_wo[i] >= p.Wo) {
        r_wo[i] -= p.Wo;
        r_ho[i] += 1;
      }
      while (r_ho[i] >= p.Ho) {
        r_ho[i] -= p.Ho;
        r_n[i] += 1;
      }
    }
  };

  auto stage = [&](int m_base, int buf) {
    if constexpr (GRP4 != 0)
      stage_grp4(m_base, buf);
    else
      stage_any(m_base, buf);
  };

  f32x4_t acc[CI][CJ];
#pragma unroll
  for (int i = 0; i < CI; ++i)
#pragma unroll
    for (int j = 0; j < CJ; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

  const int ksteps = (mB - mA + WG_BKP - 1) / WG_BKP;
  stage(mA, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // transposed-read lane roles: 16-lane group g covers pixels 8g..8g+7 of a 32-pixel sub-step;
  // lane 4q+pp supplies row q, columns 4pp..4pp+3 of the 16-column block
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  for (int s = 0; s < ksteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < ksteps) stage(mA + (s + 1) * WG_BKP, buf ^ 1);
    const char* sd = smem + buf * STAGE;
    const char* sx = sd + WG_BKP * WG_ROWB;
#pragma unroll
    for (int ks = 0; ks < WG_BKP / 32; ++ks) {
      st16x8_t af[CI], bfr[CJ];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = ks * 32 + 8 * g + 4 * h + q;
        const int f = rowf(row);
#pragma unroll
        for (int i = 0; i < CI; ++i) {
          const int blk = wr * CI + i;   // 32-byte block index of this 16-channel group
          const st16x4_t v = lds_tr(sd + row * WG_ROWB + ((blk ^ f) << 5) + pp * 8);
          af[i][4 * h + 0] = v[0];
          af[i][4 * h + 1] = v[1];
          af[i][4 * h + 2] = v[2];
          af[i][4 * h + 3] = v[3];
        }
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
          const int blk = wc * CJ + j;
          const st16x4_t v = lds_tr(sx + row * WG_ROWB + ((blk ^ f) << 5) + pp * 8);
          bfr[j][4 * h + 0] = v[0];
          bfr[j][4 * h + 1] = v[1];
          bfr[j][4 * h + 2] = v[2];
          bfr[j][4 * h + 3] = v[3];
        }
      }
      if (!(p.ablate & 4)) {
#pragma unroll
        for (int i = 0; i < CI; ++i)
#pragma unroll
          for (int j = 0; j < CJ; ++j) acc[i][j] = MI355_MFMA_16x16x32(af[i], bfr[j], acc[i][j]);
      } else {
        acc[0][0][0] += (float)af[0][0] + (float)bfr[0][0];      // keep the fragment reads alive
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // D[row = co][col = n']: lane holds rows fq*4+r, column fr
  // (measured and removed, round 4: the X fragment as the row operand, so that a lane holds four consecutive n' and the slab leaves in 16-byte stores -
  //  bit-identical, equal alone, SLOWER in the step: RetinaNet-R101-LVIS 35.6 -> 36.6 ms, profiles/r04_ab_results.md 11)
  const int fr = lane & 15, fq = lane >> 4;
  if (p.slab) {
    // split-K: plain stores of the partial tile (6 TB/s class) instead of fp32 atomics (1.3 TB/s class, contended);
    // wgrad_reduce_kernel adds the slabs into dW in a fixed order (deterministic)
    float* dst = p.slab + ((size_t)split * tiles + tile) * (WG_TILE * WG_TILE);
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < CJ; ++j)
          dst[((wr * CI + i) * 16 + fq * 4 + r) * WG_TILE + (wc * CJ + j) * 16 + fr] = acc[i][j][r];
    return;
  }
#pragma unroll
  for (int i = 0; i < CI; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + (wr * CI + i) * 16 + fq * 4 + r;
      if (co >= p.Cout) continue;
#pragma unroll
      for (int j = 0; j < CJ; ++j) {
        const int np = np0 + (wc * CJ + j) * 16 + fr;
        if (np < p.NP) p.dw[(long long)co * p.NP + np] += acc[i][j][r];   // single split: this block owns the tile
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Phase-staggered form (round 4): 256 co x 256 n' x 64 pixels per workgroup, 8 waves = 2 (n' halves, 128 each) x 4 (co quarters, 64
// each), one workgroup per CU (128 KB LDS) - the schedule of igemm8_kernel (igemm8_kernels.hip) on transposed operands.
//
// Why: the 128 x 128 kernel above is LDS-bound.  Per 64-pixel k-step its four 64 x 64 wave tiles read 64 KB of fragments and the LDS-DMA
// writes 32 KB for 128 MFMAs: 768 LDS cycles (128 B / clk) against 512 cycles of matrix pipe, and every wave runs DMA issue -> full
// vmcnt(0) -> barrier -> reads -> MFMAs in the same order (profiles/r04_wgrad_ablations.txt: without ANY operand traffic +18-24 %, without
// MFMAs +16-24 %: the k-step structure is the bound).  Here a wave tile is 128 x 64 (25 % fewer fragment bytes and 25 % fewer DMA bytes
// per MFMA), the DMA is waited for with counted vmcnt, and the two waves of a SIMD run one barrier apart, so one is in its MFMA segment
// while its partner issues DMA and waits.
//
// LDS: two k-step buffers of [X0 | X1 | D0 | D1], each a [64 pixels][256 B] half-tile in the layout of the kernel above (32-byte blocks
// XOR-swizzled with rowf(pixel row), fragments by ds_read_b64_tr_b16).  Half h of the X tile holds the n' columns every wave multiplies in
// its sub-phase h (n' = np0 + wm * 128 + h * 64 + 0..63 for wm = 0, 1: logical block b = wm * 4 + jj), half h of the dY tile the co
// columns co0 + wn * 64 + h * 32 + 0..31 (b = wn * 2 + ii), so a half is free for re-staging as soon as its sub-phase has been read.
// Phases, counted waits and the WAR / RAW argument are igemm8_kernel's (X = the 128-wide operand, D = its "W"):
//   phase 1: MFMA (D0, X0) + reads D0k1 D1k0 D1k1      issues D1 of k-step t+1
//   phase 2: MFMA (D1, X0) + reads X1                  issues X1 of t+1
//   phase 3: MFMA (D1, X1)                             issues X0 of t+2
//   phase 4: MFMA (D0, X1) + reads X0, D0k0 of t+1     issues D0 of t+2
// A wave stages pixel rows wid * 8 .. + 7 of every half-tile (two 4-pixel pieces), so rowf() of its rows is a lane constant, and one pair
// of scalar (n, ho, wo) trackers serves all four half-tiles.  STRADDLE = false requires Wo % 4 == 0 (a piece never crosses a row end).
// The partial tiles go to the slabs in the 128 x 128 layout of the kernel above: wgrad_reduce*_kernel is shared.
constexpr int W8_HALF = WG_BKP * WG_ROWB;      // 16 KB
constexpr int W8_STAGE = 4 * W8_HALF;          // 64 KB: one k-step
constexpr int W8_LDS = 2 * W8_STAGE;           // 128 KB
constexpr int W8_TILE = 256;

template <int N>
__device__ __forceinline__ void w8_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void w8_bar() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <bool STRADDLE>      // true: map widths that are not multiples of 4 - a 4-pixel piece may cross a row end (per-lane select of the next row's scalars)
__global__ __launch_bounds__(512, 2) void wgrad8_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;     // waves 0-3 = n' half 0, waves 4-7 = n' half 1 (their SIMD partners)

  int bid = blockIdx.x;
  {
    const int nblk = gridDim.x, q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int co_tiles8 = (p.Cout + W8_TILE - 1) / W8_TILE, np_tiles8 = (p.NP + W8_TILE - 1) / W8_TILE;
  const int tiles8 = co_tiles8 * np_tiles8;
  const int split = bid / tiles8, tile = bid - split * tiles8;
  const int CT = tile % co_tiles8, NT = tile / co_tiles8;
  const int co0 = CT * W8_TILE, np0 = NT * W8_TILE;
  const int mA = split * p.chunk, mB = min(p.M, mA + p.chunk);
  if (mA >= mB) return;
  const int mlen = mB - mA;

  // ---- staging roles: a piece = 4 pixel rows x 256 B; this wave's pieces of every half-tile are rows wid * 8 + i * 4 + lrow
  const int lrow = lane >> 4, cpos = lane & 15;
  const int fst = lrow | ((wid & 1) << 2);                       // rowf() of the rows this lane stages
  const int gchunk = (((cpos >> 1) ^ fst) << 1) | (cpos & 1);    // logical 16-byte chunk that lands at this lane's LDS position
  int dyv[2][2], lc[2], ty[2], cx[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int co = co0 + (gchunk >> 2) * 64 + h * 32 + (gchunk & 3) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) dyv[h][i] = co < p.Cout ? ((wid * 8 + i * 4 + lrow) * p.lddy + co) * 2 : OOB_VOFF;
    const int np = np0 + (gchunk >> 3) * 128 + h * 64 + (gchunk & 7) * 8;
    const bool x_ok = np < p.NP;
    const int t = x_ok ? (int)fdiv((unsigned)np, p.dCin) : 0;
    const int ci = np - t * p.Cin;
    const int kh = t / p.ks, kw = t - kh * p.ks;
    // address = x - bias + [((n*H + ho*s)*W + wo*s) * ldx] + [((kh*W + lrow*s + kw) * ldx + ci)], bias = (pad*W + pad) * ldx
    lc[h] = ((kh * p.W + lrow * p.stride + kw) * p.ldx + ci) * 2;
    ty[h] = x_ok ? kh - p.pad : -(1 << 24);
    cx[h] = lrow * p.stride + kw - p.pad;
  }
  const srd_t rsrc_x = make_srd(p.x - (p.pad * p.W + p.pad) * p.ldx, 0x7FFFFFF0u);
  const srd_t rsrc_dy = make_srd(p.dy + (long long)mA * p.lddy, (unsigned)((long long)mlen * p.lddy * 2));      // rows past the range read zeros

  // scalar (ho * stride, wo * stride, byte offset) of the first pixel of the wave's two pieces, advanced by 64 pixels per k-step
  const int WoS = p.Wo * p.stride, HoS = p.Ho * p.stride;
  const int c_ws = p.step_r * p.stride, c_hs = p.step_q * p.stride;
  const int c_sb = (c_hs * p.W + c_ws) * p.ldx * 2;
  const int c_row = (p.stride * p.W - WoS) * p.ldx * 2;
  const int c_img = (p.H - HoS) * p.W * p.ldx * 2;
  int g_hs[2], g_ws[2], g_sb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned um = (unsigned)(mA + wid * 8 + i * 4);
    const unsigned q1 = fdiv(um, p.dWo);
    const int wo = (int)(um - q1 * p.Wo);
    const unsigned n = fdiv(q1, p.dHo);
    const int ho = (int)(q1 - n * p.Ho);
    g_hs[i] = ho * p.stride;
    g_ws[i] = wo * p.stride;
    g_sb[i] = (((int)n * p.H + g_hs[i]) * p.W + g_ws[i]) * p.ldx * 2;
  }
  struct Slot { int hs[2], ws[2], sb[2], dyoff, rem; };
  int pf = 0;                                   // pixel offset (from mA) of the next k-step to set up
  auto next_slot = [&]() {
    Slot s;
    s.rem = mlen - pf;                          // <= 0: a dead k-step stages zeros (the counted waits stay uniform)
    s.dyoff = pf * p.lddy * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      s.hs[i] = g_hs[i];
      s.ws[i] = g_ws[i];
      s.sb[i] = g_sb[i];
      g_ws[i] += c_ws;
      g_hs[i] += c_hs;
      g_sb[i] += c_sb;
      if (g_ws[i] >= WoS) {
        g_ws[i] -= WoS;
        g_hs[i] += p.stride;
        g_sb[i] += c_row;
      }
      if (g_hs[i] >= HoS) {                      // a single wrap: the host picks this kernel only when 64 pixels span at most Ho - 1 rows
        g_hs[i] -= HoS;
        g_sb[i] += c_img;
      }
    }
    pf += WG_BKP;
    return s;
  };
  auto issue_x = [&](const Slot& s, int h, int buf) {
    char* dst = smem + buf * W8_STAGE + h * W8_HALF + (2 * wid) * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row0 = wid * 8 + i * 4;
      if (!STRADDLE || (s.ws[i] + 4 * p.stride <= WoS && row0 + 4 <= s.rem) || row0 >= s.rem) {      // wave-uniform: the piece lies in one image row (or is dead)
        const bool ok = row0 < s.rem && (unsigned)(s.hs[i] + ty[h]) < (unsigned)p.H && (unsigned)(s.ws[i] + cx[h]) < (unsigned)p.W;
        bufld16(rsrc_x, dst + i * 1024, ok ? lc[h] : OOB_VOFF, s.sb[i]);
      } else {
        // the piece crosses a row end or the end of the pixel range: lanes past the row end move to the next row / image by a scalar byte delta
        const bool last_row = s.hs[i] + p.stride == HoS;
        const int hs1 = last_row ? 0 : s.hs[i] + p.stride;
        const int ex1 = c_row + (last_row ? c_img : 0);
        const bool wr = s.ws[i] + lrow * p.stride >= WoS;
        const int iy = (wr ? hs1 : s.hs[i]) + ty[h];
        const int ix = s.ws[i] + cx[h] - (wr ? WoS : 0);
        const bool ok = row0 + lrow < s.rem && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        bufld16(rsrc_x, dst + i * 1024, ok ? lc[h] + (wr ? ex1 : 0) : OOB_VOFF, s.sb[i]);
      }
    }
  };
  auto issue_d = [&](const Slot& s, int h, int buf) {
    char* dst = smem + buf * W8_STAGE + (2 + h) * W8_HALF + (2 * wid) * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) bufld16(rsrc_dy, dst + i * 1024, s.rem > 0 ? dyv[h][i] + s.dyoff : OOB_VOFF, 0);
  };

  // ---- fragment read addresses: 16-lane group g covers pixels 8g..8g+7 of a 32-pixel sub-step, lane 4q+pp supplies row q, columns
  //      4pp..4pp+3 of the 16-column block; rowf() of those rows is q | (g & 1) << 2 (immediates: + h * W8_HALF + ks * 32 rows + hh * 4 rows)
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int frd = q | ((g & 1) << 2);
  int xrd[4], drd[2];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) xrd[jj] = (8 * g + q) * WG_ROWB + pp * 8 + (((wm * 4 + jj) ^ frd) << 5);
#pragma unroll
  for (int ii = 0; ii < 2; ++ii) drd[ii] = 2 * W8_HALF + (8 * g + q) * WG_ROWB + pp * 8 + (((wn * 2 + ii) ^ frd) << 5);

  f32x4_t acc[4][8];      // [co fragment hw * 2 + ii][n' fragment hx * 4 + jj]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

  const int ksteps = (mlen + WG_BKP - 1) / WG_BKP;
  // ---- prologue: X0(0) D0(0) D1(0) X1(0) X0(1) D0(1), the steady-state issue order
  Slot s1 = next_slot();
  issue_x(s1, 0, 0);
  issue_d(s1, 0, 0);
  issue_d(s1, 1, 0);
  issue_x(s1, 1, 0);
  s1 = next_slot();
  issue_x(s1, 0, 1);
  issue_d(s1, 0, 1);
  Slot s2 = next_slot();
  w8_wait_vm<6>();
  w8_bar();
  if (wm == 1) w8_bar();          // the second n' half runs one barrier behind

  st16x8_t xf[2][4][2], df[2][2][2];      // [half][fragment][k-substep]
  auto rdd = [&](int h, int ks, bool nxt) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const st16x4_t v = lds_tr(smem + (nxt ? (drd[ii] ^ W8_STAGE) : drd[ii]) + h * W8_HALF + (ks * 32 + hh * 4) * WG_ROWB);
        df[h][ii][ks][4 * hh + 0] = v[0];
        df[h][ii][ks][4 * hh + 1] = v[1];
        df[h][ii][ks][4 * hh + 2] = v[2];
        df[h][ii][ks][4 * hh + 3] = v[3];
      }
  };
  auto rdx = [&](int h, int ks, bool nxt) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const st16x4_t v = lds_tr(smem + (nxt ? (xrd[jj] ^ W8_STAGE) : xrd[jj]) + h * W8_HALF + (ks * 32 + hh * 4) * WG_ROWB);
        xf[h][jj][ks][4 * hh + 0] = v[0];
        xf[h][jj][ks][4 * hh + 1] = v[1];
        xf[h][jj][ks][4 * hh + 2] = v[2];
        xf[h][jj][ks][4 * hh + 3] = v[3];
      }
  };
  auto mf = [&](int hd, int hx, int ks) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        acc[hd * 2 + ii][hx * 4 + jj] = MI355_MFMA_16x16x32(df[hd][ii][ks], xf[hx][jj][ks], acc[hd * 2 + ii][hx * 4 + jj]);
  };
#define W8_INTERLEAVE(nm, nr)                                                      \
  do {                                                                            \
    _Pragma("unroll") for (int q_ = 0; q_ < ((nr) < (nm) ? (nr) : (nm)); ++q_) {   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
    }                                                                             \
    if constexpr ((nm) > (nr)) __builtin_amdgcn_sched_group_barrier(0x008, (nm) > (nr) ? (nm) - (nr) : 0, 0); \
    if constexpr ((nr) > (nm)) __builtin_amdgcn_sched_group_barrier(0x100, (nr) > (nm) ? (nr) - (nm) : 0, 0); \
  } while (0)

  rdd(0, 0, false);
  rdx(0, 0, false);
  rdx(0, 1, false);
  int buf = 0;
  for (int t = 0; t < ksteps; ++t) {
    // ---- phase 1
    issue_d(s1, 1, buf ^ 1);
    w8_wait_vm<8>();
    w8_bar();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    rdd(0, 1, false);
    rdd(1, 0, false);
    mf(0, 0, 0);
    W8_INTERLEAVE(8, 8);
    rdd(1, 1, false);
    mf(0, 0, 1);
    W8_INTERLEAVE(8, 4);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    w8_wait_vm<6>();                // the partner half reads, right after this barrier, fragments of tiles this wave helped to stage
    w8_bar();
    // ---- phase 2
    issue_x(s1, 1, buf ^ 1);
    w8_wait_vm<8>();
    w8_bar();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    rdx(1, 0, false);
    mf(1, 0, 0);
    W8_INTERLEAVE(8, 8);
    rdx(1, 1, false);
    mf(1, 0, 1);
    W8_INTERLEAVE(8, 8);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    w8_wait_vm<6>();
    w8_bar();
    // ---- phase 3
    issue_x(s2, 0, buf);
    w8_wait_vm<8>();
    w8_bar();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mf(1, 1, 0);
    mf(1, 1, 1);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    w8_wait_vm<6>();
    w8_bar();
    // ---- phase 4 (its wait publishes D0 / X0 of the next k-step: their fragments are fetched here, from the other buffer)
    issue_d(s2, 0, buf);
    w8_wait_vm<8>();
    w8_bar();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    rdx(0, 0, true);
    mf(0, 1, 0);
    W8_INTERLEAVE(8, 8);
    rdd(0, 0, true);
    rdx(0, 1, true);
    mf(0, 1, 1);
    W8_INTERLEAVE(8, 12);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    w8_wait_vm<6>();
    w8_bar();
    s1 = s2;
    s2 = next_slot();
    buf ^= 1;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) xrd[jj] ^= W8_STAGE;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) drd[ii] ^= W8_STAGE;
  }
#undef W8_INTERLEAVE
  if (wm == 0) w8_bar();          // re-align the two halves
  w8_wait_vm<0>();                // the zero-fill pieces of the dead k-steps

  // D[row = co][col = n']: lane holds rows fq*4+r, column fr of every 16 x 16 fragment
  const int fr = lane & 15, fq = lane >> 4;
  const int nt128 = NT * 2 + wm;                                   // the wave's 128 n' columns are one 128-tile of the slab layout
  if (nt128 >= p.np_tiles) return;
  if (p.slab) {
    const int tiles = p.co_tiles * p.np_tiles;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int col = wn * 64 + (a >> 1) * 32 + (a & 1) * 16;      // co offset of fragment a inside the 256-tile
      const int ct128 = CT * 2 + (col >> 7);
      if (ct128 >= p.co_tiles) continue;
      float* dst = p.slab + ((size_t)split * tiles + (size_t)nt128 * p.co_tiles + ct128) * (WG_TILE * WG_TILE);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < 8; ++b) dst[((col & 127) + fq * 4 + r) * WG_TILE + (b >> 2) * 64 + (b & 3) * 16 + fr] = acc[a][b][r];
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + wn * 64 + (a >> 1) * 32 + (a & 1) * 16 + fq * 4 + r;
      if (co >= p.Cout) continue;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int np = np0 + wm * 128 + (b >> 2) * 64 + (b & 3) * 16 + fr;
        if (np < p.NP) p.dw[(long long)co * p.NP + np] += acc[a][b][r];      // single split: this workgroup owns the tile
      }
    }
}

// dW[co][np] += sum_split slab[split][tile(co,np)][...]   (float4 per thread, coalesced over np).
// blockIdx.y strides over groups of 8 splits (8 independent loads in flight per lane); a single group adds
// into dW directly (deterministic), several groups combine with a few fp32 atomics.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int NP, int co_tiles,
                                                           int np_tiles, int splits) {
  const int tiles = co_tiles * np_tiles;
  const long long total = (long long)Cout * (NP / 4);
  const size_t sstride = (size_t)tiles * (WG_TILE * WG_TILE);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int co = (int)(i / (NP / 4)), np = (int)(i - (long long)co * (NP / 4)) * 4;
    const int ct = co / WG_TILE, nt = np / WG_TILE;
    const int tile = nt * co_tiles + ct;
    const float* src = slab + (size_t)tile * (WG_TILE * WG_TILE) + (co - ct * WG_TILE) * WG_TILE + (np - nt * WG_TILE);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = blockIdx.y * 8; s0 < splits; s0 += gridDim.y * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = s0 + u < splits ? *(const float4*)(src + (size_t)(s0 + u) * sstride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w;
      }
    }
    float* d = dw + (long long)co * NP + np;
    if (gridDim.y == 1) {
      float4 o = *(const float4*)d;
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      *(float4*)d = o;
    } else {
      atomicAdd(d + 0, a.x); atomicAdd(d + 1, a.y); atomicAdd(d + 2, a.z); atomicAdd(d + 3, a.w);
    }
  }
}

// Same reduction with the splits spread over 8 lanes of the workgroup (32 float4 columns x 8 split lanes, LDS tree at the end):
// the column-per-thread form above walks the splits serially, 8 loads at a time, and is latency-bound (18 us average, 1.4 ms per
// step); here every thread has ceil(splits / 8) independent loads.  Fixed association order: deterministic.
__global__ __launch_bounds__(256) void wgrad_reduce8_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int NP, int co_tiles,
                                                            int np_tiles, int splits) {
  __shared__ float4 red[8][32];
  const int tiles = co_tiles * np_tiles;
  const long long total = (long long)Cout * (NP / 4);
  const size_t sstride = (size_t)tiles * (WG_TILE * WG_TILE);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (long long i0 = (long long)blockIdx.x * 32; i0 < total; i0 += (long long)gridDim.x * 32) {
    const long long i = i0 + tx;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    float* d = nullptr;
    if (i < total) {
      const int co = (int)(i / (NP / 4)), np = (int)(i - (long long)co * (NP / 4)) * 4;
      const int ct = co / WG_TILE, nt = np / WG_TILE;
      const int tile = nt * co_tiles + ct;
      const float* src = slab + (size_t)tile * (WG_TILE * WG_TILE) + (co - ct * WG_TILE) * WG_TILE + (np - nt * WG_TILE);
      d = dw + (long long)co * NP + np;
      for (int s0 = ty; s0 < splits; s0 += 32) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = s0 + 8 * u < splits ? *(const float4*)(src + (size_t)(s0 + 8 * u) * sstride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w;
        }
      }
    }
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && i < total) {
      float4 o = *(const float4*)d;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        o.x += red[r][tx].x; o.y += red[r][tx].y; o.z += red[r][tx].z; o.w += red[r][tx].w;
      }
      *(float4*)d = o;
    }
    __syncthreads();
  }
}

// per-channel sum over pixels of a bf16 NHWC tensor (bias gradient of the head convs)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int ld, int c, long long pixels, float* __restrict__ out) {
  const int ch = blockIdx.x * 64 + (threadIdx.x & 63);
  const int pl = threadIdx.x >> 6;
  __shared__ float red[4][64];
  float s = 0.f;
  if (ch < c)
    for (long long m = blockIdx.y * 4 + pl; m < pixels; m += (long long)gridDim.y * 4) s += s2f(x[m * ld + ch]);
  red[pl][threadIdx.x & 63] = s;
  __syncthreads();
  if (pl == 0 && ch < c) atomicAdd(out + ch, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Two-stage, fixed-order form of the same column sums (bias gradients): up to 512 workgroups, four independent 16-byte loads in flight per
// lane, one partial row per workgroup (plain stores into the weight-gradient workspace, which is dead once the slab reduce has run on the
// same stream), folded by colsum_fold_kernel.  The one-stage form below (<= 128 workgroups ending in same-address atomics) read the 82 MB
// gradient of a RetinaNet tower convolution at 0.3 TB/s: 2.3 ms of a 26 ms step over its 58 launches.
__global__ __launch_bounds__(256) void colsum8p_kernel(const bf16_t* __restrict__ x0, int ld, int c, long long pixels, float* __restrict__ partial0) {
  __shared__ float red[256 * 8];
  // blockIdx.y = chunk of 256 channel groups (2048 channels): the 1204-class head has 1355 groups
  const int groups_all = (c + 7) >> 3, g0 = blockIdx.y * 256;
  const int groups = min(256, groups_all - g0);
  const bf16_t* x = x0 + g0 * 8;
  const int npl = 256 / groups;
  const int gl = threadIdx.x % groups, pl = threadIdx.x / groups;
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = 0.f;
  if (pl < npl) {
    const long long stride = (long long)gridDim.x * npl;
    long long m = (long long)blockIdx.x * npl + pl;
    for (; m + 3 * stride < pixels; m += 4 * stride) {
      uint4 u[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) u[q] = *(const uint4*)(x + (m + q * stride) * ld + gl * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned uu[4] = {u[q].x, u[q].y, u[q].z, u[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[2 * e] += s2f((bf16_t)(uu[e] & 0xFFFF));
          a[2 * e + 1] += s2f((bf16_t)(uu[e] >> 16));
        }
      }
    }
    for (; m < pixels; m += stride) {
      const uint4 u = *(const uint4*)(x + m * ld + gl * 8);
      const unsigned uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[2 * e] += s2f((bf16_t)(uu[e] & 0xFFFF));
        a[2 * e + 1] += s2f((bf16_t)(uu[e] >> 16));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = a[k];
  __syncthreads();
  if (pl == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = 0.f;
      for (int r = 0; r < npl; ++r) t += red[(r * groups + gl) * 8 + k];
      partial0[(size_t)blockIdx.x * groups_all * 8 + (size_t)(g0 + gl) * 8 + k] = t;
    }
  }
}
// 32 channels x 8 row parts per workgroup, fixed-order LDS fold (one thread per channel walking 512 rows alone took up to 220 us)
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ partial, int rows, int c8, int c, float* __restrict__ out) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int ch = blockIdx.x * 32 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (ch < c) {
    int r = part;
    for (; r + 8 < rows; r += 16) {
      s0 += partial[(size_t)r * c8 + ch];
      s1 += partial[(size_t)(r + 8) * c8 + ch];
    }
    if (r < rows) s0 += partial[(size_t)r * c8 + ch];
  }
  red[part][cl] = s0 + s1;
  __syncthreads();
  if (part == 0 && ch < c) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][cl];
    out[ch] += t;
  }
}

// vector form: 16-byte loads (8 channels per lane), rows spread over the lanes that do not fit a channel group
__global__ __launch_bounds__(256) void colsum8_kernel(const bf16_t* __restrict__ x, int ld, int c, long long pixels, float* __restrict__ out) {
  __shared__ float red[256 * 8];
  const int groups = (c + 7) >> 3;
  const int npl = 256 / groups;                        // pixel lanes per workgroup (groups <= 256 checked on the host)
  const int gl = threadIdx.x % groups, pl = threadIdx.x / groups;
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = 0.f;
  if (pl < npl) {
    const long long stride = (long long)gridDim.x * npl;
    long long m = (long long)blockIdx.x * npl + pl;
    for (; m + stride < pixels; m += 2 * stride) {      // two independent 16-byte loads in flight
      const uint4 u = *(const uint4*)(x + m * ld + gl * 8), v = *(const uint4*)(x + (m + stride) * ld + gl * 8);
      const unsigned uu[4] = {u.x, u.y, u.z, u.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[2 * q] += s2f((bf16_t)(uu[q] & 0xFFFF)) + s2f((bf16_t)(vv[q] & 0xFFFF));
        a[2 * q + 1] += s2f((bf16_t)(uu[q] >> 16)) + s2f((bf16_t)(vv[q] >> 16));
      }
    }
    for (; m < pixels; m += stride) {
      const uint4 u = *(const uint4*)(x + m * ld + gl * 8);
      const unsigned uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[2 * q] += s2f((bf16_t)(uu[q] & 0xFFFF));
        a[2 * q + 1] += s2f((bf16_t)(uu[q] >> 16));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[threadIdx.x * 8 + k] = a[k];
  __syncthreads();
  if (pl == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = 0.f;
      for (int r = 0; r < npl; ++r) t += red[(r * groups + gl) * 8 + k];
      if (gl * 8 + k < c) atomicAdd(out + gl * 8 + k, t);
    }
  }
}

// Split count over the pixel axis from a small cost model (us): whole rounds of 512 resident workgroups, ~0.85 us per
// 64-pixel k-step at 2 workgroups/CU, plus the slab round trip (~3 TB/s) and one extra launch when splitting.
int choose_splits(int tiles, int M, double out_bytes) {
  (void)out_bytes;
  // default when the shape was not autotuned: fill (not exceed) one round of 512 resident workgroups
  int sp = max(1, min(1024, 512 / max(1, tiles)));
  sp = max(1, min(sp, M / 2048));
  while (sp > 1) {
    const int chunk = ((M + sp - 1) / sp + WG_BKP - 1) / WG_BKP * WG_BKP;
    if ((M + chunk - 1) / chunk == sp) break;
    --sp;
  }
  return sp;
}

// shape key -> split count found by mi355det_conv_autotune (part of the tune record); + WG_FORM8 = the 256 x 256 phase-staggered kernel
TuneMap& g_wgrad_tuned = tune_table(TUNE_WGRAD);
int g_wgrad_force = 0;
constexpr int WG_FORM8 = 1 << 16;

unsigned long long wgrad_key(const mi355det_conv_shape* s) {
  unsigned long long k = (unsigned long long)(s->n * s->ho * s->wo);
  k = k * 4099 + s->cout;
  k = k * 4099 + s->cin;
  k = k * 17 + s->ksize * 4 + s->stride;
  k = k * 2 + MI355_F16;
  return k;
}

bool split_valid(int M, int sp) {
  const int chunk = ((M + sp - 1) / sp + WG_BKP - 1) / WG_BKP * WG_BKP;
  return (M + chunk - 1) / chunk == sp;
}

bf16_t* g_zero_page_w = nullptr;
int ensure_zero_page_w() {
  if (g_zero_page_w) return 0;
  void* p = nullptr;
  if (hipMalloc(&p, 4096) != hipSuccess || hipMemset(p, 0, 4096) != hipSuccess) return fail(MI355DET_ELAUNCH, "%s: zero page alloc failed", "wgrad");
  g_zero_page_w = (bf16_t*)p;
  return 0;
}

// wgrad8_kernel: pieces of 4 pixels spanning at most two image rows, one wrap per 64-pixel advance, 31-bit byte offsets, at least one whole 256-wide
// tile in both directions (a narrower output would multiply zero fragments: the 128 x 128 kernel is the better tile there; the last co tile
// of a wide output may be partial - the 10 836 channels of the 1204-class cls_logits are 42.3 tiles)
bool wgrad8_applicable(const mi355det_conv_shape* s) {
  const long long NP = (long long)s->ksize * s->ksize * s->cin;
  if (s->wo < 4 || WG_BKP / s->wo + 1 > s->ho || s->cout < 256 || NP < 256 || s->cin % 8 != 0) return false;
  return ((long long)s->n * s->h * s->w + (long long)s->pad * (s->w + 1)) * s->in_ld * 2 < 0x7FFFFFF0ll;
}

}  // namespace

extern "C" {

size_t mi355det_conv_wgrad_workspace(const mi355det_conv_shape* s) {
  if (!s) return 0;
  // room for the largest split count the autotuner may pick: capped at 128 MiB, but never below three splits (the 1204-class cls_logits has
  // 100 MB of dW: its 387 tiles of 256 x 256 are 1.5 rounds of 256 CUs with one pixel range and 3.0 with two)
  const size_t tiles = (size_t)((s->cout + WG_TILE - 1) / WG_TILE) * (size_t)((s->ksize * s->ksize * s->cin + WG_TILE - 1) / WG_TILE);
  const size_t per_split = tiles * WG_TILE * WG_TILE * sizeof(float);
  size_t splits = 1024;
  while (splits > 3 && splits * per_split > ((size_t)128 << 20)) --splits;
  return splits * per_split;
}

// Times the candidate split counts on the caller's buffers (synchronises: plan-build time only, never in the step)
// and remembers the fastest for this shape.  dw receives garbage accumulations: the caller re-zeroes it.
int mi355det_conv_wgrad_autotune(const mi355det_conv_shape* s, const void* x, const void* dy, float* dw, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (!s) return fail(MI355DET_EINVAL, "%s: null shape", "wgrad_autotune");
  if (tune_locked_has(TUNE_WGRAD, wgrad_key(s))) return g_wgrad_tuned[wgrad_key(s)];      // the split came from a tune record: not timed again
  const int M = s->n * s->ho * s->wo;
  const size_t tiles = (size_t)((s->cout + WG_TILE - 1) / WG_TILE) * (size_t)((s->ksize * s->ksize * s->cin + WG_TILE - 1) / WG_TILE);
  const size_t per_split = tiles * WG_TILE * WG_TILE * sizeof(float);
  const int cands[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64, 96, 128, 192, 256, 384, 512, 768, 1024};
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(MI355DET_ELAUNCH, "%s: event create failed", "wgrad_autotune");
  int best = -1;
  float best_ms = 1e30f;
  int tried[32], n_tried = 0;
  float tried_ms[32];
  for (int sp : cands) {
    if (sp > 1 && (sp * per_split > workspace_bytes || M / sp < 512 || (size_t)sp * tiles > 4096)) continue;
    if (!split_valid(M, sp)) continue;
    g_wgrad_force = sp;
    int e = mi355det_conv_wgrad(s, x, dy, dw, nullptr, workspace, workspace_bytes, stream);   // warm-up
    if (e) {
      g_wgrad_force = 0;
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      return e;
    }
    float ms = 1e30f;
    for (int b = 0; b < 2; ++b) {      // the faster of two batches of three (one batch let a cold L2 or a neighbour's burst decide)
      (void)hipEventRecord(e0, S(stream));
      for (int r = 0; r < 3; ++r) (void)mi355det_conv_wgrad(s, x, dy, dw, nullptr, workspace, workspace_bytes, stream);
      (void)hipEventRecord(e1, S(stream));
      (void)hipEventSynchronize(e1);
      float t = 0.f;
      (void)hipEventElapsedTime(&t, e0, e1);
      if (t < ms) ms = t;
    }
    tried[n_tried] = sp;
    tried_ms[n_tried++] = ms;
    if (ms < best_ms) {
      best_ms = ms;
      best = sp;
    }
  }
  // the phase-staggered 256 x 256 form: one or two whole rounds of one-workgroup-per-CU launches
  int best8 = -1;
  float best8_ms = 1e30f;
  if (wgrad8_applicable(s) && !g_wgrad8_off) {
    const int t8 = (int)(((s->cout + 255) / 256) * ((s->ksize * s->ksize * s->cin + 255) / 256));
    int c8[7] = {256 / t8, 512 / t8, 128 / t8, 768 / t8, 1, 2, 3};
    for (int a = 0; a < 7; ++a) {
      const int sp = c8[a];
      bool dup = sp < 1;
      for (int b = 0; b < a; ++b) dup = dup || c8[b] == sp;
      if (dup || (sp > 1 && (sp * per_split > workspace_bytes || M / sp < 512)) || !split_valid(M, sp)) continue;
      g_wgrad_force = sp | WG_FORM8;
      if (int e = mi355det_conv_wgrad(s, x, dy, dw, nullptr, workspace, workspace_bytes, stream)) {
        g_wgrad_force = 0;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return e;
      }
      float ms = 1e30f;
      for (int b = 0; b < 2; ++b) {
        (void)hipEventRecord(e0, S(stream));
        for (int r = 0; r < 3; ++r) (void)mi355det_conv_wgrad(s, x, dy, dw, nullptr, workspace, workspace_bytes, stream);
        (void)hipEventRecord(e1, S(stream));
        (void)hipEventSynchronize(e1);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0, e1);
        if (t < ms) ms = t;
      }
      if (ms < best8_ms) {
        best8_ms = ms;
        best8 = sp;
      }
    }
  }
  // Beside the data-gradient stream fewer, longer workgroups and less slab traffic win over the split count that is fastest alone (the step-level
  // refinement of round 4 halved the split counts of the big layers: profiles/r04_ab_results.md 7): take the SMALLEST split count within 4 % of
  // the fastest one.
  for (int i = 0; i < n_tried; ++i)
    if (tried_ms[i] <= best_ms * 1.04f) {
      best = tried[i];
      break;
    }
  if (best8 > 0 && best8_ms < best_ms * 0.97f) best = best8 | WG_FORM8;
  g_wgrad_force = 0;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (best > 0) {
    g_wgrad_tuned[wgrad_key(s)] = best;
    tune_mark_timed(TUNE_WGRAD, wgrad_key(s));
  }
  return best;
}

int mi355det_conv_wgrad(const mi355det_conv_shape* s, const void* x, const void* dy, float* dw, float* dbias, void* workspace,
                        size_t workspace_bytes, void* stream) {
  if (!s) return fail(MI355DET_EINVAL, "%s: null shape", "conv_wgrad");
  if (int e = ensure_zero_page_w()) return e;
  if (s->cin % 8 != 0) return fail(MI355DET_EINVAL, "%s: Cin must be a multiple of 8 (got %lld)", "conv_wgrad", s->cin);
  if (s->out_ld % 8 != 0 || s->in_ld % 8 != 0) return fail(MI355DET_EINVAL, "%s: pixel pitches must be multiples of 8", "conv_wgrad");
  WgradParams p{};
  p.x = (const bf16_t*)x;
  p.dy = (const bf16_t*)dy;
  p.dw = dw;
  p.zero = g_zero_page_w;
  p.M = s->n * s->ho * s->wo;
  p.Ho = s->ho; p.Wo = s->wo; p.H = s->h; p.W = s->w;
  p.ldx = s->in_ld; p.lddy = s->out_ld;
  p.Cin = s->cin; p.Cout = s->cout; p.stride = s->stride; p.pad = s->pad; p.ks = s->ksize;
  p.T = s->ksize * s->ksize;
  p.NP = p.T * p.Cin;
  p.co_tiles = (p.Cout + WG_TILE - 1) / WG_TILE;
  p.np_tiles = (p.NP + WG_TILE - 1) / WG_TILE;
  p.ablate = g_wgrad_ablate;
  const int tiles = p.co_tiles * p.np_tiles;
  int splits = choose_splits(tiles, p.M, 4.0 * p.Cout * (double)p.NP);
  bool form8 = false;
  {
    auto it = g_wgrad_tuned.find(wgrad_key(s));
    if (it != g_wgrad_tuned.end()) {
      splits = it->second & (WG_FORM8 - 1);
      form8 = (it->second & WG_FORM8) != 0;
    }
    const int force = g_wgrad_force > 0 ? g_wgrad_force : g_wgrad_force_dbg;
    if (force > 0 && split_valid(p.M, force & (WG_FORM8 - 1))) {
      splits = force & (WG_FORM8 - 1);
      form8 = (force & WG_FORM8) != 0;
    }
    if (splits < 1) splits = 1;
    if (g_wgrad_force <= 0 && (g_wgrad_force_dbg & WG_FORM8) && !(form8 && wgrad8_applicable(s)))      // the diagnostic switch must not fall back silently
      return fail(MI355DET_EINVAL, "%s: the phase-staggered kernel was forced (debug key 7) for a shape or split count it does not take", "conv_wgrad");
    form8 = form8 && wgrad8_applicable(s);      // (a record written for another build: fall back to the 128 x 128 kernel, same split count)
    const size_t per_split = (size_t)tiles * WG_TILE * WG_TILE * sizeof(float);
    while (splits > 1 && (splits * per_split > workspace_bytes || !split_valid(p.M, splits))) --splits;
  }
  int chunk = (p.M + splits - 1) / splits;
  chunk = (chunk + WG_BKP - 1) / WG_BKP * WG_BKP;
  splits = (p.M + chunk - 1) / chunk;
  p.splits = splits;
  p.chunk = chunk;
  p.slab = nullptr;
  if (splits > 1) {
    const size_t need = (size_t)splits * tiles * WG_TILE * WG_TILE * sizeof(float);
    if (!workspace || workspace_bytes < need) return fail(MI355DET_EWORKSPACE, "%s: workspace too small (%lld bytes needed)", "conv_wgrad", (long long)need);
    if (p.NP % 4 != 0) return fail(MI355DET_EINVAL, "%s: k*k*Cin must be a multiple of 4", "conv_wgrad");
    p.slab = (float*)workspace;
  }
  p.step_q = WG_BKP / p.Wo;
  p.step_r = WG_BKP % p.Wo;
  p.dWo = make_fastdiv((unsigned)p.Wo);
  p.dHo = make_fastdiv((unsigned)p.Ho);
  p.dCin = make_fastdiv((unsigned)p.Cin);
  // (Measured and reverted, round 4: ONE workgroup per CU - 96 KB of dynamic LDS - for launches with many LONG workgroups.  The 1204-class
  //  cls_logits weight gradient of RetinaNet-LVIS has 1530 tiles x 1250 k-steps = 1.6 ms per workgroup and no split fits the workspace; two of
  //  them per CU leave 32 KB of LDS, so a 64 KB tile of the dependency-chain stream waits for a whole round: three FPN data gradients of
  //  20-50 us each took 1.6 ms (round 3's "FPN dgrad 4.84 ms at 34 TFLOP/s", profiles/r04_retinanet_r101_timeline.md).  One per CU removed
  //  that wait (FPN dgrad 5.15 -> 0.89 ms) but the step went 40.7 -> 42.3 ms: this kernel lost 12-48 %, and the wait moved to the next
  //  small kernels of both streams.  The step is bound by the total work of the cls_logits kernels, not by who waits for whom.)
  const int lds = 2 * 2 * WG_BKP * WG_ROWB;
  // per-wave fragment counts: the four waves (2 x 2) split min(Cout,128) x min(NP,128) real channels
  const int ci = p.Cout <= 32 ? 1 : (p.Cout <= 64 ? 2 : 4), cj = p.NP <= 32 ? 1 : (p.NP <= 64 ? 2 : 4);
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(256), lds, S(stream), p);
  };
  // scalar pixel bookkeeping: a 4-pixel piece may span at most two image rows, byte offsets into x must fit 31 bits
  const int grp4 = g_wgrad_general != 0 || p.Wo < 4 || WG_BKP / p.Wo + 1 > p.Ho ? 0 : (p.Wo % 16 == 0 ? 3 : (p.Wo % 4 == 0 ? 1 : 2));
  const bool fits =
                    ((long long)s->n * p.H * p.W + (long long)p.pad * (p.W + 1)) * p.ldx * 2 < 0x7FFFFFF0ll &&
                    (long long)chunk * p.lddy * 2 < 0x7FFFFFF0ll;
  if (form8 && fits) {
    const int tiles8 = ((p.Cout + W8_TILE - 1) / W8_TILE) * ((p.NP + W8_TILE - 1) / W8_TILE);
    static DeviceOnce attr8;
    attr8.once([&] {
      (void)hipFuncSetAttribute((const void*)wgrad8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
      (void)hipFuncSetAttribute((const void*)wgrad8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS);
    });
    if (p.Wo % 4 == 0 && p.M % 4 == 0) hipLaunchKernelGGL(wgrad8_kernel<false>, dim3(tiles8 * splits), dim3(512), W8_LDS, S(stream), p);
    else hipLaunchKernelGGL(wgrad8_kernel<true>, dim3(tiles8 * splits), dim3(512), W8_LDS, S(stream), p);
  } else
#define WG_GO(a, b) (!fits || grp4 == 0 ? go(wgrad_kernel<a, b, 0>) : grp4 == 1 ? go(wgrad_kernel<a, b, 1>) : grp4 == 2 ? go(wgrad_kernel<a, b, 2>) : go(wgrad_kernel<a, b, 3>))
  switch (ci * 8 + cj) {
    case 1 * 8 + 1: WG_GO(1, 1); break;
    case 1 * 8 + 2: WG_GO(1, 2); break;
    case 1 * 8 + 4: WG_GO(1, 4); break;
    case 2 * 8 + 1: WG_GO(2, 1); break;
    case 2 * 8 + 2: WG_GO(2, 2); break;
    case 2 * 8 + 4: WG_GO(2, 4); break;
    case 4 * 8 + 1: WG_GO(4, 1); break;
    case 4 * 8 + 2: WG_GO(4, 2); break;
    default: WG_GO(4, 4); break;
  }
#undef WG_GO
  if (p.slab) {
    const long long total = (long long)p.Cout * (p.NP / 4);
    if (splits >= 8) {
      const int gx = (int)min((long long)4096, (total + 31) / 32);
      hipLaunchKernelGGL(wgrad_reduce8_kernel, dim3(gx), dim3(256), 0, S(stream), p.slab, dw, p.Cout, p.NP, p.co_tiles, p.np_tiles, splits);
    } else {
      const int gx = (int)min((long long)2048, (total + 255) / 256);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, 1), dim3(256), 0, S(stream), p.slab, dw, p.Cout, p.NP, p.co_tiles, p.np_tiles, splits);
    }
  }
  if (dbias) {
    const int groups = (s->cout + 7) / 8;
    const bool vec = s->out_ld % 8 == 0 && groups * 8 <= s->out_ld && (((uintptr_t)dy) & 15) == 0;
    const int npl = 256 / (groups < 256 ? groups : 256);
    const int gp = (int)max(1ll, min((long long)512, ((long long)p.M + (long long)npl * 8 - 1) / ((long long)npl * 8)));
    if (vec && gp > 1 && workspace && workspace_bytes >= (size_t)gp * groups * 8 * sizeof(float)) {
      // two stages, fixed order (the slabs at the head of the workspace are dead: their reduce ran on this stream above)
      hipLaunchKernelGGL(colsum8p_kernel, dim3(gp, (groups + 255) / 256), dim3(256), 0, S(stream), (const bf16_t*)dy, s->out_ld, s->cout, (long long)p.M, (float*)workspace);
      hipLaunchKernelGGL(colsum_fold_kernel, dim3((s->cout + 31) / 32), dim3(256), 0, S(stream), (const float*)workspace, gp, groups * 8, s->cout, dbias);
    } else if (vec && groups <= 256) {
      // few workgroups: every one ends with c same-address atomics (1024 of them cost more than the reads)
      const int gx = (int)max(1ll, min((long long)128, ((long long)p.M + (long long)npl * 16 - 1) / ((long long)npl * 16)));
      hipLaunchKernelGGL(colsum8_kernel, dim3(gx), dim3(256), 0, S(stream), (const bf16_t*)dy, s->out_ld, s->cout, (long long)p.M, dbias);
    } else {
      const int gy = (int)min((long long)256, ((long long)p.M + 255) / 256);
      hipLaunchKernelGGL(colsum_kernel, dim3((p.Cout + 63) / 64, gy), dim3(256), 0, S(stream), (const bf16_t*)dy, s->out_ld, s->cout, (long long)p.M, dbias);
    }
  }
  return check_launch("conv_wgrad");
}

}  // extern "C"
