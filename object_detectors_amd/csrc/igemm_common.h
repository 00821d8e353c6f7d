// Shared pieces of the implicit-GEMM convolution kernels (gfx950): problem description, LDS-DMA helpers, LDS swizzles and the
// common epilogue (BN statistics / fp32 head / residual / affine / fused BN-backward sums).  Included by conv_kernels.hip and
// igemm8_kernels.hip; everything is internal linkage.
#pragma once
#include "common.h"

using namespace mi355;

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;

#define MAX_TAPS 9

// division by a launch-invariant divisor without the ~40-instruction integer divide (n < 2^31)
struct FastDiv {
  unsigned mul, shift;
};
static FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.shift = l;
  f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv f) { return (__umulhi(f.mul, n) + n) >> f.shift; }

struct IgemmParams {
  const bf16_t* x;       // input activations (fwd: x, dgrad: dy)
  const bf16_t* w;       // packed weights [CoutPad][T*Cin]
  void* y;               // output (bf16 or fp32)
  const float* bias;     // EPI_F32 / EPI_AFF: per-channel shift
  const float* scale;    // EPI_F32 / EPI_AFF: per-channel multiplier on the accumulator or null (FrozenBatchNorm2d folded into the conv)
  int relu;              // EPI_F32 / EPI_AFF: 1 = ReLU after (scale, shift, residual); 2 = LeakyReLU(slope) BEFORE the residual (Darknet)
                         // EPI_RES: 3 / 4 = MASK mode: `res` is not added, it is the ACTIVATION a of the layer whose gradient is being written and
                         //          out = bf16(acc) * scale[c] * (a > 0) (3) or bf16(acc) * scale[c] (4): the FrozenBN + ReLU backward of that layer
                         //          (mi355det_relu_affine_bwd) folded into the data gradient that produces its input
  long long ynstride;    // EPI_F32: elements between images of y (heads write straight into the level-concatenated tensor)
  const bf16_t* z;       // EPI_BNRED: pre-BN output of the layer whose activation gradient this dgrad writes
  const float* ss;       // EPI_BNRED: that layer's [4*Cout] scale, shift, mean, invstd
  int ldz;
  float slope;
  float* stats;          // EPI_STATS: [gridM][2][CoutPad] partial sum / sumsq
  const bf16_t* res;     // EPI_RES: residual to add
  const bf16_t* zero;    // >= 256 B of zeros
  int M, MH, MW;         // lattice: M = N*MH*MW
  int Hin, Win, ldin, Cin, sin;
  int Hout, Wout, ldout, so, oy0, ox0;
  int sox;               // output stride along x (0 = the same as `so`): the class-concatenated stride-2 data gradient steps 2 in y and 1 in x
  int Cout, CoutPad, ldres;
  int T;
  int dy[MAX_TAPS], dx[MAX_TAPS];
  FastDiv dMW, dMH;
  int tap_pad;           // elements: -min over taps of (dy*Win+dx)*ldin, >= 0 (keeps scalar tap offsets non-negative)
  unsigned long long dy_pack, dx_pack;   // 4-bit fields (value+2) per tap: the tap table in two scalar registers
  int ksplit;            // > 1: split-K over the channel axis (igemm_kernel, EPI_F32 only): grid = tiles * ksplit, split sp reduces channels
                         //      [sp*Cin/ksplit, (sp+1)*Cin/ksplit) of every tap and writes its fp32 partial tile at y + sp * ysplit
  long long ysplit;      // elements between the partial outputs of two splits
  int lin_in, lin_out;   // 1: lattice pixel m IS the input pixel (single centred tap, unit stride) / the output pixel: no divisions per row
  unsigned long long* dbg;   // diagnostic build only (PROF): per-wave phase cycle sums
};

enum { EPI_STATS = 0, EPI_F32 = 1, EPI_RES = 2, EPI_PLAIN = 3, EPI_AFF = 4, EPI_BNRED = 5 };
#define EPI_LDS_OFF 4096                               // epilogue staging starts behind the BN-statistics scratch
#define EPI_LDS_BYTES(nwaves) (EPI_LDS_OFF + (nwaves) * (64 * (8 * 16 * 2 + 16) + 256))   // upper bound (TN <= 8)

// phase-staggered 256x256x64 kernel (igemm8_kernels.hip)
bool igemm8_applicable(const IgemmParams& p);
int igemm8_launch(int epi, const IgemmParams& p, hipStream_t st, int rows = 256);      // rows: pixel-tile height 256 (default), 224 or 208
void igemm8_set_dbg(unsigned long long* ptr);
void igemm8_set_dbg_mode(int mode);

namespace {

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
}

// LDS-DMA through a buffer descriptor: per-lane 32-bit byte offset + SCALAR offset (the tap / k-step part of the
// address costs no vector instruction); a lane whose voffset is out of range gets zeros written to LDS, which is
// exactly the convolution's zero padding.
// Issued as inline assembly ON PURPOSE: for the builtin the compiler's wait-count pass assumes every later ds_read may alias
// the DMA's LDS destination and puts s_waitcnt vmcnt(0) in front of the next fragment read, which turns every counted wait of
// the ring into a full drain (no load/compute overlap inside a workgroup, whatever the ring depth).  The kernels order DMA and
// reads themselves (counted vmcnt + barrier), so the compiler must not know about the LDS side of these loads.
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

// sum over the 16 lanes of a DPP row (lanes sharing lane>>4), result in every lane: 4 VALU+DPP ops, no LDS traffic
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

template <int BK>
__device__ __forceinline__ int swz(int row) {
  return BK == 64 ? ((row >> 1) & 7) : ((-(row >> 2)) & 3);
}

// Swizzle of the SHARED pixel tile: its fragments are read at row offsets -1 / 0 / +1, and the pair-wise pattern above
// ((row >> 1) & 7) is conflict-free only for even starts: on odd starts two of the sixteen 16-byte slots of a ds_read_b128 lane
// group collide (SQ_LDS_BANK_CONFLICT measured 7 % of the kernel time against 1.4 % for the plain tile).  row & 7 is
// conflict-free for every start (exhaustive check over all 16 alignments and both k halves).
template <int BK>
__device__ __forceinline__ int swz_shift(int row) {
  return BK == 64 ? (row & 7) : swz<BK>(row);
}

// WM x WN waves; each wave computes 64 pixels x (TN*16) channels.  NST-deep LDS ring: the LDS-DMA of
// k-steps s+1 .. s+NST-1 stays in flight (counted vmcnt, raw s_barrier) while k-step s runs on the MFMAs —
// with ~1.5-2 us of loaded-memory latency the bytes in flight per CU, not the MFMA rate, set the speed.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Per-channel partial sums of one workgroup tile -> rows of the partial buffer [row][2][CoutPad].  A row always stands for
// 128 consecutive lattice pixels (row = m / 128) whatever the tile height, so every tile configuration fills the same
// rows (a 256-pixel tile writes two) and no row is left stale when the autotuner switches configurations.
template <int WM, int TM, int BN>
__device__ __forceinline__ void write_partial_rows(const IgemmParams& p, const float* sred, int tid, int nthreads, int mt, int n0) {
  constexpr int BM = WM * TM * 16;
  static_assert(BM % 128 == 0, "tiles are multiples of 128 pixels");
  constexpr int G = BM / 128;            // rows per tile
  constexpr int WPG = WM / G;            // waves (in M) per row
  static_assert(WM % G == 0, "a 128-pixel row must be covered by whole waves");
  const int nrows = (p.M + 127) / 128;
  for (int i = tid; i < BN * G; i += nthreads) {
    const int c = i % BN, g = i / BN;
    const int row = mt * G + g;
    if (row >= nrows) continue;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < WPG; ++w) {
      s1 += sred[((g * WPG + w) * BN + c) * 2 + 0];
      s2 += sred[((g * WPG + w) * BN + c) * 2 + 1];
    }
    float* dst = p.stats + (long long)row * 2 * p.CoutPad;
    dst[n0 + c] = s1;
    dst[p.CoutPad + n0 + c] = s2;
  }
}

// The same rows for a tile height that is NOT a multiple of 128 pixels (the 224 / 208 / 192-pixel tiles of igemm8_kernel): tile t covers
// lattice pixels [t * BMV, (t + 1) * BMV) and OWNS the rows [floor(t * BMV / 128), floor((t + 1) * BMV / 128)) - the row its first pixel lies
// in (no other tile starts there, since BMV >= 128) and the rows up to the one the next tile starts in; the last tile owns the rest.  The
// tile's sums (both wave halves) go to its first row, zeros to the others (at most three rows in all): the finalisation kernels add all rows,
// so the partial buffer keeps its configuration-independent shape and no row is ever stale.
template <int BMV, int BN>
__device__ __forceinline__ void write_partial_rows_var(const IgemmParams& p, const float* sred, int tid, int nthreads, int mt, int n0) {
  static_assert(BMV >= 128 && BMV <= 256, "every tile starts in a row of its own");
  const int nrows = (p.M + 127) / 128;
  const int last = (p.M + BMV - 1) / BMV - 1;
  const int r0 = (mt * BMV) >> 7, r1 = mt == last ? nrows : ((mt + 1) * BMV) >> 7;      // owned rows [r0, r1)
  for (int i = tid; i < BN * 3; i += nthreads) {
    const int c = i % BN, g = i / BN;
    const int row = r0 + g;
    if (row >= r1 || row >= nrows) continue;
    float s1 = 0.f, s2 = 0.f;
    if (g == 0) {
      s1 = sred[(0 * BN + c) * 2 + 0] + sred[(1 * BN + c) * 2 + 0];
      s2 = sred[(0 * BN + c) * 2 + 1] + sred[(1 * BN + c) * 2 + 1];
    }
    float* dst = p.stats + (long long)row * 2 * p.CoutPad;
    dst[n0 + c] = s1;
    dst[p.CoutPad + n0 + c] = s2;
  }
}

// ---- shared epilogue.  `consumer` = this wave holds accumulators (false for a dedicated loader wave, which only
//      takes part in the barrier and the final statistics write)
//      BMV > 0 (igemm8_kernel's 224 / 208 / 192-pixel tiles, WM == 2): the wave's rows start at tile row `wrow` instead of wm * TM * 16 and only
//      its first `jlim` 16-pixel tiles exist (the accumulators of the others are zero and are neither staged nor stored)
template <int WM, int WN, int TM, int TN, int EPI, int BMV = 0>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x4_t (&acc)[TN][TM], char* smem, int tid, int nthreads, bool consumer,
                                               int wm, int wn, int lane, int mt, int n0, int m0, long long yoff = 0, int wrow = 0, int jlim = TM) {
  constexpr int BN = WN * TN * 16;
  const int fr = lane & 15, fq = lane >> 4;
  const int wbase = BMV > 0 ? wrow : wm * (TM * 16);          // first tile row of this wave
  const int rlim = BMV > 0 ? jlim * 16 : TM * 16;             // rows of this wave that exist
  // ---- epilogue: lane holds channels co = n0 + wn*TN*16 + i*16 + fq*4 + r (r=0..3) of pixel
  //      m = m0 + wm*TM*16 + j*16 + fr
  if (EPI == EPI_STATS) {
    float* sred = (float*)smem;   // [WM][BN][2]
#pragma unroll
    for (int i = 0; consumer && i < TN; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          // statistics of the STORED (bf16-rounded) tensor: the BN backward formulas then hold exactly
          const float v = s2f(f2s(acc[i][j][r]));
          s1 += v;
          s2 += v * v;
        }
        s1 = row16_sum(s1);
        s2 = row16_sum(s2);
        if (fr == 0) {
          const int c = wn * (TN * 16) + i * 16 + fq * 4 + r;
          sred[(wm * BN + c) * 2 + 0] = s1;
          sred[(wm * BN + c) * 2 + 1] = s2;
        }
      }
    }
    __syncthreads();
    if constexpr (BMV > 0) write_partial_rows_var<BMV, BN>(p, (const float*)smem, tid, nthreads, mt, n0);
    else write_partial_rows<WM, TM, BN>(p, (const float*)smem, tid, nthreads, mt, n0);
  }
  if (!consumer) return;
  if (EPI == EPI_F32) {
    // head conv_out: fp32 + bias straight from the accumulator layout: a lane holds 4 consecutive channels of one pixel = one 16-byte store
    // where the group lies inside the real channels and the address is aligned (every group but the last of a 255-channel YOLO head, whose
    // rows have a 256-float pitch); the scalar form - 4-byte pieces scattered over 16 rows per instruction - wrote the 209 MB of the 80 x 80
    // head at 1.9 TB/s including the reads.  Per-channel scale / shift are fetched once per channel group, not per pixel.
    float e_sc[TN][4], e_sh[TN][4];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = n0 + wn * (TN * 16) + i * 16 + fq * 4 + r;
        e_sc[i][r] = (p.scale && c < p.Cout) ? p.scale[c] : 1.f;
        e_sh[i][r] = (p.bias && c < p.Cout) ? p.bias[c] : 0.f;
      }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wbase + j * 16 + fr;
      if (m >= p.M || (BMV > 0 && j >= jlim)) continue;
      const int t1 = (int)fdiv((unsigned)m, p.dMW), xx = m - t1 * p.MW;
      const int n = (int)fdiv((unsigned)t1, p.dMH), yy = t1 - n * p.MH;
      const int oy = yy * p.so + p.oy0, ox = xx * (p.sox ? p.sox : p.so) + p.ox0;
      if (oy >= p.Hout || ox >= p.Wout) continue;
      float* orow = (float*)p.y + yoff + (long long)n * p.ynstride + (long long)(oy * p.Wout + ox) * p.ldout;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int co = n0 + wn * (TN * 16) + i * 16 + fq * 4;
        if (co >= p.Cout) continue;
        float* o = orow + co;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r] * e_sc[i][r] + e_sh[i][r];
          if (p.relu == 1) v[r] = fmaxf(v[r], 0.f);
          else if (p.relu == 2) v[r] = v[r] > 0.f ? v[r] : v[r] * p.slope;
        }
        if (co + 3 < p.Cout && (((unsigned long long)o) & 15ull) == 0) {
          *(f32x4_t*)o = f32x4_t{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (co + r < p.Cout) o[r] = v[r];
        }
      }
    }
    return;
  }
  // bf16 outputs: the accumulator layout gives each lane 4 channels of one pixel (8-byte pieces scattered over
  // 16 rows per instruction = partial-line writes).  Stage the wave's tile through a wave-private LDS region and
  // write whole 128-byte channel runs with 16-byte stores (and read the residual the same way).
  constexpr int CW = TN * 16;                 // channels per wave
  constexpr int PITCH = CW * 2 + 16;          // bytes per staged pixel row (+16: spreads the 8-byte writes over banks)
  constexpr int CH16 = CW / 8;                // 16-byte chunks per row
  constexpr int RPP = 64 / CH16;              // rows per pass of the wave
  const int wid = wm + wn * WM;
  char* reg = smem + EPI_LDS_OFF + wid * (64 * PITCH + 256);
  int* rowpix = (int*)(reg + 64 * PITCH);     // pixel index (or -1) of the 64 staged rows
  // EPI_BNRED: BatchNorm-backward partial sums of the gradient tile being written (this lane's 8 channels)
  float bn_sc[EPI == EPI_BNRED ? 8 : 1], bn_sh[EPI == EPI_BNRED ? 8 : 1], bn_mu[EPI == EPI_BNRED ? 8 : 1], bn_is[EPI == EPI_BNRED ? 8 : 1];
  float bn_a1[EPI == EPI_BNRED ? 8 : 1], bn_a2[EPI == EPI_BNRED ? 8 : 1];
  if (EPI == EPI_BNRED) {
    const int cb = n0 + wn * CW + (lane % CH16) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool in = cb + k < p.Cout;
      bn_sc[k] = in ? p.ss[cb + k] : 0.f;
      bn_sh[k] = in ? p.ss[p.Cout + cb + k] : 0.f;
      bn_mu[k] = in ? p.ss[2 * p.Cout + cb + k] : 0.f;
      bn_is[k] = in ? p.ss[3 * p.Cout + cb + k] : 0.f;
      bn_a1[k] = 0.f;
      bn_a2[k] = 0.f;
    }
  }
  f32x4_t aff_sc[EPI == EPI_AFF ? TN : 1], aff_sh[EPI == EPI_AFF ? TN : 1];
  if (EPI == EPI_AFF) {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int cb = n0 + wn * CW + i * 16 + fq * 4;      // Cout % 4 == 0 (checked by the entry point)
      const bool in = cb < p.Cout;
      aff_sc[i] = (in && p.scale) ? *(const f32x4_t*)(p.scale + cb) : f32x4_t{1.f, 1.f, 1.f, 1.f};
      aff_sh[i] = (in && p.bias) ? *(const f32x4_t*)(p.bias + cb) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  }
  const int ch = lane % CH16, rsub = lane / CH16;
  const int co = n0 + wn * CW + ch * 8;
  // EPI_RES mask mode: per-channel multipliers of this lane's 8 channels
  float msc[EPI == EPI_RES ? 8 : 1];
  const bool mask_mode = EPI == EPI_RES && p.relu >= 3;
  if (EPI == EPI_RES) {
#pragma unroll
    for (int k = 0; k < 8; ++k) msc[k] = (mask_mode && p.scale && co + k < p.Cout) ? p.scale[co + k] : 1.f;
  }
  constexpr int NPASS = 64 / RPP;
  constexpr bool kReadsSide = EPI == EPI_RES || EPI == EPI_AFF || EPI == EPI_BNRED;    // residual and / or z tiles are read back
#pragma unroll
  for (int jh = 0; jh < TM / 4; ++jh) {       // 64 pixels at a time
    // Side tiles (skip-connection residual, pre-BN z) come from HBM: issue every load of this 64-pixel block FIRST, so
    // their latency runs under the LDS staging below instead of once per pass.
    int pixp[kReadsSide ? NPASS : 1];
    uint4 rpre[kReadsSide ? NPASS : 1], zpre[EPI == EPI_BNRED ? NPASS : 1];
    if (kReadsSide) {
      const bool want_res = EPI == EPI_RES || p.res != nullptr;
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        const int lr = jh * 64 + pass * RPP + rsub;
        const int m = (BMV > 0 && lr >= rlim) ? 0x7FFFFFFF : m0 + wbase + lr;
        int pixi = -1;
        if (p.lin_out) {
          if (m < p.M && co < p.Cout) pixi = m;
        } else if (m < p.M && co < p.Cout) {
          const int t1 = (int)fdiv((unsigned)m, p.dMW), xx = m - t1 * p.MW;
          const int n = (int)fdiv((unsigned)t1, p.dMH), yy = t1 - n * p.MH;
          const int oy = yy * p.so + p.oy0, ox = xx * (p.sox ? p.sox : p.so) + p.ox0;
          if (oy < p.Hout && ox < p.Wout) pixi = (n * p.Hout + oy) * p.Wout + ox;
        }
        pixp[pass] = pixi;
        rpre[pass] = (want_res && pixi >= 0) ? *(const uint4*)(p.res + (long long)pixi * p.ldres + co) : make_uint4(0, 0, 0, 0);
        if (EPI == EPI_BNRED) zpre[pass] = pixi >= 0 ? *(const uint4*)(p.z + (long long)pixi * p.ldz + co) : make_uint4(0, 0, 0, 0);
      }
    } else {
      const int m = (BMV > 0 && jh * 64 + lane >= rlim) ? 0x7FFFFFFF : m0 + wbase + jh * 64 + lane;
      int pixi = -1;
      if (p.lin_out) {
        if (m < p.M) pixi = m;
      } else if (m < p.M) {
        const int t1 = (int)fdiv((unsigned)m, p.dMW), xx = m - t1 * p.MW;
        const int n = (int)fdiv((unsigned)t1, p.dMH), yy = t1 - n * p.MH;
        const int oy = yy * p.so + p.oy0, ox = xx * (p.sox ? p.sox : p.so) + p.ox0;
        if (oy < p.Hout && ox < p.Wout) pixi = (n * p.Hout + oy) * p.Wout + ox;
      }
      rowpix[lane] = pixi;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = jh * 4 + jj;
      if (BMV > 0 && j >= jlim) continue;           // (wave-uniform) tiles this wave does not have
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        f32x4_t v = acc[i][j];
        if (EPI == EPI_AFF) {
          v = v * aff_sc[i] + aff_sh[i];
          if (p.relu == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : v[r] * p.slope;
          }
        }
        uint2 o;
        o.x = (unsigned)f2s(v[0]) | ((unsigned)f2s(v[1]) << 16);
        o.y = (unsigned)f2s(v[2]) | ((unsigned)f2s(v[3]) << 16);
        *(uint2*)(reg + (jj * 16 + fr) * PITCH + (i * 16 + fq * 4) * 2) = o;
      }
    }
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + rsub;
      const int pixi = kReadsSide ? pixp[pass] : rowpix[row];
      uint4 v = *(const uint4*)(reg + row * PITCH + ch * 16);
      if (pixi >= 0 && co < p.Cout) {
        if (EPI == EPI_RES || ((EPI == EPI_AFF || EPI == EPI_BNRED) && (p.res != nullptr || p.relu == 1))) {
          const uint4 rr = rpre[kReadsSide ? pass : 0];
          const bool relu = EPI == EPI_AFF && p.relu == 1;
          const unsigned vi[4] = {v.x, v.y, v.z, v.w}, ri[4] = {rr.x, rr.y, rr.z, rr.w};
          unsigned oo[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float lo = s2f((bf16_t)(vi[q] & 0xFFFF)) + s2f((bf16_t)(ri[q] & 0xFFFF));
            float hi = s2f((bf16_t)(vi[q] >> 16)) + s2f((bf16_t)(ri[q] >> 16));
            if (EPI == EPI_RES && mask_mode) {
              const float alo = s2f((bf16_t)(ri[q] & 0xFFFF)), ahi = s2f((bf16_t)(ri[q] >> 16));
              lo = (p.relu == 4 || alo > 0.f) ? s2f((bf16_t)(vi[q] & 0xFFFF)) * msc[EPI == EPI_RES ? 2 * q : 0] : 0.f;
              hi = (p.relu == 4 || ahi > 0.f) ? s2f((bf16_t)(vi[q] >> 16)) * msc[EPI == EPI_RES ? 2 * q + 1 : 0] : 0.f;
            }
            if (relu) {
              lo = fmaxf(lo, 0.f);
              hi = fmaxf(hi, 0.f);
            }
            oo[q] = (unsigned)f2s(lo) | ((unsigned)f2s(hi) << 16);
          }
          v = make_uint4(oo[0], oo[1], oo[2], oo[3]);
        }
        *(uint4*)((bf16_t*)p.y + (long long)pixi * p.ldout + co) = v;
        if (EPI == EPI_BNRED) {
          // sums over the STORED (bf16) gradient, as the separate reduce kernel would read it back
          const uint4 zz = zpre[EPI == EPI_BNRED ? pass : 0];
          const unsigned gi[4] = {v.x, v.y, v.z, v.w}, zi[4] = {zz.x, zz.y, zz.z, zz.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int hsel = 0; hsel < 2; ++hsel) {
              const int k = q * 2 + hsel;
              const float gv = s2f((bf16_t)(hsel ? gi[q] >> 16 : gi[q] & 0xFFFF));
              const float zv = s2f((bf16_t)(hsel ? zi[q] >> 16 : zi[q] & 0xFFFF));
              const float yv = zv * bn_sc[k] + bn_sh[k];
              const float dy = yv > 0.f ? gv : gv * p.slope;
              bn_a1[k] += dy;
              bn_a2[k] += dy * ((zv - bn_mu[k]) * bn_is[k]);
            }
          }
        }
      }
    }
  }
  if (EPI == EPI_BNRED) {
    // lanes that share a channel chunk (same lane % CH16) hold partial sums of different rows: fold them, then one lane per
    // chunk publishes the wave's sums; waves stacked in M are combined per 128-pixel row by write_partial_rows
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int o = CH16; o < 64; o <<= 1) {
        bn_a1[k] += __shfl_xor(bn_a1[k], o, 64);
        bn_a2[k] += __shfl_xor(bn_a2[k], o, 64);
      }
    }
    float* sred = (float*)smem;   // [WM][BN][2]
    if (lane < CH16) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = wn * CW + lane * 8 + k;
        sred[(wm * BN + c) * 2 + 0] = bn_a1[k];
        sred[(wm * BN + c) * 2 + 1] = bn_a2[k];
      }
    }
    __syncthreads();
    if constexpr (BMV > 0) write_partial_rows_var<BMV, BN>(p, sred, tid, nthreads, mt, n0);
    else write_partial_rows<WM, TM, BN>(p, sred, tid, nthreads, mt, n0);
  }
}

}  // namespace
