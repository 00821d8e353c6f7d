// Implicit-GEMM convolution on MFMA (gfx950): forward and data-gradient.
//
//   D[co][pix] = sum_k W[co][k] * A[pix][k],   k = tap*Cin + c   (NHWC bf16, fp32 accumulate)
//
// * activations stay NHWC bf16 in HBM; the im2col tile A[pix][k-chunk] is gathered straight into
//   LDS with global_load_lds (16 B per lane, per-lane source address = padding/stride handled by
//   pointing out-of-image rows at a zero page), weights W[co][k] are K-contiguous.
// * LDS rows are XOR-swizzled on the SOURCE side (LDS-DMA writes lane-linear) so the ds_read_b128
//   fragment reads are bank-conflict free for 128-B (BK=64) and 64-B (BK=32) rows.
// * v_mfma_f32_16x16x32_bf16 with weights as the MFMA A operand (rows = output channels): each lane
//   ends up with 4 consecutive channels of one pixel -> 8-byte NHWC stores, and per-channel BatchNorm
//   statistics reduce over 16 lanes.
// * one generic "lattice + tap table" addressing covers 3x3/1x1, stride 1/2 forward and the
//   transposed (dgrad) problems incl. the four parity classes of a stride-2 dgrad.
// * 2-stage LDS pipeline: LDS-DMA of k-step t+1 is in flight while k-step t runs on the MFMAs;
//   2 workgroups per CU overlap each other's waits.
#include "common.h"

#include <cstdlib>

#include <unordered_map>

#include "tune_record.h"

using namespace mi355;

#include "igemm_common.h"

extern int g_wgrad_general;
extern int g_wgrad_force_dbg;
extern int g_wgrad8_off;
extern int g_wgrad_ablate;   // wgrad_kernels.hip
extern int g_dgrad_s2_off;     // dgrad_s2_kernels.hip
int mi355det_internal_dgrad_s2(const mi355det_conv_shape* s, const void* dy, const void* wt, void* dx, const void* residual, int32_t residual_ld,
                               void* stream);

namespace {

#define STAMP(v) do { if (PROF) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } } while (0)

// ABL (ablation, diagnostic builds only): 1 = no LDS-DMA loads, 2 = no fragment reads, 3 = no MFMAs
template <int WM, int WN, int TM, int TN, int BK, int NST, int EPI, bool PROF = false, bool ILV = false, int OCC = 0, int ABL = 0>
__global__ __launch_bounds__(WM* WN * 64, (OCC > 0 ? OCC : (WM * WN >= 8 ? 2 : 1))) void igemm_kernel(const IgemmParams p) {
  unsigned long long t_a = 0, t_b = 0, t_c = 0, t_d = 0, t_e = 0, c_wait = 0, c_bar = 0, c_issue = 0, c_comp = 0, t_begin = 0, t_loop = 0;
  STAMP(t_begin);
  constexpr int NT = WM * WN * 64;
  constexpr int NW = WM * WN;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int ROWB = BK * 2;               // bytes per LDS row
  constexpr int R = 1024 / ROWB;             // rows per LDS-DMA wave instruction
  constexpr int CPR = BK / 8;                // 16-B chunks per row
  constexpr int A_INSTR = BM / R, B_INSTR = BN / R;
  constexpr int A_PER = (A_INSTR + NW - 1) / NW, B_PER = (B_INSTR + NW - 1) / NW;
  constexpr int PER = A_PER + B_PER;         // LDS-DMA instructions per wave per stage (same for every wave)
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int D = NST - 1;                 // prefetch distance
  static_assert(A_INSTR % NW == 0, "A tile must split evenly over the waves");
  // ALL LDS lives in this one array (a second __shared__ object makes hipcc drain vmcnt before ds_reads)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_toff = (int*)(smem + NST * STAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid % WM, wn = wid / WM;

  // XCD-aware block remap (bijective form): blocks that share an A tile (same m-tile, all n-tiles) and
  // neighbouring m-tiles run on one XCD so the re-reads hit that XCD's L2.
  const int ntn = p.CoutPad / BN;
  int nblk = gridDim.x;
  int bid = blockIdx.x;
  int sp = 0;                                  // split-K: the splits of one tile are neighbouring blocks
  if (p.ksplit > 1) {
    const int q = __builtin_amdgcn_readfirstlane(bid / p.ksplit);      // integer division runs on the vector ALU: back to scalar registers
    sp = bid - q * p.ksplit;
    bid = q;
    nblk = __builtin_amdgcn_readfirstlane(nblk / p.ksplit);
  }
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int mt = bid / ntn, nt = bid - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;

  // scalar (per-tap) byte offsets, made non-negative by moving tap_pad into the descriptor base
  if (tid == 0) {
#pragma unroll
    for (int t = 0; t < MAX_TAPS; ++t)
      if (t < p.T) s_toff[t] = ((p.dy[t] * p.Win + p.dx[t]) * p.ldin + p.tap_pad) * 2;
  }

  unsigned long long t_p1 = 0, t_p2 = 0, t_p3 = 0, t_p4 = 0;
  STAMP(t_p1);
  // ---- buffer descriptors (wave-uniform): activations relative to the first image this tile touches
  const int n_first = (int)fdiv(fdiv((unsigned)m0, p.dMW), p.dMH);
  const int Ktot = p.T * p.Cin;
  const srd_t rsrc_x = make_srd(p.x + (long long)n_first * p.Hin * p.Win * p.ldin - p.tap_pad, 0x7FFFFFF0u);
  const srd_t rsrc_w = make_srd(p.w + (long long)n0 * Ktot, 0x7FFFFFF0u);

  // ---- per-lane byte offsets of the rows this lane stages
  const int lrow = lane / CPR, cpos = lane % CPR;
  int a_voff[A_PER];
  unsigned a_valid[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int instr = wid * A_PER + i;
    const int row = instr * R + lrow;
    const int m = m0 + row;
    unsigned vm = 0;
    int voff = OOB_VOFF;
    if (p.lin_in) {   // 1x1, unit stride: the row's pixel index is the address, its only tap is always inside (uniform branch)
      if (m < p.M) {
        voff = ((m - n_first * p.Hin * p.Win) * p.ldin + (cpos ^ swz<BK>(row)) * 8) * 2;
        vm = 1u;
      }
    } else if (m < p.M) {
      const int t1 = (int)fdiv((unsigned)m, p.dMW), xx = m - t1 * p.MW;
      const int n = (int)fdiv((unsigned)t1, p.dMH), yy = t1 - n * p.MH;
      const int iy0 = yy * p.sin, ix0 = xx * p.sin;
      voff = ((((n - n_first) * p.Hin + iy0) * p.Win + ix0) * p.ldin + (cpos ^ swz<BK>(row)) * 8) * 2;
#pragma unroll
      for (int t = 0; t < MAX_TAPS; ++t) {   // tap table unpacked from two scalar registers (no memory access)
        if (t >= p.T) break;                 // uniform: the 1-, 2- and 4-tap classes of a stride-2 data gradient stop early
        const int dyt = (int)((p.dy_pack >> (4 * t)) & 0xF) - 2, dxt = (int)((p.dx_pack >> (4 * t)) & 0xF) - 2;
        const bool ok = (unsigned)(iy0 + dyt) < (unsigned)p.Hin && (unsigned)(ix0 + dxt) < (unsigned)p.Win;
        vm |= ok ? (1u << t) : 0u;
      }
    }
    a_voff[i] = voff;
    a_valid[i] = vm;
  }
  if (PROF) asm volatile("s_nop 0" ::"v"(a_voff[0]), "v"(a_valid[A_PER - 1]));
  STAMP(t_p2);
  int b_voff[B_PER], b_instr[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    // when the B tile has fewer LDS-DMA pieces than waves, several waves write the same piece (same bytes)
    const int instr = (wid * B_PER + i) % B_INSTR;
    const int row = instr * R + lrow;
    b_instr[i] = instr;
    b_voff[i] = (row * Ktot + (cpos ^ swz<BK>(row)) * 8) * 2;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): s_toff written
  STAMP(t_p3);
  __builtin_amdgcn_s_barrier();
  STAMP(t_p4);

  // k-steps walk (tap, channel chunk); a split-K block walks only its channel range [c0b, c0b + cin_steps*BK) of every tap
  const int cin_steps = __builtin_amdgcn_readfirstlane((p.ksplit > 1 ? p.Cin / p.ksplit : p.Cin) / BK);
  const int c0b = sp * cin_steps * (BK * 2);                     // byte offset of the split's first channel (0 without split-K)
  const int ksteps = ABL == 5 ? 0 : p.T * cin_steps;            // ABL 5: prologue + epilogue only
  int pf_t = 0, pf_c = 0;   // (tap, cin-step) of the next stage to prefetch

  auto stage = [&](int s, int buf) {
    const int coff = c0b + pf_c * (BK * 2);
    const int soff = __builtin_amdgcn_readfirstlane(s_toff[pf_t]) + coff;
    const int boff = pf_t * (p.Cin * 2) + coff;                  // = s * BK * 2 without split-K
    char* sa = smem + buf * STAGE;
    char* sb = sa + BM * ROWB;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int instr = wid * A_PER + i;
      bufld16(rsrc_x, sa + instr * 1024, ((a_valid[i] >> pf_t) & 1u) ? a_voff[i] : OOB_VOFF, soff);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) bufld16(rsrc_w, sb + b_instr[i] * 1024, b_voff[i], boff);
    if (++pf_c == cin_steps) {
      pf_c = 0;
      ++pf_t;
    }
  };

  // one LDS-DMA piece of the stage being prefetched (q < A_PER: activation rows, else weight rows)
  auto issue_piece = [&](int q, int boff, int soff, int tap, int buf) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + BM * ROWB;
    if (q < A_PER) {
      const int instr = wid * A_PER + q;
      bufld16(rsrc_x, sa + instr * 1024, ((a_valid[q] >> tap) & 1u) ? a_voff[q] : OOB_VOFF, soff);
    } else {
      bufld16(rsrc_w, sb + b_instr[q - A_PER] * 1024, b_voff[q - A_PER], boff);
    }
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < ksteps && ABL != 1) stage(s, s);

  const int fr = lane & 15, fq = lane >> 4;
  int buf = 0;
  STAMP(t_loop);
  for (int s = 0; s < ksteps; ++s) {
    STAMP(t_a);
    // stage s has landed once at most min(D-1, ksteps-1-s) younger stages of THIS wave are still in flight
    const int younger = min(D - 1, ksteps - 1 - s);
    if (D >= 1 && younger >= D - 1) wait_vmcnt<(D - 1 > 0 ? D - 1 : 0) * PER>();
    else if (D >= 3 && younger == D - 2) wait_vmcnt<(D - 2 > 0 ? D - 2 : 0) * PER>();
    else if (D >= 4 && younger == D - 3) wait_vmcnt<(D - 3 > 0 ? D - 3 : 0) * PER>();
    else if (D >= 5 && younger == D - 4) wait_vmcnt<(D - 4 > 0 ? D - 4 : 0) * PER>();
    else if (D >= 6 && younger == D - 5) wait_vmcnt<(D - 5 > 0 ? D - 5 : 0) * PER>();
    else wait_vmcnt<0>();
    STAMP(t_b);
    __builtin_amdgcn_s_barrier();             // everyone's pieces of stage s landed; everyone left buffer (s-1)%NST
    STAMP(t_c);
    const bool pf = s + D < ksteps;
    const int pbuf = buf == 0 ? NST - 1 : buf - 1;
    int il_soff = 0, il_tap = 0, il_boff = 0;
    if (!ILV) {
      if (pf && ABL != 1) stage(s + D, pbuf);
    } else if (pf) {
      il_tap = pf_t;
      il_soff = __builtin_amdgcn_readfirstlane(s_toff[pf_t]) + c0b + pf_c * (BK * 2);
      il_boff = pf_t * (p.Cin * 2) + c0b + pf_c * (BK * 2);
      if (++pf_c == cin_steps) {
        pf_c = 0;
        ++pf_t;
      }
    }
    STAMP(t_d);
    const char* sa = smem + buf * STAGE;
    const char* sb = sa + BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      st16x8_t wf[TN], af[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn * (TN * 16) + i * 16 + fr;
        if (ABL == 2) { wf[i] = __builtin_bit_cast(st16x8_t, make_uint4(row, s, ks, i)); asm volatile("" : "+v"(wf[i])); }
        else wf[i] = *(const st16x8_t*)(sb + row * ROWB + (((ks * 4 + fq) ^ swz<BK>(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int row = wm * (TM * 16) + j * 16 + fr;
        if (ABL == 2) { af[j] = __builtin_bit_cast(st16x8_t, make_uint4(row, s, ks, j)); asm volatile("" : "+v"(af[j])); }
        else af[j] = *(const st16x8_t*)(sa + row * ROWB + (((ks * 4 + fq) ^ swz<BK>(row)) << 4));
      }
      constexpr int NMF = TN * TM * (BK / 32);            // MFMAs per k-step per wave
      constexpr int G = NMF / PER > 0 ? NMF / PER : 1;     // MFMAs between two LDS-DMA pieces
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          if (ABL == 3) { asm volatile("" ::"v"(wf[i]), "v"(af[j])); acc[i][j][0] += 1.0f; }
          else acc[i][j] = MI355_MFMA_16x16x32(wf[i], af[j], acc[i][j]);
          if (ILV) {
            // spread the next stage's LDS-DMA pieces between the MFMAs: a piece costs ~100+ issue cycles, which
            // then overlap with this wave's own MFMAs still running in the matrix pipe
            const int cnt = ks * TN * TM + i * TM + j + 1;
            if (cnt % G == 0 && cnt / G - 1 < PER) {
              if (pf) issue_piece(cnt / G - 1, il_boff, il_soff, il_tap, pbuf);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
    }
    buf = buf + 1 == NST ? 0 : buf + 1;
    if (PROF) {
      asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[TN - 1][TM - 1][3]));
      STAMP(t_e);
      c_wait += t_b - t_a;
      c_bar += t_c - t_b;
      c_issue += t_d - t_c;
      c_comp += t_e - t_d;
    }
  }
  if (PROF) {
    STAMP(t_e);
    if (lane == 0 && p.dbg) {
      unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wid) * 8;
      d[0] = c_wait; d[1] = c_bar; d[2] = c_issue; d[3] = c_comp; d[4] = t_loop - t_begin; d[5] = t_e - t_loop;
      d[6] = ((t_p1 - t_begin) << 32) | (t_p2 - t_p1);
      d[7] = ((t_p3 - t_p2) << 32) | (t_p4 - t_p3);
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();               // the epilogue reuses smem

  igemm_epilogue<WM, WN, TM, TN, EPI>(p, acc, smem, tid, NT, true, wm, wn, lane, mt, n0, m0, EPI == EPI_F32 ? sp * p.ysplit : 0ll);
}

// ---- 3x3 stride-1 form with shared pixel tiles ("dx reuse") ----------------------------------------------------------
// The implicit GEMM above stages one pixel tile per tap: nine per channel chunk, although the three taps of one kernel
// row read the SAME pixels shifted by one.  Measured (tools/micro/l2_lds_bw.hip): a CU stages ~73-89 GB/s from L2, and at
// 64 FLOP per staged byte that rate, not the MFMA pipe, bounds the 128x128 tile.  Here the loop is ordered
// (channel chunk, kernel row, dx): per kernel row ONE extended pixel tile (rows m0-8 .. m0+BM+7 of the linear lattice,
// shifted by dy*W) is staged and the three dx taps read it at row offsets dx; lanes whose pixel sits on the image's left /
// right edge get a zero fragment for dx = -1 / +1 (the neighbour in memory is the previous / next image row).  Pixel bytes
// per chunk drop 3x (staged bytes -31 % at 128x128).  Weights keep their own 2-deep ring (one tile per tap).
// vmcnt bookkeeping: per step the wave issues B(s+1) first, then its share (2,2,1 pieces) of the next group's pixel tile,
// so "all but the pieces issued after B(s)" is a compile-time count at each of the three unrolled positions.
template <int WM, int WN, int TM, int TN, int EPI, int NSTB = 2, bool PROF = false, int BK = 64>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN >= 8 || TM >= 8 || NSTB > 2 ? 1 : (BK == 32 ? 3 : 2))) void igemm_dx_kernel(const IgemmParams p) {
  unsigned long long t_a = 0, t_b = 0, t_c = 0, t_d = 0, t_e = 0, t_f = 0, c_wait = 0, c_bar = 0, c_issue = 0, c_read = 0, c_mfma = 0, t_begin = 0, t_loop = 0;
  STAMP(t_begin);
  constexpr int NT = WM * WN * 64, NW = WM * WN;
  static_assert(BK == 64 || BK == 32, "k-step of 64 or 32 channels");
  constexpr int D = NSTB - 1;                            // weight tiles in flight ahead of the one being consumed
  static_assert(NSTB == 2 || NSTB == 3, "weight ring depth 2 or 3");
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, ROWB = BK * 2, R = 1024 / ROWB, CPR = BK / 8, KS = BK / 32;
  constexpr int A_INSTR = BM / R + 2;                    // one LDS-DMA piece (R rows) of halo on each side
  constexpr int A_PER = (A_INSTR + NW - 1) / NW;         // pieces per wave per group (the surplus ones are dummies)
  constexpr int A_ROWS = A_PER * NW * R;
  constexpr int B_INSTR = BN / R, B_PER = B_INSTR / NW;
  static_assert(B_INSTR % NW == 0, "weight tile must split evenly over the waves");
  constexpr int AC0 = (A_PER + 2) / 3, AC1 = (A_PER + 1) / 3, AC2 = A_PER / 3;   // pieces issued at dx position 0, 1, 2
  constexpr int ABYTES = A_ROWS * ROWB, BBYTES = BN * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const abuf0 = smem;
  char* const bbuf0 = smem + 2 * ABYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid % WM, wn = wid / WM;
  const int ntn = p.CoutPad / BN, nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int mt = bid / ntn, nt = bid - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;

  const int n_first = (int)fdiv(fdiv((unsigned)max(m0 - R, 0), p.dMW), p.dMH);
  const int Ktot = 9 * p.Cin;
  const srd_t rsrc_x = make_srd(p.x + (long long)n_first * p.Hin * p.Win * p.ldin - p.tap_pad, 0x7FFFFFF0u);
  const srd_t rsrc_w = make_srd(p.w + (long long)n0 * Ktot, 0x7FFFFFF0u);

  // scalar tap geometry: kernel row g uses taps 3g..3g+2 (same dy); dx of position i is the same for every row
  int dyg[3], dxi[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    dyg[g] = (int)((p.dy_pack >> (12 * g)) & 0xF) - 2;
    dxi[g] = (int)((p.dx_pack >> (4 * g)) & 0xF) - 2;
  }

  // ---- pixel-tile rows this lane stages: extended row e <-> lattice pixel m0 - R + e
  const int lrow = lane / CPR, cpos = lane % CPR;
  int a_voff[A_PER];
  unsigned a_valid[A_PER];      // bit g: source row y + dy_g inside the image
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int instr = wid * A_PER + i;
    const int e = instr * R + lrow;
    const int m = m0 - R + e;
    unsigned vm = 0;
    int voff = OOB_VOFF;
    if (instr < A_INSTR && m >= 0 && m < p.M) {
      const int t1 = (int)fdiv((unsigned)m, p.dMW), xx = m - t1 * p.MW;
      const int n = (int)fdiv((unsigned)t1, p.dMH), yy = t1 - n * p.MH;
      voff = ((((n - n_first) * p.Hin + yy) * p.Win + xx) * p.ldin + (cpos ^ swz_shift<BK>(e)) * 8) * 2;
#pragma unroll
      for (int g = 0; g < 3; ++g) vm |= ((unsigned)(yy + dyg[g]) < (unsigned)p.Hin) ? (1u << g) : 0u;
    }
    a_voff[i] = voff;
    a_valid[i] = vm;
  }
  int b_voff[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int row = (wid * B_PER + i) * R + lrow;
    b_voff[i] = (row * Ktot + (cpos ^ swz<BK>(row)) * 8) * 2;
  }

  // ---- fragment read offsets: pixel fragments per dx position (row shift + its own swizzle phase).  A lane whose pixel has no
  // neighbour in direction dx_i (image edge: the neighbour in memory is the previous / next image row) reads a ZERO row instead:
  // the dummy pieces behind the tile are staged out of range, i.e. as zeros, in every group, so the edge costs no instruction in
  // the loop (it used to be 64 v_cndmask per three k-steps)
  static_assert(A_PER * NW > A_INSTR, "needs a dummy (zero) piece behind the pixel tile");
  const int fr = lane & 15, fq = lane >> 4;
  int afrag[3][TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int r = wm * (TM * 16) + j * 16 + fr;
    const int m = m0 + r;
    const int t1 = (int)fdiv((unsigned)m, p.dMW), xx = m - t1 * p.MW;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = r + R + dxi[i];
      const bool off = (dxi[i] < 0 && xx == 0) || (dxi[i] > 0 && xx == p.MW - 1);
      afrag[i][j] = off ? A_INSTR * R * ROWB + (fq << 4) : e * ROWB + ((fq ^ swz_shift<BK>(e)) << 4);
    }
  }
  int wfrag[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int row = wn * (TN * 16) + i * 16 + fr;
    wfrag[i] = row * ROWB + ((fq ^ swz<BK>(row)) << 4);
  }

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

  const int cin_steps = p.Cin / BK;
  const int NQ = 3 * cin_steps;             // groups: channel chunk outer, kernel row inner
  const int rowpitch2 = p.Win * p.ldin * 2; // bytes per image row

  // group q = (chunk c, kernel row g): scalar byte offset of its pixel tile; step (q, i): scalar offset of its weight tile
  auto a_soff = [&](int c, int g) { return (dyg[g] * p.Win * p.ldin + p.tap_pad) * 2 + c * (BK * 2); };
  auto issue_a = [&](int lo, int hi, int c, int g, char* ab) {
    const int soff = a_soff(c, g);
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      if (i >= lo && i < hi) bufld16(rsrc_x, ab + (wid * A_PER + i) * 1024, ((a_valid[i] >> g) & 1u) ? a_voff[i] : OOB_VOFF, soff);
  };
  auto issue_b = [&](int c, int g, int i, char* bb) {
    const int koff = ((3 * g + i) * p.Cin + c * BK) * 2;
#pragma unroll
    for (int q = 0; q < B_PER; ++q) bufld16(rsrc_w, bb + (wid * B_PER + q) * 1024, b_voff[q], koff);
  };
  (void)rowpitch2;
  // step index -> (chunk, kernel row, dx position) of the weight tile D steps ahead, kept incrementally
  const int NS = 3 * NQ;
  issue_a(0, A_PER, 0, 0, abuf0);
  int pc = 0, pg = 0, pi = 0, ps = 0;      // next weight tile to prefetch
  auto issue_next_b = [&]() {
    char* bb = bbuf0 + (ps % NSTB) * BBYTES;
    if (ps < NS) {
      issue_b(pc, pg, pi, bb);
    } else {                                 // past the end: keep the vmcnt bookkeeping uniform (zero fill of a free slot)
#pragma unroll
      for (int q = 0; q < B_PER; ++q) bufld16(rsrc_w, bb + (wid * B_PER + q) * 1024, OOB_VOFF, 0);
    }
    ++ps;
    if (++pi == 3) {
      pi = 0;
      if (++pg == 3) {
        pg = 0;
        ++pc;
      }
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) issue_next_b();

  int c = 0, g = 0;           // current group
  int sb = 0;                 // weight buffer of the current step
  for (int q = 0; q < NQ; ++q) {
    const bool has_next = q + 1 < NQ;
    int cn = c, gn = g + 1;   // next group
    if (gn == 3) {
      gn = 0;
      ++cn;
    }
    char* const ab = abuf0 + (q & 1) * ABYTES;
    char* const abn = abuf0 + ((q + 1) & 1) * ABYTES;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      STAMP(t_a);
      if (NSTB == 2) {
        // order per step: B(s+1), then this step's share of the next pixel tile -> those pieces may stay in flight
        if (i == 0 || !has_next) wait_vmcnt<0>();
        else if (i == 1) wait_vmcnt<AC0>();
        else wait_vmcnt<AC1>();
      } else {
        // order per step: share of the next pixel tile FIRST, then B(s+2): the youngest weight tile (and the pixel pieces issued
        // one step ago) may stay in flight; at i == 0 the whole pixel tile of this group must have landed
        if (i == 0) wait_vmcnt<B_PER>();
        else if (i == 1) { if (has_next) wait_vmcnt<B_PER + AC0>(); else wait_vmcnt<B_PER>(); }
        else { if (has_next) wait_vmcnt<B_PER + AC1>(); else wait_vmcnt<B_PER>(); }
      }
      STAMP(t_b);
      __builtin_amdgcn_s_barrier();
      STAMP(t_c);
      char* const bb = bbuf0 + sb * BBYTES;
      if (NSTB == 2) issue_next_b();
      if (has_next) {
        if (i == 0) issue_a(0, AC0, cn, gn, abn);
        else if (i == 1) issue_a(AC0, AC0 + AC1, cn, gn, abn);
        else issue_a(AC0 + AC1, A_PER, cn, gn, abn);
      }
      if (NSTB != 2) issue_next_b();
      STAMP(t_d);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        st16x8_t wf[TN], af[TM];
#pragma unroll
        for (int t = 0; t < TN; ++t) wf[t] = *(const st16x8_t*)(bb + (wfrag[t] ^ (ks << 6)));
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          af[j] = *(const st16x8_t*)(ab + (afrag[i][j] ^ (ks << 6)));
        }
        if (PROF) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // diagnostic build: separate the fragment reads from the MFMAs
          STAMP(t_e);
          if (ks == 0) c_read += t_e - t_d; else c_read += t_e - t_f;
        }
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[t][j] = MI355_MFMA_16x16x32(wf[t], af[j], acc[t][j]);
        if (PROF) {
          asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[TN - 1][TM - 1][3]));
          STAMP(t_f);
          c_mfma += t_f - t_e;
        }
      }
      if (PROF) {
        c_wait += t_b - t_a;
        c_bar += t_c - t_b;
        c_issue += t_d - t_c;
      }
      sb = sb + 1 == NSTB ? 0 : sb + 1;
    }
    c = cn;
    g = gn;
  }
  wait_vmcnt<0>();            // the uniform-count dummy pieces of the last steps
  if (PROF) {
    STAMP(t_loop);
    if (lane == 0 && p.dbg) {
      unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wid) * 8;
      d[0] = c_wait; d[1] = c_bar; d[2] = c_issue; d[3] = c_read; d[4] = c_mfma; d[5] = t_loop - t_begin; d[6] = NQ * 3; d[7] = 0;
    }
  }
  (void)AC2;
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();               // the epilogue reuses smem
  igemm_epilogue<WM, WN, TM, TN, EPI>(p, acc, smem, tid, NT, true, wm, wn, lane, mt, n0, m0);
}

// Interleaved variant (2-buffer ring): all fragment reads of a k-step are issued up front (the second half's LDS
// latency hides under the first half's MFMAs) and the next stage's LDS-DMA pieces are issued BRANCH-FREE between
// groups of MFMAs (past the last stage the pieces carry an out-of-range offset and just zero-fill the idle buffer),
// so a piece's ~100-cycle issue stall overlaps with this wave's own MFMAs still executing in the matrix pipe.
template <int WM, int WN, int TM, int TN, int BK, int EPI, int SCHED>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN >= 8 ? 2 : 1)) void igemm_il_kernel(const IgemmParams p) {
  constexpr int NT = WM * WN * 64, NW = WM * WN, NST = 2;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int ROWB = BK * 2, R = 1024 / ROWB, CPR = BK / 8, KS = BK / 32;
  constexpr int A_INSTR = BM / R, B_INSTR = BN / R;
  constexpr int A_PER = A_INSTR / NW, B_PER = (B_INSTR + NW - 1) / NW, PER = A_PER + B_PER;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int NMF = TN * TM * KS, G = NMF / PER;
  static_assert(A_INSTR % NW == 0 && G >= 1, "tile must split evenly");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_toff = (int*)(smem + NST * STAGE);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid % WM, wn = wid / WM;
  const int ntn = p.CoutPad / BN, nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int mt = bid / ntn, nt = bid - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  if (tid == 0) {
#pragma unroll
    for (int t = 0; t < MAX_TAPS; ++t)
      if (t < p.T) s_toff[t] = ((p.dy[t] * p.Win + p.dx[t]) * p.ldin + p.tap_pad) * 2;
  }
  const int n_first = (int)fdiv(fdiv((unsigned)m0, p.dMW), p.dMH);
  const int Ktot = p.T * p.Cin;
  const srd_t rsrc_x = make_srd(p.x + (long long)n_first * p.Hin * p.Win * p.ldin - p.tap_pad, 0x7FFFFFF0u);
  const srd_t rsrc_w = make_srd(p.w + (long long)n0 * Ktot, 0x7FFFFFF0u);
  const int lrow = lane / CPR, cpos = lane % CPR;
  int a_voff[A_PER], b_voff[B_PER], b_instr[B_PER];
  unsigned a_valid[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int row = (wid * A_PER + i) * R + lrow;
    const int m = m0 + row;
    unsigned vm = 0;
    int voff = OOB_VOFF;
    if (m < p.M) {
      const int t1 = (int)fdiv((unsigned)m, p.dMW), xx = m - t1 * p.MW;
      const int n = (int)fdiv((unsigned)t1, p.dMH), yy = t1 - n * p.MH;
      const int iy0 = yy * p.sin, ix0 = xx * p.sin;
      voff = ((((n - n_first) * p.Hin + iy0) * p.Win + ix0) * p.ldin + (cpos ^ swz<BK>(row)) * 8) * 2;
#pragma unroll
      for (int t = 0; t < MAX_TAPS; ++t) {
        const int dyt = (int)((p.dy_pack >> (4 * t)) & 0xF) - 2, dxt = (int)((p.dx_pack >> (4 * t)) & 0xF) - 2;
        const bool ok = t < p.T && (unsigned)(iy0 + dyt) < (unsigned)p.Hin && (unsigned)(ix0 + dxt) < (unsigned)p.Win;
        vm |= ok ? (1u << t) : 0u;
      }
    }
    a_voff[i] = voff;
    a_valid[i] = vm;
  }
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int instr = (wid * B_PER + i) % B_INSTR;
    const int row = instr * R + lrow;
    b_instr[i] = instr;
    b_voff[i] = (row * Ktot + (cpos ^ swz<BK>(row)) * 8) * 2;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  const int ksteps = Ktot / BK, cin_steps = p.Cin / BK;
  int pf_t = 0, pf_c = 0;

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

  // one piece q of stage s into buffer `buf`; `live` false -> out-of-range offsets (zero fill, no memory traffic)
  auto piece = [&](int q, int s, int soff, int tap, int buf, bool live) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + BM * ROWB;
    if (q < A_PER) {
      const bool ok = live && ((a_valid[q] >> tap) & 1u);
      bufld16(rsrc_x, sa + (wid * A_PER + q) * 1024, ok ? a_voff[q] : OOB_VOFF, soff);
    } else {
      bufld16(rsrc_w, sb + b_instr[q - A_PER] * 1024, live ? b_voff[q - A_PER] : OOB_VOFF, s * (BK * 2));
    }
  };
  {
    const int soff = __builtin_amdgcn_readfirstlane(s_toff[0]);
#pragma unroll
    for (int q = 0; q < PER; ++q) piece(q, 0, soff, 0, 0, true);
    if (++pf_c == cin_steps) {
      pf_c = 0;
      ++pf_t;
    }
  }
  const int fr = lane & 15, fq = lane >> 4;
  for (int s = 0; s < ksteps; ++s) {
    const int buf = s & 1;
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    const bool live = s + 1 < ksteps;
    const int tap = live ? pf_t : 0;
    const int soff = __builtin_amdgcn_readfirstlane(s_toff[tap]) + pf_c * (BK * 2);
    if (++pf_c == cin_steps) {
      pf_c = 0;
      ++pf_t;
    }
    const char* sa = smem + buf * STAGE;
    const char* sb = sa + BM * ROWB;
    st16x8_t wf[KS][TN], af[KS][TM];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn * (TN * 16) + i * 16 + fr;
        wf[ks][i] = *(const st16x8_t*)(sb + row * ROWB + (((ks * 4 + fq) ^ swz<BK>(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int row = wm * (TM * 16) + j * 16 + fr;
        af[ks][j] = *(const st16x8_t*)(sa + row * ROWB + (((ks * 4 + fq) ^ swz<BK>(row)) << 4));
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          acc[i][j] = MI355_MFMA_16x16x32(wf[ks][i], af[ks][j], acc[i][j]);
          const int cnt = (ks * TN + i) * TM + j + 1;
          if (cnt % G == 0 && cnt / G - 1 < PER) piece(cnt / G - 1, s + 1, soff, tap, buf ^ 1, live);
        }
    if (SCHED == 1) {
      // ask the scheduler for: all DS reads first, then G MFMAs / 1 LDS-DMA piece alternating
      __builtin_amdgcn_sched_group_barrier(0x100, KS * (TN + TM), 0);
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x8, G, 0);
        __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  igemm_epilogue<WM, WN, TM, TN, EPI>(p, acc, smem, tid, NT, true, wm, wn, lane, mt, n0, m0);
}

// ------------------------------------------------------------------------------------------
// weight packing: fp32 master [cout][cin][k][k] -> bf16 [cout_pad][tap][cin]  (K contiguous)
__global__ void pack_fwd_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cout, int cout_pad, int cin, int cin_pad, int kk,
                                long long s_co, long long s_ci, long long s_t) {
  const long long total = (long long)cout_pad * kk * cin_pad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cin_pad);
    const int t = (int)((i / cin_pad) % kk);
    const int co = (int)(i / ((long long)cin_pad * kk));
    float v = 0.f;
    if (co < cout && c < cin) v = w[co * s_co + c * s_ci + t * s_t];
    out[i] = f2s(v);
  }
}

// dgrad pack: rows = input channel ci, K = (tap list) x cout.  taps[j] gives the forward tap index
// (kh*k+kw) whose weights feed dgrad tap j.  out [cin_pad][ntaps][cout]
__global__ void pack_dgrad_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cout, int cin, int cin_pad, int kk, int ntaps,
                                  long long s_co, long long s_ci, long long s_t, int t0, int t1, int t2, int t3, int t4, int t5, int t6,
                                  int t7, int t8) {
  const int taps[9] = {t0, t1, t2, t3, t4, t5, t6, t7, t8};
  const long long total = (long long)cin_pad * ntaps * cout;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i % cout);
    const int j = (int)((i / cout) % ntaps);
    const int ci = (int)(i / ((long long)cout * ntaps));
    int tap = taps[0];
#pragma unroll
    for (int q = 1; q < 9; ++q)
      if (j == q) tap = taps[q];
    float v = 0.f;
    if (ci < cin) v = w[co * s_co + ci * s_ci + tap * s_t];
    out[i] = f2s(v);
  }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dw, float* __restrict__ grad, int cout, int cin, int cin_pad, int kk) {
  const long long total = (long long)cout * cin * kk;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % kk);
    const int c = (int)((i / kk) % cin);
    const int co = (int)(i / ((long long)kk * cin));
    grad[i] = dw[((long long)co * kk + t) * cin_pad + c];
  }
}

// stem im2col: NCHW fp32 image [n,3,h,w] -> [n*h*w][32] bf16 rows, k = (kh*3+kw)*3 + c for k<27, zeros after
// (darknet.py:41 conv1 3->32 3x3 s1 p1; K=27 is below the MFMA K granularity, so the stem runs as a
//  1x1 conv over this 32-wide matrix, which forward and wgrad share).
// One workgroup = 128 pixels of one image row: the 3 x 3 input rows (channel planes x kernel rows) are staged through LDS with
// coalesced loads, and every thread assembles 16-byte pieces so that a store instruction writes 4 KB of consecutive output (the
// pixel-per-thread form spent three 64-bit divisions per pixel and wrote 16-byte pieces at a 64-byte stride: 2.1 TB/s).
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int n, int h, int w, int segs) {
  __shared__ float tile[9][132];   // [kh*3 + c][x - x0 + 1]
  const int seg = blockIdx.x % segs, row = blockIdx.x / segs;
  const int y = row % h, b = row / h, x0 = seg * 128;
  for (int e = threadIdx.x; e < 9 * 130; e += 256) {
    const int r = e / 130, xx = e - r * 130;
    const int kh = r / 3, c = r - kh * 3;
    const int iy = y + kh - 1, ix = x0 + xx - 1;
    float v = 0.f;
    if (iy >= 0 && iy < h && ix >= 0 && ix < w) v = img[((long long)(b * 3 + c) * h + iy) * w + ix];
    tile[r][xx] = v;
  }
  __syncthreads();
  const int q = threadIdx.x & 3;
  int lrow[8], lcol[8];            // LDS coordinates of this thread's eight k values (k = (kh*3+kw)*3 + c), -1: zero padding of K
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = q * 8 + j;
    const int t = k / 3, c = k - t * 3, kh = t / 3, kw = t - kh * 3;
    lrow[j] = k < 27 ? kh * 3 + c : -1;
    lcol[j] = kw;
  }
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int p = pass * 64 + (threadIdx.x >> 2);
    const int x = x0 + p;
    if (x >= w) continue;
    unsigned short v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = lrow[j] >= 0 ? f2s(tile[lrow[j]][p + lcol[j]]) : (unsigned short)0;
    uint4 u;
    u.x = v[0] | ((unsigned)v[1] << 16);
    u.y = v[2] | ((unsigned)v[3] << 16);
    u.z = v[4] | ((unsigned)v[5] << 16);
    u.w = v[6] | ((unsigned)v[7] << 16);
    *(uint4*)(out + ((long long)row * w + x) * 32 + q * 8) = u;
  }
}


// ---- class-concatenated form of the 3x3 / stride-2 data gradient ("s2cat") ------------------------------------------------------------
// The four output-parity classes of a stride-2 data gradient read the SAME dy lattice pixels (offsets oy, ox in {0, 1}) with different taps.
// Run as four implicit GEMMs they have N = Cin (64 for the 64->128 layer: narrow tiles), K = 1, 2, 2, 4 taps and four launches, and
// they take 2-3x the forward time of the same layer.  Concatenating the classes of one output-row parity py along N gives two GEMMs
//   py = 0:  N = (px, ci) = 2*Cin,  K = (ox, co)     = 2*Cout      dx rows 2*yy
//   py = 1:  N = (px, ci) = 2*Cin,  K = (oy, ox, co) = 4*Cout      dx rows 2*yy + 1
// whose output row for lattice pixel (yy, xx) is 2*Cin CONTIGUOUS values - dx pixels (2xx, 2xx+1) - so the ordinary epilogue writes it
// with y stride 2 and x stride 1 on the view [n, 2Ho, Wo, 2*Cin].  The weight matrix has zero blocks where a class does not use an
// offset (px = 0 never looks at ox = 1, py = 0 never at oy = 1): 12 blocks are multiplied for 9 that carry weights (1.33x the FLOPs), at the
// rate of the wide tiles instead of the narrow ones.  Chosen per shape by the autotuner against the four class launches.
__host__ __device__ inline int s2cat_k(int parity, int off) {      // kernel row / column a class of this parity reads at lattice offset `off`, or -1
  return parity == 0 ? (off == 0 ? 1 : -1) : (off == 0 ? 2 : 0);
}
static bool s2cat_eligible(const mi355det_conv_shape* s) {
  return s->ksize == 3 && s->stride == 2 && s->pad == 1 && !(s->h & 1) && !(s->w & 1) && (s->cin == 64 || s->cin == 128) && s->cout % 64 == 0 &&
         s->in_ld == s->cin;
}
static size_t s2cat_elems(const mi355det_conv_shape* s) { return s2cat_eligible(s) ? (size_t)12 * s->cin * s->cout : 0; }
static size_t dgrad_class_elems(const mi355det_conv_shape* s) { return (size_t)((s->cin + 31) / 32 * 32) * (size_t)(s->ksize * s->ksize) * (size_t)s->cout; }

// out: [py = 0: [2*cin][2][cout]] then [py = 1: [2*cin][4][cout]]
__global__ void pack_s2cat_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cout, int cin, long long s_co, long long s_ci, long long s_t) {
  const long long n0 = (long long)2 * cin * 2 * cout, total = n0 + (long long)2 * cin * 4 * cout;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int py = i >= n0;
    const long long j = py ? i - n0 : i;
    const int T = py ? 4 : 2;
    const int co = (int)(j % cout), tap = (int)((j / cout) % T), r = (int)(j / ((long long)cout * T));
    const int px = r / cin, ci = r - px * cin;
    const int oy = py ? tap >> 1 : 0, ox = py ? tap & 1 : tap;
    const int kh = s2cat_k(py, oy), kw = s2cat_k(px, ox);
    float v = 0.f;
    if (kh >= 0 && kw >= 0) v = w[co * s_co + ci * s_ci + (kh * 3 + kw) * s_t];
    out[i] = f2s(v);
  }
}

bf16_t* g_zero_page = nullptr;

int ensure_zero_page() {
  if (g_zero_page) return 0;
  void* p = nullptr;
  if (hipMalloc(&p, 4096) != hipSuccess) return fail(MI355DET_ELAUNCH, "%s: hipMalloc(zero page) failed", "conv");
  if (hipMemset(p, 0, 4096) != hipSuccess) return fail(MI355DET_ELAUNCH, "%s: hipMemset(zero page) failed", "conv");
  g_zero_page = (bf16_t*)p;
  return 0;
}

unsigned long long* g_dbg = nullptr;

template <int WM, int WN, int TM, int TN, int BK, int NST, int EPI, bool PROF = false, bool ILV = false, int OCC = 0, int ABL = 0>
int launch_cfg(const IgemmParams& p_in, hipStream_t st) {
  IgemmParams p = p_in;
  p.dbg = PROF ? g_dbg : nullptr;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int lds_ring = NST * (BM + BN) * BK * 2 + 64;
  constexpr int lds_epi = EPI_LDS_OFF + WM * WN * (64 * (TN * 32 + 16) + 256);
  constexpr int lds = lds_ring > lds_epi ? lds_ring : lds_epi;
  const int gm = (p.M + BM - 1) / BM, gn = p.CoutPad / BN;
  auto k = igemm_kernel<WM, WN, TM, TN, BK, NST, EPI, PROF, ILV, OCC, ABL>;
  static DeviceOnce attr_done;
  attr_done.once([&] {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  hipLaunchKernelGGL(k, dim3(gm * gn * (p.ksplit > 1 ? p.ksplit : 1)), dim3(WM * WN * 64), lds, st, p);
  return check_launch("igemm");
}

template <int WM, int WN, int TM, int TN, int BK, int EPI, int SCHED>
int launch_il(const IgemmParams& p, hipStream_t st) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int lds_ring = 2 * (BM + BN) * BK * 2 + 64;
  constexpr int lds_epi = EPI_LDS_OFF + WM * WN * (64 * (TN * 32 + 16) + 256);
  constexpr int lds = lds_ring > lds_epi ? lds_ring : lds_epi;
  const int gm = (p.M + BM - 1) / BM, gn = p.CoutPad / BN;
  auto k = igemm_il_kernel<WM, WN, TM, TN, BK, EPI, SCHED>;
  static DeviceOnce attr_done;
  attr_done.once([&] {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  hipLaunchKernelGGL(k, dim3(gm * gn), dim3(WM * WN * 64), lds, st, p);
  return check_launch("igemm_il");
}

// a pair of timing events destroyed on every exit path (the tuning helpers return early on launch errors)
struct EventPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool ok = false;
  EventPair() {
    if (hipEventCreate(&e0) != hipSuccess) { e0 = nullptr; return; }
    if (hipEventCreate(&e1) != hipSuccess) { e1 = nullptr; return; }
    ok = true;
  }
  ~EventPair() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  }
  EventPair(const EventPair&) = delete;
  EventPair& operator=(const EventPair&) = delete;
};

// Time `fn` for the tuning helpers: one warm-up launch, then the FASTER of two batches of three launches (a single batch of three let a
// neighbour's burst or a cold L2 decide: round 4 saw the tuner keep a 128 us configuration for 256->512 s2 @40 where the same kernel list
// held one of 106 us).  Returns milliseconds per batch, < 0 on a launch error (code in *err).
template <class F>
float time_candidate(F&& fn, hipEvent_t e0, hipEvent_t e1, hipStream_t st, int* err) {
  *err = fn();
  if (*err) return -1.f;
  float best = 1e30f;
  for (int b = 0; b < 2; ++b) {
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < 3; ++r) (void)fn();
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int g_tune = 0;   // bring-up knob (mi355det_debug_set(0, v)): forces a tile configuration
TuneMap& g_igemm_tuned = tune_table(TUNE_IGEMM);   // shape key -> configuration found by mi355det_conv_autotune (part of the tune record)

unsigned long long igemm_key(const IgemmParams& p, int epi) {
  unsigned long long k = (unsigned long long)p.M;
  k = k * 4099 + p.CoutPad;
  k = k * 4099 + p.Cin;
  k = k * 31 + p.T;
  k = k * 7 + p.sin * 2 + p.so;
  k = k * 3 + p.sox;
  k = k * 5 + epi;
  k = k * 2 + MI355_F16;      // the fp16-storage objects keep their own choices in the shared tune record
  return k;
}

// applicability of the dx-reuse kernel: 3x3, unit strides on both sides, taps in three rows of equal dy with dx stepping by +-1
static bool dx_applicable(const IgemmParams& p, int bn = 128, int bk = 64) {
  if (p.T != 9 || p.so != 1 || p.sox != 0 || p.sin != 1 || p.oy0 != 0 || p.ox0 != 0) return false;
  if (p.Hin != p.Hout || p.Win != p.Wout || p.MH != p.Hin || p.MW != p.Win) return false;
  if (p.Cin % bk != 0 || p.CoutPad % bn != 0) return false;
  for (int g = 0; g < 3; ++g) {
    if (p.dy[3 * g] != p.dy[3 * g + 1] || p.dy[3 * g] != p.dy[3 * g + 2]) return false;
    for (int i = 0; i < 3; ++i)
      if (p.dx[3 * g + i] != p.dx[i] || p.dx[i] < -1 || p.dx[i] > 1) return false;
    if (p.dy[3 * g] < -1 || p.dy[3 * g] > 1) return false;
  }
  return p.dx[0] != p.dx[1] && p.dx[1] != p.dx[2] && p.dx[0] != p.dx[2];
}

template <int WM, int WN, int TM, int TN, int EPI, int NSTB = 2, bool PROF = false, int BK = 64>
int launch_dx(const IgemmParams& p_in, hipStream_t st) {
  IgemmParams p = p_in;
  p.dbg = PROF ? g_dbg : nullptr;
  if (!dx_applicable(p, WN * TN * 16, BK)) return fail(MI355DET_EINVAL, "%s: shape not supported by the dx-reuse kernel", "igemm_dx");
  constexpr int NW = WM * WN, BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int R = 1024 / (BK * 2);
  constexpr int A_PER = (BM / R + 2 + NW - 1) / NW;
  constexpr int lds_ring = 2 * A_PER * NW * 1024 + NSTB * BN * BK * 2;
  constexpr int lds_epi = EPI_LDS_OFF + WM * WN * (64 * (TN * 32 + 16) + 256);
  constexpr int lds = lds_ring > lds_epi ? lds_ring : lds_epi;
  const int gm = (p.M + BM - 1) / BM, gn = p.CoutPad / BN;
  auto k = igemm_dx_kernel<WM, WN, TM, TN, EPI, NSTB, PROF, BK>;
  static DeviceOnce attr_done;
  attr_done.once([&] {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  hipLaunchKernelGGL(k, dim3(gm * gn), dim3(WM * WN * 64), lds, st, p);
  return check_launch("igemm_dx");
}

// tile configurations for Cout % 128 == 0 and Cin % 64 == 0 (pixels x channels x k-step, ring depth):
//   1: 128x128x64 x2, 4 waves of 64x64, 2 workgroups/CU        2: 256x128x32 x2, 4 waves of 128x64, 48 KB
//   3: 256x256x64 x2, 8 waves of 128x64, 1 workgroup/CU        4: 128x128x32 x3, 4 waves of 64x64, 3 workgroups/CU
//   5: 128x128x32 x2 at 4 workgroups/CU                         6: 256x256x64 with LDS-DMA pieces interleaved between MFMAs
template <int EPI>
int run_cfg(int cfg, const IgemmParams& p, hipStream_t st) {
  switch (cfg) {
    case 2: return launch_cfg<2, 2, 8, 4, 32, 2, EPI>(p, st);
    case 3: if (p.CoutPad % 256 == 0) return launch_cfg<2, 4, 8, 4, 64, 2, EPI>(p, st); break;
    case 4: return launch_cfg<2, 2, 4, 4, 32, 3, EPI>(p, st);
    case 5: return launch_cfg<2, 2, 4, 4, 32, 2, EPI, false, false, 4>(p, st);   // 32 KB LDS, <=128 VGPR: 4 workgroups/CU
    case 6: if (p.CoutPad % 256 == 0) return launch_il<2, 4, 8, 4, 64, EPI, 0>(p, st); break;   // interleaved 256x256x64, 8 waves of 128x64
    // 50-56: deeper rings, diagnostic (forward only): measured slower than depth 2 once the LDS-DMA really stays in flight
    case 50: if (EPI == EPI_STATS && p.CoutPad % 256 == 0) return launch_cfg<2, 4, 8, 4, 32, 3, EPI>(p, st); break;   // 256x256x32 ring 3 (96 KB)
    case 51: if (EPI == EPI_STATS && p.CoutPad % 256 == 0) return launch_cfg<2, 4, 8, 4, 32, 4, EPI>(p, st); break;   // 256x256x32 ring 4 (128 KB)
    case 52: if (EPI == EPI_STATS) return launch_cfg<2, 2, 8, 4, 32, 3, EPI>(p, st); break;                                     // 256x128x32 ring 3 (72 KB, 2 workgroups/CU)
    case 53: if (EPI == EPI_STATS) return launch_cfg<2, 2, 8, 4, 32, 4, EPI>(p, st); break;                                     // 256x128x32 ring 4 (96 KB)
    case 54: if (EPI == EPI_STATS) return launch_cfg<2, 2, 4, 4, 32, 4, EPI>(p, st); break;                                     // 128x128x32 ring 4 (64 KB, 2 workgroups/CU)
    case 55: if (EPI == EPI_STATS) return launch_cfg<2, 2, 4, 4, 64, 3, EPI>(p, st); break;                                     // 128x128x64 ring 3 (96 KB, 1 workgroup/CU)
    case 56: if (EPI == EPI_STATS) return launch_cfg<2, 2, 4, 4, 32, 6, EPI>(p, st); break;                                     // 128x128x32 ring 6 (96 KB)
    case 40: if (igemm8_applicable(p) && (EPI == EPI_STATS || EPI == EPI_PLAIN || EPI == EPI_RES || EPI == EPI_AFF || EPI == EPI_F32)) return igemm8_launch(EPI, p, st); break;   // phase-staggered 256x256x64, 8 waves
    // 44 / 45: the same kernel on 224 / 208-pixel tiles (tile quantisation: 800 / 400 / 200 tiles of 256 pixels on 256 CUs; a 192-pixel
    // tile was measured too - 1067 / 534 / 268 tiles need one round more: 3-65 % slower - and removed, profiles/r04_ab_results.md)
    case 44: if (igemm8_applicable(p) && (EPI == EPI_STATS || EPI == EPI_PLAIN || EPI == EPI_RES || EPI == EPI_AFF)) return igemm8_launch(EPI, p, st, 224); break;
    case 45: if (igemm8_applicable(p) && (EPI == EPI_STATS || EPI == EPI_PLAIN || EPI == EPI_RES || EPI == EPI_AFF)) return igemm8_launch(EPI, p, st, 208); break;
    case 15: if (dx_applicable(p)) return launch_dx<2, 2, 4, 4, EPI>(p, st); break;             // 3x3 s1: shared pixel tiles (dx reuse), 128x128
    case 16: if (dx_applicable(p)) return launch_dx<4, 2, 4, 4, EPI>(p, st); break;             // dx reuse 256x128, 8 waves, 1 workgroup/CU
    case 17: if (dx_applicable(p) && p.CoutPad % 256 == 0) return launch_dx<2, 4, 4, 4, EPI>(p, st); break;   // dx reuse 128x256, 8 waves
    case 18: if (dx_applicable(p) && p.CoutPad % 256 == 0) return launch_dx<2, 4, 8, 4, EPI>(p, st); break;   // dx reuse 256x256, 8 waves of 128x64
    case 19: if (dx_applicable(p)) return launch_dx<2, 2, 8, 4, EPI>(p, st); break;                           // dx reuse 256x128, 4 waves of 128x64
    case 35: if (dx_applicable(p)) return launch_dx<2, 2, 4, 4, EPI, 2, false, 32>(p, st); break;            // dx reuse 128x128x32: 3 workgroups/CU
    case 26: if (dx_applicable(p)) return launch_dx<4, 2, 4, 4, EPI, 3>(p, st); break;                        // dx reuse 256x128, 8 waves, weight ring 3
    case 27: if (dx_applicable(p)) return launch_dx<2, 2, 4, 4, EPI, 3>(p, st); break;                        // dx reuse 128x128, weight ring 3 (1 WG/CU)
    case 28: if (dx_applicable(p) && p.CoutPad % 256 == 0) return launch_dx<2, 4, 4, 4, EPI, 3>(p, st); break; // dx reuse 128x256, 8 waves, ring 3
    case 21: if (EPI == EPI_STATS) return launch_cfg<2, 2, 4, 4, 64, 2, EPI, false, false, 0, 1>(p, st); break;   // ablations of cfg 1
    case 22: if (EPI == EPI_STATS) return launch_cfg<2, 2, 4, 4, 64, 2, EPI, false, false, 0, 2>(p, st); break;
    case 23: if (EPI == EPI_STATS) return launch_cfg<2, 2, 4, 4, 64, 2, EPI, false, false, 0, 3>(p, st); break;
    case 25: if (EPI == EPI_STATS) return launch_cfg<2, 2, 4, 4, 64, 2, EPI, false, false, 0, 5>(p, st); break;
    case 99: if (EPI == EPI_STATS) return launch_cfg<2, 2, 4, 4, 64, 2, EPI, true>(p, st); break;   // phase-stamp diagnostic build
    case 98: if (EPI == EPI_STATS && dx_applicable(p)) return launch_dx<2, 2, 4, 4, EPI, 2, true>(p, st); break;   // same for the shared-pixel-tile kernel
    case 97: if (EPI == EPI_STATS && dx_applicable(p)) return launch_dx<4, 2, 4, 4, EPI, 3, true>(p, st); break;   // 256x128, 8 waves, ring 3
    default: break;
  }
  return launch_cfg<2, 2, 4, 4, 64, 2, EPI>(p, st);
}

template <int EPI>
int launch_igemm(const IgemmParams& p, hipStream_t st) {
  if (p.Cin % 32 != 0) return fail(MI355DET_EINVAL, "%s: Cin must be a multiple of 32 (got %lld)", "conv", p.Cin);
  const bool k64 = p.Cin % 64 == 0;
  if (p.CoutPad % 128 == 0 && k64) {
    int cfg = 1;
    auto it = g_igemm_tuned.find(igemm_key(p, EPI));
    if (it != g_igemm_tuned.end()) cfg = it->second;
    if (g_tune) cfg = g_tune;
    return run_cfg<EPI>(cfg, p, st);
  }
  if (p.CoutPad % 128 == 0) return launch_cfg<2, 2, 4, 4, 32, 3, EPI>(p, st);
  // narrow outputs (32 / 64 channels: the first layers and their data gradients): one plain tile shape each, or the
  // shared-pixel-tile kernel at 256 x 64 / 256 x 32 when the autotuner found it faster
  int narrow = 0;
  {
    auto it = g_igemm_tuned.find(igemm_key(p, EPI));
    if (it != g_igemm_tuned.end()) narrow = it->second;
    if (g_tune == 29 || g_tune == 30 || g_tune == 31) narrow = g_tune;
  }
  if (p.CoutPad % 64 == 0) {
    if (narrow == 30 && dx_applicable(p, 64)) return launch_dx<4, 1, 4, 4, EPI>(p, st);
    if (narrow == 31 && dx_applicable(p, 64, 32)) return launch_dx<4, 1, 4, 4, EPI, 2, false, 32>(p, st);      // 32 input channels: k-step of 32
    return k64 ? launch_cfg<4, 1, 4, 4, 64, 2, EPI>(p, st) : launch_cfg<4, 1, 4, 4, 32, 3, EPI>(p, st);
  }
  if (p.CoutPad % 32 == 0) {
    if (narrow == 29 && dx_applicable(p, 32)) return launch_dx<4, 1, 4, 2, EPI>(p, st);
    return k64 ? launch_cfg<4, 1, 4, 2, 64, 2, EPI>(p, st) : launch_cfg<4, 1, 4, 2, 32, 4, EPI>(p, st);
  }
  return fail(MI355DET_EINVAL, "%s: padded Cout must be a multiple of 32 (got %lld)", "conv", p.CoutPad);
}

// plan-build helper: time the candidate configurations of one launch and remember the fastest
template <int EPI>
int autotune_igemm(const IgemmParams& p, hipStream_t st) {
  if (tune_locked_has(TUNE_IGEMM, igemm_key(p, EPI))) return launch_igemm<EPI>(p, st);      // the choice came from a tune record: not timed again
  const bool narrow32 = p.CoutPad % 128 != 0 && p.CoutPad % 64 == 0 && p.Cin % 64 != 0 && dx_applicable(p, 64, 32);      // 32 -> 64 @320
  if (narrow32 || (p.CoutPad % 128 != 0 && p.Cin % 64 == 0 && (p.CoutPad % 64 == 0 ? dx_applicable(p, 64) : dx_applicable(p, 32)))) {
    // narrow output: plain tile (id 0) against the shared-pixel-tile kernel (id 30 / 29; 31 = its 32-deep k-step for 32 input channels)
    const int alt = narrow32 ? 31 : (p.CoutPad % 64 == 0 ? 30 : 29);
    EventPair ev;
    if (!ev.ok) return fail(MI355DET_ELAUNCH, "%s: event create failed", "conv_autotune");
    hipEvent_t e0 = ev.e0, e1 = ev.e1;
    float best_ms = 1e30f;
    int best = 0;
    for (int cfg : {0, alt}) {
      g_igemm_tuned[igemm_key(p, EPI)] = cfg;
      int e = 0;
      const float ms = time_candidate([&] { return launch_igemm<EPI>(p, st); }, e0, e1, st, &e);
      if (e) return e;
      if (ms < best_ms) {
        best_ms = ms;
        best = cfg;
      }
    }
    g_igemm_tuned[igemm_key(p, EPI)] = best;
    tune_mark_timed(TUNE_IGEMM, igemm_key(p, EPI));
    return best;
  }
  if (!(p.CoutPad % 128 == 0 && p.Cin % 64 == 0)) return launch_igemm<EPI>(p, st);   // nothing to choose: a plain launch, the output stays valid
  EventPair ev;
  if (!ev.ok) return fail(MI355DET_ELAUNCH, "%s: event create failed", "conv_autotune");
  hipEvent_t e0 = ev.e0, e1 = ev.e1;
  int best = 1;
  float best_ms = 1e30f;
  const int cands[] = {1, 2, 3, 4, 5, 6, 15, 16, 17, 18, 19, 26, 27, 28, 40, 44, 45};     // 35 (BK 32, 3 workgroups per CU) measured slower: not tried
  for (int cfg : cands) {
    // data gradients run next to the weight-gradient stream: only tiles of <= 64 KB LDS (two workgroups per CU), which can share a CU
    // with a 64 KB weight-gradient workgroup; the one-per-CU tiles are a little faster alone and slower in the step (same-box A/B:
    // 1046-1047 vs 1039-1041 images/s; round 3's knob for the full list showed no difference on YOLO and is gone).
    if ((EPI == EPI_PLAIN || EPI == EPI_RES) && (cfg == 3 || cfg == 6 || cfg == 16 || cfg == 17 || cfg == 18 || cfg == 19 || cfg == 26 || cfg == 27 || cfg == 28)) continue;
    if ((cfg == 3 || cfg == 6) && p.CoutPad % 256 != 0) continue;
    if (cfg == 40 && !(igemm8_applicable(p) && (EPI == EPI_STATS || EPI == EPI_PLAIN || EPI == EPI_RES || EPI == EPI_AFF || EPI == EPI_F32))) continue;
    if ((cfg == 44 || cfg == 45) && !(igemm8_applicable(p) && (EPI == EPI_STATS || EPI == EPI_PLAIN || EPI == EPI_RES || EPI == EPI_AFF))) continue;
    if (cfg >= 15 && (!dx_applicable(p) || ((cfg == 17 || cfg == 18 || cfg == 28) && p.CoutPad % 256 != 0))) continue;
    int e = 0;
    const float ms = time_candidate([&] { return run_cfg<EPI>(cfg, p, st); }, e0, e1, st, &e);
    if (e) return e;
    if (ms < best_ms) {
      best_ms = ms;
      best = cfg;
    }
  }
  g_igemm_tuned[igemm_key(p, EPI)] = best;
  tune_mark_timed(TUNE_IGEMM, igemm_key(p, EPI));
  static const bool tune_log = getenv("MI355DET_TUNE_LOG") != nullptr;
  if (tune_log) fprintf(stderr, "[mi355det] igemm tune: M=%d Cout=%d Cin=%d T=%d epi=%d -> cfg %d (%.1f us)\n", p.M, p.CoutPad, p.Cin, p.T, EPI, best, best_ms * 1e3f / 3.f);
  return best;
}

// dx[n, 2*yy+py, 2*xx+px, :] = residual or 0 on one stride-2 parity class (16-byte pieces)
__global__ __launch_bounds__(256) void lattice_fill_kernel(bf16_t* __restrict__ dx, int ld, const bf16_t* __restrict__ res, int ldres, int n, int h, int w,
                                                            int mh, int mw, int py, int px, int c) {
  const int c8 = c / 8;
  const long long total = (long long)n * mh * mw * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c8);
    long long t = i / c8;
    const int xx = (int)(t % mw);
    t /= mw;
    const int yy = (int)(t % mh), im = (int)(t / mh);
    const int oy = 2 * yy + py, ox = 2 * xx + px;
    if (oy >= h || ox >= w) continue;
    const long long pix = ((long long)im * h + oy) * w + ox;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (res) v = *(const uint4*)(res + pix * ldres + ch * 8);
    *(uint4*)(dx + pix * ld + ch * 8) = v;
  }
}

bool g_autotune_mode = false;   // set by mi355det_conv_autotune around a regular entry-point call
TuneMap& g_s2cat_tuned = tune_table(TUNE_S2CAT);   // stride-2 data gradient: 1 = class-concatenated form, 0 = four class launches (tune record)
int g_s2cat_force = -1;          // mi355det_debug_set(5, v): force a form (tests compare the two)

template <int EPI>
int dispatch_igemm(const IgemmParams& p_in, hipStream_t st) {
  IgemmParams p = p_in;
  p.lin_in = p.T == 1 && p.dy[0] == 0 && p.dx[0] == 0 && p.sin == 1 && p.MH == p.Hin && p.MW == p.Win;
  p.lin_out = p.so == 1 && p.sox == 0 && p.oy0 == 0 && p.ox0 == 0 && p.MH == p.Hout && p.MW == p.Wout;
  return g_autotune_mode ? autotune_igemm<EPI>(p, st) : launch_igemm<EPI>(p, st);
}

void set_tap_pad(IgemmParams& p) {
  int mn = 0;
  p.dy_pack = p.dx_pack = 0;
  for (int t = 0; t < p.T; ++t) {
    mn = min(mn, (p.dy[t] * p.Win + p.dx[t]) * p.ldin);
    p.dy_pack |= (unsigned long long)(p.dy[t] + 2) << (4 * t);
    p.dx_pack |= (unsigned long long)(p.dx[t] + 2) << (4 * t);
  }
  p.tap_pad = -mn;
}

int check_shape(const mi355det_conv_shape* s, const char* what) {
  if (!s) return fail(MI355DET_EINVAL, "%s: null shape", what);
  if (!((s->ksize == 1 && s->pad == 0) || (s->ksize == 3 && s->pad == 1)) || (s->stride != 1 && s->stride != 2))
    return fail(MI355DET_EINVAL, "%s: only 1x1/p0 and 3x3/p1 with stride 1 or 2 are supported", what);
  if (s->ho != (s->h + 2 * s->pad - s->ksize) / s->stride + 1 || s->wo != (s->w + 2 * s->pad - s->ksize) / s->stride + 1)
    return fail(MI355DET_EINVAL, "%s: output size does not match the convolution geometry", what);
  if (s->in_ld < s->cin || s->out_ld < s->cout) return fail(MI355DET_EINVAL, "%s: pixel pitch smaller than channel count", what);
  return 0;
}

int grid_m_rows(const mi355det_conv_shape* s, int cout_pad) {
  (void)cout_pad;
  const int bm = 128;   // upper bound on the number of pixel tiles of any configuration
  const long long M = (long long)s->n * s->ho * s->wo;
  return (int)((M + bm - 1) / bm);
}

}  // namespace

extern "C" {

int mi355det_debug_set(int key, int value) {
  if (key == 0) g_tune = value;
  if (key == 1) g_wgrad_general = value;
  if (key == 2) g_dgrad_s2_off = value;
  if (key == 3) igemm8_set_dbg_mode(value);      // 1 = phase stamps, 2 = k-step starts only (tools/prof_ig8.py)
  if (key == 8) g_wgrad8_off = value;            // 1 = the weight-gradient tuner leaves the 256 x 256 phase-staggered kernel out (A/B)
  if (key == 7) g_wgrad_force_dbg = value;       // weight gradient: split count (+ 65536 = the 256 x 256 phase-staggered kernel) for every launch, 0 = tuned
  if (key == 6) g_wgrad_ablate = value;          // weight-gradient ablation builds (timing only): wgrad_kernels.hip
  if (key == 5) g_s2cat_force = value;           // stride-2 data gradient: 1 = class-concatenated form, 0 = four class launches, -1 = tuned choice
  return 0;
}

// plan-build helper (synchronises; never part of the step): the next conv_fwd / conv_dgrad calls made while the mode
// is on time their candidate tile configurations and remember the fastest per shape.
int mi355det_conv_autotune_mode(int on) {
  g_autotune_mode = on != 0;
  return 0;
}

int mi355det_debug_ptr(int key, void* ptr) {
  if (key == 0) g_dbg = (unsigned long long*)ptr;
  if (key == 1) igemm8_set_dbg((unsigned long long*)ptr);
  return 0;
}

int mi355det_conv_stats_rows(const mi355det_conv_shape* s, int32_t cout_pad) {
  if (!s) return 0;
  return grid_m_rows(s, cout_pad);
}

static int conv_fwd_impl(const mi355det_conv_shape* s, const void* x, const void* w, const float* bias, void* y, int out_f32, float* stats,
                         int32_t cout_pad, const mi355det_conv_epilogue* ex, void* stream) {
  if (int e = check_shape(s, "conv_fwd")) return e;
  if (int e = ensure_zero_page()) return e;
  if (cout_pad < s->cout) return fail(MI355DET_EINVAL, "%s: cout_pad < cout", "conv_fwd");
  IgemmParams p{};
  p.x = (const bf16_t*)x;
  p.w = (const bf16_t*)w;
  p.y = y;
  p.bias = bias;
  p.stats = stats;
  p.zero = g_zero_page;
  p.MH = s->ho;
  p.MW = s->wo;
  p.M = s->n * s->ho * s->wo;
  p.Hin = s->h; p.Win = s->w; p.ldin = s->in_ld; p.Cin = s->cin; p.sin = s->stride;
  p.Hout = s->ho; p.Wout = s->wo; p.ldout = s->out_ld; p.so = 1; p.oy0 = 0; p.ox0 = 0;
  p.Cout = s->cout; p.CoutPad = cout_pad;
  p.T = s->ksize * s->ksize;
  for (int kh = 0; kh < s->ksize; ++kh)
    for (int kw = 0; kw < s->ksize; ++kw) {
      p.dy[kh * s->ksize + kw] = kh - s->pad;
      p.dx[kh * s->ksize + kw] = kw - s->pad;
    }
  p.dMW = make_fastdiv((unsigned)p.MW);
  p.dMH = make_fastdiv((unsigned)p.MH);
  p.ynstride = (long long)s->ho * s->wo * s->out_ld;
  set_tap_pad(p);
  if (ex) {
    p.scale = ex->scale;
    p.bias = ex->shift;
    p.relu = ex->relu;
    p.slope = ex->slope;
    p.res = (const bf16_t*)ex->residual;
    p.ldres = ex->residual_ld;
    if (ex->out_image_stride) {
      if (!out_f32) return fail(MI355DET_EINVAL, "%s: out_image_stride needs an fp32 output", "conv_fwd_ex");
      p.ynstride = ex->out_image_stride;
    }
    if (out_f32 && ex->residual) return fail(MI355DET_EINVAL, "%s: residual needs a bf16 output", "conv_fwd_ex");
    if (!out_f32 && (s->cout % 8)) return fail(MI355DET_EINVAL, "%s: bf16 outputs need cout %% 8 == 0", "conv_fwd_ex");
    const int r = out_f32 ? dispatch_igemm<EPI_F32>(p, S(stream)) : dispatch_igemm<EPI_AFF>(p, S(stream));
    return r < 0 ? r : 0;
  }
  const int r = out_f32 ? dispatch_igemm<EPI_F32>(p, S(stream)) : stats ? dispatch_igemm<EPI_STATS>(p, S(stream)) : dispatch_igemm<EPI_PLAIN>(p, S(stream));
  return r < 0 ? r : 0;
}

int mi355det_conv_fwd(const mi355det_conv_shape* s, const void* x, const void* w, const float* bias, void* y, int out_f32, float* stats,
                      int32_t cout_pad, void* stream) {
  return conv_fwd_impl(s, x, w, bias, y, out_f32, stats, cout_pad, nullptr, stream);
}

int mi355det_conv_fwd_ex(const mi355det_conv_shape* s, const void* x, const void* w, const mi355det_conv_epilogue* e, void* y, int out_f32,
                         int32_t cout_pad, void* stream) {
  if (!e) return fail(MI355DET_EINVAL, "%s: null epilogue", "conv_fwd_ex");
  return conv_fwd_impl(s, x, w, nullptr, y, out_f32, nullptr, cout_pad, e, stream);
}

// dgrad tap lists.  stride 1: dx[y,x] = sum_{kh,kw} dy[y+p-kh, x+p-kw] * w[kh,kw]  -> tap j=(kh,kw): d = p-kh.
// stride 2 (k=3,p=1): output pixel parity (py,px); valid kh have (y+1-kh) even: py=0 -> kh=1 (dy 0);
// py=1 -> kh=0 (dy +1), kh=2 (dy 0)  [in units of the dy lattice: (y+1-kh)/2 with y=2*yy+py].
static int dgrad_taps(const mi355det_conv_shape* s, int py, int px, int* fwd_tap, int* dy, int* dx) {
  int n = 0;
  const int k = s->ksize, pd = s->pad;
  for (int kh = 0; kh < k; ++kh)
    for (int kw = 0; kw < k; ++kw) {
      if (s->stride == 1) {
        fwd_tap[n] = kh * k + kw;
        dy[n] = pd - kh;
        dx[n] = pd - kw;
        ++n;
      } else {
        const int ny = py + pd - kh, nx = px + pd - kw;
        if ((ny & 1) || (nx & 1)) continue;
        fwd_tap[n] = kh * k + kw;
        dy[n] = ny / 2;   // exact (even), may be negative zero-side only for ny in {-0}: ny in {-1..2}
        dx[n] = nx / 2;
        ++n;
      }
    }
  return n;
}

size_t mi355det_dgrad_pack_elems(const mi355det_conv_shape* s) {
  if (!s) return 0;
  mi355det_conv_shape d = *s;
  d.in_ld = d.cin;                                       // sizes only: the class-concatenated pack exists for the shape, whatever pitch a call uses
  return dgrad_class_elems(s) + s2cat_elems(&d) + 64;    // [four class packs | stride-2 concatenated packs | slack]
}

int mi355det_pack_weights(const mi355det_conv_shape* s, const float* w, int w_is_ohwi, void* w_fwd, int32_t cout_pad, void* w_dgrad,
                          void* stream) {
  if (int e = check_shape(s, "pack_weights")) return e;
  const int kk = s->ksize * s->ksize;
  // master layout: torch OIHW [cout][cin][k][k] or engine OHWI [cout][k][k][cin]
  const long long s_co = (long long)s->cin * kk, s_ci = w_is_ohwi ? 1 : kk, s_t = w_is_ohwi ? s->cin : 1;
  if (w_fwd) {
    const long long total = (long long)cout_pad * kk * s->cin;
    hipLaunchKernelGGL(pack_fwd_kernel, dim3((int)min((long long)2048, (total + 255) / 256)), dim3(256), 0, S(stream), w, (bf16_t*)w_fwd, s->cout,
                       cout_pad, s->cin, s->cin, kk, s_co, s_ci, s_t);
  }
  if (w_dgrad) {
    const int cin_pad = (s->cin + 31) / 32 * 32;
    bf16_t* out = (bf16_t*)w_dgrad;
    const int classes = s->stride == 1 ? 1 : 4;
    for (int c = 0; c < classes; ++c) {
      int ft[9] = {0}, dy[9], dx[9];
      const int nt = dgrad_taps(s, c >> 1, c & 1, ft, dy, dx);
      const long long total = (long long)cin_pad * nt * s->cout;
      if (total == 0) continue;     // 1x1 stride-2: the odd parity classes have no taps
      hipLaunchKernelGGL(pack_dgrad_kernel, dim3((int)min((long long)2048, (total + 255) / 256)), dim3(256), 0, S(stream), w, out, s->cout, s->cin,
                         cin_pad, kk, nt, s_co, s_ci, s_t, ft[0], ft[1], ft[2], ft[3], ft[4], ft[5], ft[6], ft[7], ft[8]);
      out += total;
    }
    mi355det_conv_shape d = *s;
    d.in_ld = d.cin;
    if (s2cat_eligible(&d)) {
      const long long total = (long long)12 * s->cin * s->cout;
      hipLaunchKernelGGL(pack_s2cat_kernel, dim3((int)min((long long)2048, (total + 255) / 256)), dim3(256), 0, S(stream), w,
                         (bf16_t*)w_dgrad + dgrad_class_elems(s), s->cout, s->cin, s_co, s_ci, s_t);
    }
  }
  return check_launch("pack_weights");
}

// ---- batched packing: every layer of a model in ONE launch (a step re-packs ~75 layers; 130 tiny launches cost
// ~1 ms of pure launch latency).  The host builds two tables once: entries (one per forward pack / dgrad class) and a
// block table (entry, first element) so that each workgroup converts one 4096-element chunk of one entry.
struct PackEntry {
  const float* src;
  bf16_t* dst;
  long long s_co, s_ci, s_t;
  int total, rows_valid, cin, kk, ntaps, mode, cdim, cob;   // mode 0: fwd [rows_pad][kk][cin]; 1: dgrad [cin_pad][ntaps][cout]
  int taps[9];
  int ncib, ncob, pad1;   // dgrad tiles: ncib blocks of 32 input channels x ncob blocks of cob output channels per tap
};
#define PACK_CHUNK 4096

__global__ __launch_bounds__(256) void pack_batched_kernel(const PackEntry* __restrict__ entries, const int2* __restrict__ blocks) {
  const int2 bt = blocks[blockIdx.x];
  const PackEntry e = entries[bt.x];
  if (e.mode == 2) {
    // class-concatenated stride-2 data-gradient packs (pack_s2cat_kernel's layout): elementwise, two small layers per step
    const long long n0 = (long long)2 * e.cin * 2 * e.cdim;
    const int end = min(e.total, bt.y + PACK_CHUNK);
    for (int i = bt.y + threadIdx.x; i < end; i += 256) {
      const int py = i >= n0;
      const long long j = py ? i - n0 : i;
      const int T = py ? 4 : 2;
      const int co = (int)(j % e.cdim), tap = (int)((j / e.cdim) % T), r = (int)(j / ((long long)e.cdim * T));
      const int px = r / e.cin, ci = r - px * e.cin;
      const int oy = py ? tap >> 1 : 0, ox = py ? tap & 1 : tap;
      const int kh = s2cat_k(py, oy), kw = s2cat_k(px, ox);
      float v = 0.f;
      if (kh >= 0 && kw >= 0) v = e.src[co * e.s_co + ci * e.s_ci + (kh * 3 + kw) * e.s_t];
      e.dst[i] = f2s(v);
    }
    return;
  }
  if (e.mode == 0) {
    // forward pack: the OHWI master is already K-contiguous -> linear, coalesced both ways
    const int end = min(e.total, bt.y + PACK_CHUNK);
    for (int i = bt.y + threadIdx.x; i < end; i += 256) {
      const int c = i % e.cin, t = (i / e.cin) % e.kk, co = i / (e.cin * e.kk);
      float v = 0.f;
      if (co < e.rows_valid) v = e.src[co * e.s_co + c * e.s_ci + t * e.s_t];
      e.dst[i] = f2s(v);
    }
    return;
  }
  // dgrad pack = per-tap transpose [cout][cin] -> [cin_pad][tap'][cout]: 32(ci) x cob(co) tile through LDS so that both
  // the fp32 reads (128 B runs along ci) and the bf16 writes (runs along co) are coalesced (the direct form fetched 27x
  // the bytes it needed: 6.7 GB per step, measured with FETCH_SIZE)
  __shared__ float tl[64][33];
  int t = bt.y;
  const int co_blk = t % e.ncob;
  t /= e.ncob;
  const int ci_blk = t % e.ncib, j = t / e.ncib;
  int tap = e.taps[0];
#pragma unroll
  for (int q = 1; q < 9; ++q)
    if (j == q) tap = e.taps[q];
  const int cl = threadIdx.x & 31;
  for (int r = threadIdx.x >> 5; r < e.cob; r += 8) {
    const int co = co_blk * e.cob + r, ci = ci_blk * 32 + cl;
    tl[r][cl] = ci < e.rows_valid ? e.src[co * e.s_co + ci * e.s_ci + tap * e.s_t] : 0.f;
  }
  __syncthreads();
  const int col = threadIdx.x % e.cob;
  for (int r = threadIdx.x / e.cob; r < 32; r += 256 / e.cob) {
    const long long ci = ci_blk * 32 + r;
    e.dst[(ci * e.ntaps + j) * e.cdim + co_blk * e.cob + col] = f2s(tl[col][r]);
  }
}

size_t mi355det_pack_table_bytes(const mi355det_pack_item* items, int32_t n, int32_t* n_entries, int32_t* n_blocks) {
  int ne = 0;
  long long nb = 0;
  for (int i = 0; i < n; ++i) {
    const mi355det_conv_shape* s = &items[i].shape;
    const int kk = s->ksize * s->ksize;
    if (items[i].w_fwd) {
      ++ne;
      nb += ((long long)items[i].cout_pad * kk * s->cin + PACK_CHUNK - 1) / PACK_CHUNK;
    }
    if (items[i].w_dgrad) {
      const int cin_pad = (s->cin + 31) / 32 * 32;
      const int classes = s->stride == 1 ? 1 : 4;
      for (int c = 0; c < classes; ++c) {
        int ft[9], dy[9], dx[9];
        const int nt = dgrad_taps(s, c >> 1, c & 1, ft, dy, dx);
        ++ne;
        const int cob = s->cout % 64 == 0 ? 64 : 32;
        nb += (long long)nt * (cin_pad / 32) * (s->cout / cob);
      }
      mi355det_conv_shape d = *s;
      d.in_ld = d.cin;
      if (s2cat_eligible(&d)) {
        ++ne;
        nb += ((long long)s2cat_elems(&d) + PACK_CHUNK - 1) / PACK_CHUNK;
      }
    }
  }
  if (n_entries) *n_entries = ne;
  if (n_blocks) *n_blocks = (int)nb;
  return (size_t)ne * sizeof(PackEntry) + (size_t)nb * sizeof(int2);
}

int mi355det_pack_table_build(const mi355det_pack_item* items, int32_t n, void* host_table, size_t host_bytes) {
  int ne = 0, nb = 0;
  const size_t need = mi355det_pack_table_bytes(items, n, &ne, &nb);
  if (host_bytes < need) return fail(MI355DET_EWORKSPACE, "%s: host table too small", "pack_table_build");
  PackEntry* E = (PackEntry*)host_table;
  int2* B = (int2*)((char*)host_table + (size_t)ne * sizeof(PackEntry));
  int ei = 0, bi = 0;
  auto add_blocks = [&](int entry, long long total) {
    for (long long o = 0; o < total; o += PACK_CHUNK) B[bi++] = make_int2(entry, (int)o);
  };
  for (int i = 0; i < n; ++i) {
    const mi355det_conv_shape* s = &items[i].shape;
    if (int e = check_shape(s, "pack_table_build")) return e;
    if (items[i].w_dgrad && s->cout % 32 != 0) return fail(MI355DET_EINVAL, "%s: dgrad pack needs cout %% 32 == 0", "pack_table_build");
    const int kk = s->ksize * s->ksize;
    const long long s_co = (long long)s->cin * kk, s_ci = items[i].w_is_ohwi ? 1 : kk, s_t = items[i].w_is_ohwi ? s->cin : 1;
    if (items[i].w_fwd) {
      PackEntry& e = E[ei];
      e = PackEntry{};
      e.src = items[i].w; e.dst = (bf16_t*)items[i].w_fwd; e.s_co = s_co; e.s_ci = s_ci; e.s_t = s_t;
      e.total = items[i].cout_pad * kk * s->cin; e.rows_valid = s->cout; e.cin = s->cin; e.kk = kk; e.mode = 0;
      add_blocks(ei++, e.total);
    }
    if (items[i].w_dgrad) {
      const int cin_pad = (s->cin + 31) / 32 * 32;
      bf16_t* out = (bf16_t*)items[i].w_dgrad;
      const int classes = s->stride == 1 ? 1 : 4;
      for (int c = 0; c < classes; ++c) {
        int ft[9] = {0}, dy[9], dx[9];
        const int nt = dgrad_taps(s, c >> 1, c & 1, ft, dy, dx);
        PackEntry& e = E[ei];
        e = PackEntry{};
        e.src = items[i].w; e.dst = out; e.s_co = s_co; e.s_ci = s_ci; e.s_t = s_t;
        e.total = cin_pad * nt * s->cout; e.rows_valid = s->cin; e.cin = s->cin; e.kk = kk; e.ntaps = nt; e.mode = 1; e.cdim = s->cout;
        for (int q = 0; q < 9; ++q) e.taps[q] = ft[q];
        e.cob = s->cout % 64 == 0 ? 64 : 32;
        e.ncib = cin_pad / 32;
        e.ncob = s->cout / e.cob;
        for (int tix = 0; tix < nt * e.ncib * e.ncob; ++tix) B[bi++] = make_int2(ei, tix);
        ++ei;
        out += e.total;
      }
      mi355det_conv_shape d = *s;
      d.in_ld = d.cin;
      if (s2cat_eligible(&d)) {
        PackEntry& e = E[ei];
        e = PackEntry{};
        e.src = items[i].w; e.dst = (bf16_t*)items[i].w_dgrad + dgrad_class_elems(s); e.s_co = s_co; e.s_ci = s_ci; e.s_t = s_t;
        e.total = (int)s2cat_elems(&d); e.cin = s->cin; e.cdim = s->cout; e.mode = 2;
        add_blocks(ei++, e.total);
      }
    }
  }
  return 0;
}

int mi355det_pack_weights_batched(const void* dev_table, int32_t n_entries, int32_t n_blocks, void* stream) {
  if (!dev_table || n_entries <= 0 || n_blocks <= 0) return fail(MI355DET_EINVAL, "%s: bad table", "pack_weights_batched");
  const PackEntry* E = (const PackEntry*)dev_table;
  const int2* B = (const int2*)((const char*)dev_table + (size_t)n_entries * sizeof(PackEntry));
  hipLaunchKernelGGL(pack_batched_kernel, dim3(n_blocks), dim3(256), 0, S(stream), E, B);
  return check_launch("pack_weights_batched");
}

int mi355det_unpack_wgrad(const mi355det_conv_shape* s, const float* dw, float* grad, void* stream) {
  if (int e = check_shape(s, "unpack_wgrad")) return e;
  const int kk = s->ksize * s->ksize;
  const long long total = (long long)s->cout * s->cin * kk;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((int)min((long long)2048, (total + 255) / 256)), dim3(256), 0, S(stream), dw, grad, s->cout, s->cin,
                     s->cin, kk);
  return check_launch("unpack_wgrad");
}

static int conv_dgrad_impl(const mi355det_conv_shape* s, const void* dy, const void* wt, void* dx, const void* residual, int32_t residual_ld,
                           const void* z, int32_t z_ld, const float* scale_shift, float slope, float* partials, void* stream) {
  if (int e = check_shape(s, "conv_dgrad")) return e;
  if (int e = ensure_zero_page()) return e;
  if (s->cout % 32 != 0) return fail(MI355DET_EINVAL, "%s: Cout (the dgrad reduction dim) must be a multiple of 32 (got %lld)", "conv_dgrad", s->cout);
  if (!partials && !g_autotune_mode) {
    // few-channel 3x3 stride-2 layers: all four parity classes in one launch over shared dy tiles (dgrad_s2_kernels.hip)
    const int r = mi355det_internal_dgrad_s2(s, dy, wt, dx, residual, residual_ld, stream);
    if (r != 0) return r < 0 ? r : 0;
  }
  const int cin_pad = (s->cin + 31) / 32 * 32;
  const bf16_t* wp = (const bf16_t*)wt;
  const int classes = s->stride == 1 ? 1 : 4;
  if (!partials && s2cat_eligible(s) && (!residual || residual_ld == s->cin)) {
    // class-concatenated form (two wide GEMMs) or the four class launches: per shape, whichever the plan-build timing found faster
    const unsigned long long key = ((((unsigned long long)s->n * 4099 + s->h) * 4099 + s->w) * 4099 + s->cin) * 4099 + s->cout + (residual ? 1ull << 62 : 0) + (MI355_F16 ? 1ull << 61 : 0);
    int choice = -1;
    auto it = g_s2cat_tuned.find(key);
    if (it != g_s2cat_tuned.end()) choice = it->second;
    if (g_s2cat_force >= 0) choice = g_s2cat_force;
    auto run_cat = [&]() -> int {
      const bf16_t* wc = (const bf16_t*)wt + dgrad_class_elems(s);
      for (int py = 0; py < 2; ++py) {
        IgemmParams p{};
        p.T = py ? 4 : 2;
        for (int t = 0; t < p.T; ++t) {
          p.dy[t] = py ? t >> 1 : 0;
          p.dx[t] = py ? t & 1 : t;
        }
        p.x = (const bf16_t*)dy;
        p.w = wc;
        p.y = dx;
        p.res = (const bf16_t*)residual;
        p.ldres = 2 * s->cin;
        p.zero = g_zero_page;
        p.MH = s->ho; p.MW = s->wo;
        p.M = s->n * s->ho * s->wo;
        p.Hin = s->ho; p.Win = s->wo; p.ldin = s->out_ld; p.Cin = s->cout; p.sin = 1;
        p.Hout = s->h; p.Wout = s->wo; p.ldout = 2 * s->cin;       // view [n, h, wo, 2*cin] of dx: a row of the view = the two pixels 2xx, 2xx+1
        p.so = 2; p.sox = 1; p.oy0 = py; p.ox0 = 0;
        p.Cout = 2 * s->cin; p.CoutPad = 2 * s->cin;
        p.dMW = make_fastdiv((unsigned)p.MW);
        p.dMH = make_fastdiv((unsigned)p.MH);
        set_tap_pad(p);
        const int e = residual ? dispatch_igemm<EPI_RES>(p, S(stream)) : dispatch_igemm<EPI_PLAIN>(p, S(stream));
        if (e < 0) return e;
        wc += (size_t)2 * s->cin * p.T * s->cout;
      }
      return 0;
    };
    if (g_autotune_mode && g_s2cat_force < 0 && !tune_locked_has(TUNE_S2CAT, key)) {
      // time both forms (each tunes its own tiles first); the output stays valid either way
      // (outputs are scratch while tuning: with an in-place residual (residual == dx) the timing runs accumulate into dx; both engines discard
      //  the outputs of their tuning pass)
      EventPair ev;
      if (!ev.ok) return fail(MI355DET_ELAUNCH, "%s: event create failed", "conv_dgrad");
      hipEvent_t e0 = ev.e0, e1 = ev.e1;
      float ms[2] = {0.f, 0.f};
      for (int form = 0; form < 2; ++form) {
        g_s2cat_force = form;
        int e = conv_dgrad_impl(s, dy, wt, dx, residual, residual_ld, nullptr, 0, nullptr, 0.f, nullptr, stream);      // tunes the tiles of this form
        g_autotune_mode = false;
        if (!e) ms[form] = time_candidate([&] { return conv_dgrad_impl(s, dy, wt, dx, residual, residual_ld, nullptr, 0, nullptr, 0.f, nullptr, stream); }, e0, e1, S(stream), &e);
        g_autotune_mode = true;
        g_s2cat_force = -1;
        if (e) return e;
      }
      choice = ms[1] < ms[0] ? 1 : 0;
      g_s2cat_tuned[key] = choice;
      tune_mark_timed(TUNE_S2CAT, key);
      static const bool tune_log = getenv("MI355DET_TUNE_LOG") != nullptr;
      if (tune_log) fprintf(stderr, "[mi355det] stride-2 dgrad %d->%d @%dx%d: four classes %.1f us, concatenated %.1f us\n", s->cin, s->cout, s->h, s->w,
                            ms[0] * 1e3f / 3.f, ms[1] * 1e3f / 3.f);
      if (choice == 1) return run_cat();       // leave the output of the chosen form (identical up to summation order)
    } else if (choice == 1) {
      return run_cat();
    }
  }
  for (int c = 0; c < classes; ++c) {
    IgemmParams p{};
    int ft[9];
    p.T = dgrad_taps(s, c >> 1, c & 1, ft, p.dy, p.dx);
    if (p.T == 0 && partials) return fail(MI355DET_EINVAL, "%s: 1x1 stride-2 layers are not supported by the fused BN reduction", "conv_dgrad_bn");
    if (p.T == 0) {
      // 1x1 stride-2 convolutions never read the odd input rows / columns: their data gradient is zero (+ residual)
      const int mh = (s->h + 1) / 2, mw = (s->w + 1) / 2;
      const long long total = (long long)s->n * mh * mw * (s->cin / 8);
      if (s->cin % 8) return fail(MI355DET_EINVAL, "%s: cin must be a multiple of 8", "conv_dgrad");
      hipLaunchKernelGGL(lattice_fill_kernel, dim3((int)min((long long)4096, (total + 255) / 256)), dim3(256), 0, S(stream), (bf16_t*)dx, s->in_ld,
                         (const bf16_t*)residual, residual_ld, s->n, s->h, s->w, mh, mw, c >> 1, c & 1, s->cin);
      continue;
    }
    p.x = (const bf16_t*)dy;
    p.w = wp;
    p.y = dx;
    p.res = (const bf16_t*)residual;
    p.ldres = residual_ld;
    p.zero = g_zero_page;
    if (s->stride == 1) {
      p.MH = s->h; p.MW = s->w; p.so = 1; p.oy0 = 0; p.ox0 = 0;
    } else {
      p.MH = (s->h + 1) / 2; p.MW = (s->w + 1) / 2; p.so = 2; p.oy0 = c >> 1; p.ox0 = c & 1;
    }
    p.M = s->n * p.MH * p.MW;
    p.Hin = s->ho; p.Win = s->wo; p.ldin = s->out_ld; p.Cin = s->cout; p.sin = 1;
    p.Hout = s->h; p.Wout = s->w; p.ldout = s->in_ld;
    p.Cout = s->cin; p.CoutPad = cin_pad;
    p.dMW = make_fastdiv((unsigned)p.MW);
    p.dMH = make_fastdiv((unsigned)p.MH);
    set_tap_pad(p);
    int e;
    if (partials) {
      p.z = (const bf16_t*)z;
      p.ldz = z_ld;
      p.ss = scale_shift;
      p.slope = slope;
      p.stats = partials + (long long)c * ((p.M + 127) / 128) * 2 * cin_pad;    // each parity class fills its own rows
      e = dispatch_igemm<EPI_BNRED>(p, S(stream));
    } else {
      e = residual ? dispatch_igemm<EPI_RES>(p, S(stream)) : dispatch_igemm<EPI_PLAIN>(p, S(stream));
    }
    if (e < 0) return e;
    wp += (long long)cin_pad * p.T * s->cout;
  }
  return 0;
}

int mi355det_conv_dgrad(const mi355det_conv_shape* s, const void* dy, const void* wt, void* dx, const void* residual, int32_t residual_ld,
                        void* stream) {
  return conv_dgrad_impl(s, dy, wt, dx, residual, residual_ld, nullptr, 0, nullptr, 0.f, nullptr, stream);
}

// Data gradient + the FrozenBN / ReLU backward of the layer that PRODUCED the convolution's input, in the epilogue:
//   dx = bf16(conv2d_input(dy)) * scale[c] * (act > 0)        (relu != 0; scale may be NULL)
// i.e. what mi355det_conv_dgrad followed by mi355det_relu_affine_bwd(g = dx, a = act, scale) stores, without the pass over g and act
// (6 B per element) and its launch.  Stride-1 shapes only (the stride-2 forms write parity classes / class-concatenated views).
int mi355det_conv_dgrad_mask(const mi355det_conv_shape* s, const void* dy, const void* wt, void* dx, const void* act, int32_t act_ld,
                             const float* scale, int32_t relu, void* stream) {
  if (int e = check_shape(s, "conv_dgrad_mask")) return e;
  if (int e = ensure_zero_page()) return e;
  if (!dy || !wt || !dx || !act) return fail(MI355DET_EINVAL, "%s: null argument", "conv_dgrad_mask");
  if (s->stride != 1) return fail(MI355DET_EINVAL, "%s: stride-1 convolutions only", "conv_dgrad_mask");
  if (s->cout % 32 != 0) return fail(MI355DET_EINVAL, "%s: Cout (the dgrad reduction dim) must be a multiple of 32 (got %lld)", "conv_dgrad_mask", s->cout);
  if (s->cin % 8 != 0 || act_ld % 8 != 0 || act_ld < s->cin) return fail(MI355DET_EINVAL, "%s: cin and the activation pitch must be multiples of 8", "conv_dgrad_mask");
  const int cin_pad = (s->cin + 31) / 32 * 32;
  IgemmParams p{};
  int ft[9];
  p.T = dgrad_taps(s, 0, 0, ft, p.dy, p.dx);
  p.x = (const bf16_t*)dy;
  p.w = (const bf16_t*)wt;
  p.y = dx;
  p.res = (const bf16_t*)act;
  p.ldres = act_ld;
  p.scale = scale;
  p.relu = relu ? 3 : 4;
  p.zero = g_zero_page;
  p.MH = s->h; p.MW = s->w; p.so = 1;
  p.M = s->n * p.MH * p.MW;
  p.Hin = s->ho; p.Win = s->wo; p.ldin = s->out_ld; p.Cin = s->cout; p.sin = 1;
  p.Hout = s->h; p.Wout = s->w; p.ldout = s->in_ld;
  p.Cout = s->cin; p.CoutPad = cin_pad;
  p.dMW = make_fastdiv((unsigned)p.MW);
  p.dMH = make_fastdiv((unsigned)p.MH);
  set_tap_pad(p);
  const int e = dispatch_igemm<EPI_RES>(p, S(stream));
  return e < 0 ? e : 0;
}

// ---- split-K data gradient: few output pixels, very deep reduction (the 1204-class RetinaNet head on the small pyramid levels:
//      8 x 7 x 7 pixels against K = 9 x 10 880).  The plain form leaves 8-80 workgroups walking 1530 k-steps each; here the channel axis
//      is cut into `ksplit` ranges, every (tile, range) is a workgroup writing an fp32 partial tile, and splitk_reduce_kernel adds the
//      ranges in a fixed order (deterministic), adds the residual and rounds to bf16 once.
static int dgrad_ksplit(const mi355det_conv_shape* s) {
  if (!s || s->stride != 1 || s->ksize * s->ksize > MAX_TAPS) return 0;
  const int cin_pad = (s->cin + 31) / 32 * 32;
  if (cin_pad % 128 != 0 || s->cout % 64 != 0 || s->in_ld % 8 != 0) return 0;
  const long long M = (long long)s->n * s->h * s->w, K = (long long)s->ksize * s->ksize * s->cout;
  const long long tiles = (M + 127) / 128 * (cin_pad / 128);
  if (tiles > 96 || K < 8192) return 0;
  const int chunks = s->cout / 64;
  int best = 0;
  for (int d = 2; d <= chunks; ++d)
    if (chunks % d == 0 && tiles * d <= 512 && chunks / d >= 4) best = d;
  return best;
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, long long sstride, bf16_t* __restrict__ dx, int ld,
                                                            const bf16_t* __restrict__ res, int ldres, long long M, int C, int CP) {
  const int c8 = C / 8;
  const long long total = M * c8;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / c8;
    const int c = (int)(i - m * c8) * 8;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* src = part + m * CP + c;
    for (int sp = 0; sp < S; ++sp) {
      const float4 v0 = *(const float4*)(src + sp * sstride), v1 = *(const float4*)(src + sp * sstride + 4);
      a[0] += v0.x; a[1] += v0.y; a[2] += v0.z; a[3] += v0.w;
      a[4] += v1.x; a[5] += v1.y; a[6] += v1.z; a[7] += v1.w;
    }
    if (res) {
      const uint4 r = *(const uint4*)(res + m * ldres + c);
      const unsigned rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a[2 * k] += s2f((unsigned short)(rr[k] & 0xFFFFu));
        a[2 * k + 1] += s2f((unsigned short)(rr[k] >> 16));
      }
    }
    uint4 o;
    o.x = f2s(a[0]) | ((unsigned)f2s(a[1]) << 16);
    o.y = f2s(a[2]) | ((unsigned)f2s(a[3]) << 16);
    o.z = f2s(a[4]) | ((unsigned)f2s(a[5]) << 16);
    o.w = f2s(a[6]) | ((unsigned)f2s(a[7]) << 16);
    *(uint4*)(dx + m * ld + c) = o;
  }
}

size_t mi355det_conv_dgrad_workspace(const mi355det_conv_shape* s) {
  const int ks = dgrad_ksplit(s);
  if (ks < 2) return 0;
  const int cin_pad = (s->cin + 31) / 32 * 32;
  return (size_t)ks * (size_t)s->n * s->h * s->w * cin_pad * sizeof(float);
}

int mi355det_conv_dgrad_ws(const mi355det_conv_shape* s, const void* dy, const void* wt, void* dx, const void* residual, int32_t residual_ld,
                           void* workspace, size_t workspace_bytes, void* stream) {
  const int ks = dgrad_ksplit(s);
  if (ks < 2 || !workspace || g_autotune_mode) return conv_dgrad_impl(s, dy, wt, dx, residual, residual_ld, nullptr, 0, nullptr, 0.f, nullptr, stream);
  if (int e = check_shape(s, "conv_dgrad_ws")) return e;
  if (int e = ensure_zero_page()) return e;
  if (!dy || !wt || !dx) return fail(MI355DET_EINVAL, "%s: null argument", "conv_dgrad_ws");
  if (workspace_bytes < mi355det_conv_dgrad_workspace(s)) return fail(MI355DET_EINVAL, "%s: workspace too small (%zu < %zu bytes)", "conv_dgrad_ws", workspace_bytes, mi355det_conv_dgrad_workspace(s));
  if (s->cin % 8 != 0 || (residual && residual_ld % 8 != 0)) return fail(MI355DET_EINVAL, "%s: cin and the residual pitch must be multiples of 8", "conv_dgrad_ws");
  const int cin_pad = (s->cin + 31) / 32 * 32;
  IgemmParams p{};
  int ft[9];
  p.T = dgrad_taps(s, 0, 0, ft, p.dy, p.dx);
  p.x = (const bf16_t*)dy;
  p.w = (const bf16_t*)wt;
  p.y = workspace;
  p.zero = g_zero_page;
  p.MH = s->h; p.MW = s->w; p.so = 1;
  p.M = s->n * p.MH * p.MW;
  p.Hin = s->ho; p.Win = s->wo; p.ldin = s->out_ld; p.Cin = s->cout; p.sin = 1;
  p.Hout = s->h; p.Wout = s->w; p.ldout = cin_pad;
  p.ynstride = (long long)s->h * s->w * cin_pad;
  p.Cout = cin_pad; p.CoutPad = cin_pad;                          // the padded channels of the partial tiles are exact zeros (zero weight rows)
  p.ksplit = ks;
  p.ysplit = (long long)p.M * cin_pad;
  p.dMW = make_fastdiv((unsigned)p.MW);
  p.dMH = make_fastdiv((unsigned)p.MH);
  set_tap_pad(p);
  p.lin_in = p.T == 1 && p.dy[0] == 0 && p.dx[0] == 0 && p.sin == 1 && p.MH == p.Hin && p.MW == p.Win;
  p.lin_out = 1;
  if (int e = launch_cfg<2, 2, 4, 4, 64, 2, EPI_F32>(p, S(stream))) return e;
  const long long total = (long long)p.M * (s->cin / 8);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)min((long long)2048, (total + 255) / 256)), dim3(256), 0, S(stream), (const float*)workspace, ks, p.ysplit,
                     (bf16_t*)dx, s->in_ld, (const bf16_t*)residual, residual_ld, (long long)p.M, s->cin, cin_pad);
  return check_launch("conv_dgrad_ws");
}

int mi355det_conv_dgrad_bn_rows(const mi355det_conv_shape* s) {
  if (!s) return 0;
  if (s->stride == 1) return (int)(((long long)s->n * s->h * s->w + 127) / 128);
  const long long mc = (long long)s->n * ((s->h + 1) / 2) * ((s->w + 1) / 2);
  return (int)(4 * ((mc + 127) / 128));
}

int mi355det_conv_dgrad_bn(const mi355det_conv_shape* s, const void* dy, const void* wt, void* dx, const void* residual, int32_t residual_ld,
                           const void* z, int32_t z_ld, const float* scale_shift, float slope, float* partials, void* stream) {
  if (!z || !scale_shift || !partials) return fail(MI355DET_EINVAL, "%s: null argument", "conv_dgrad_bn");
  if (s && s->cin % 8 != 0) return fail(MI355DET_EINVAL, "%s: cin must be a multiple of 8", "conv_dgrad_bn");
  return conv_dgrad_impl(s, dy, wt, dx, residual, residual_ld, z, z_ld, scale_shift, slope, partials, stream);
}

int mi355det_stem_im2col(const float* img, void* out, int32_t n, int32_t h, int32_t w, void* stream) {
  if (n <= 0 || h <= 0 || w <= 0) return fail(MI355DET_EINVAL, "%s: bad shape", "stem_im2col");
  const int segs = (w + 127) / 128;
  const long long blocks = (long long)n * h * segs;
  if (blocks > 0x7FFFFFFFll) return fail(MI355DET_EINVAL, "%s: image batch too large", "stem_im2col");
  hipLaunchKernelGGL(stem_im2col_kernel, dim3((int)blocks), dim3(256), 0, S(stream), img, (bf16_t*)out, n, h, w, segs);
  return check_launch("stem_im2col");
}

}  // extern "C"
