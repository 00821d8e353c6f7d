// Implicit-GEMM convolution, phase-staggered form (gfx950): 256 pixels x 256 channels x 64 (k-step) per workgroup,
// 8 waves = 2 (pixels) x 4 (channels), wave tile 128 pixels x 64 channels, one workgroup per CU (128 KB LDS).
//
// Why a second main loop: in igemm_kernel (conv_kernels.hip) every wave of a workgroup issues its LDS-DMA pieces, reads its
// fragments and runs its MFMAs in the same order between the same barriers, so the texture path, the LDS and the matrix pipe are
// used in turn (profiles/r01_dx_kernel_phase_stamps.txt).  Here a k-step is cut into four phases of 16 MFMAs, each phase is
// [fragment reads + 2 LDS-DMA pieces + counted vmcnt] s_barrier [16 MFMAs] s_barrier, and the waves of the second pixel half
// (waves 4-7: the SIMD partners of waves 0-3) run ONE barrier behind: while one wave of a SIMD is in its MFMA segment its
// partner is in its load segment (MI355X_MICROARCH.md, "Two waves per SIMD").
//
// LDS: two k-step buffers of [X half0 | X half1 | W half0 | W half1], 16 KB each.  "half h" of the pixel tile = the rows every
// wave reads in its sub-phase h (tile rows wm*128 + h*64 + 0..63), likewise for the weight rows (wn*64 + h*32 + 0..31), so a
// half is free for re-staging as soon as that sub-phase has been read.
//   phase 1: reads X0, W0   MFMA (W0, X0)    issues W1 of tile t+1
//   phase 2: reads W1       MFMA (W1, X0)    issues X1 of tile t+1
//   phase 3: reads X1       MFMA (W1, X1)    issues X0 of tile t+2   (X0 of tile t last read in phase 1: two phases ago)
//   phase 4: --             MFMA (W0, X1)    issues W0 of tile t+2
// Every phase issues 2 pieces per wave and then waits vmcnt(8): the half-tile issued four phases earlier has landed in every
// wave before the phase's first barrier, and it is read one phase later (RAW through counted vmcnt + barrier; WAR by
// re-staging a half no earlier than two phases after its last read, which covers the one-barrier stagger).
// Addressing, zero padding (out-of-range buffer offsets), swizzle and the epilogues are those of igemm_kernel.
//
// Tile heights below 256 (round 4): T1A / T1B = the number of 16-pixel tiles in X half 1 of the waves with wm = 0 / 1
// (half 0 always has four), i.e. a tile of 128 + 16 * (T1A + T1B) pixels: (4, 4) = 256, (3, 3) = 224, (3, 2) = 208 ((2, 2) = 192 was measured and dropped).  The LDS
// image keeps its 64-row slots per wave half (rows a wave does not have are staged as out-of-range lanes: zeros, no memory traffic), the
// phases that multiply X half 1 run 4 * T1 MFMAs instead of 16, and the epilogue skips the missing tiles.  With 208-pixel tiles the three
// big layer families have 985 / 494 / 248 tiles instead of 800 / 400 / 200 on 256 CUs (0.96 / 0.96 / 0.97 of whole rounds instead of 0.78);
// what that buys against the per-tile costs that do not shrink (weight staging, prologue, epilogue) is in profiles/r04_ab_results.md.
//
// Tile quantisation: the big Darknet layers have 800 / 400 / 200 tiles on 256 CUs.  A stream-K form of this kernel (one persistent
// workgroup per CU over (tile, k-step) units, fp32 slab hand-off) and a persistent whole-tile form were built and measured in round 2
// (profiles/r02_streamk_timeline.txt, r02_igemm8_persistent_whole_tiles.txt): slower on every YOLO shape, removed in round 3.
#include "igemm_common.h"

struct SkParams {
  unsigned long long* dbg;   // diagnostic builds: per-workgroup stamps (mi355det_debug_ptr key 1), or null
};

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

constexpr int kBM = 256, kBN = 256, kBK = 64, kROWB = 128;
constexpr int kHALF = 128 * kROWB;            // 16 KB: one half-tile
constexpr int kSTAGE = 4 * kHALF;             // 64 KB: one k-step
constexpr int kLDS = 2 * kSTAGE;              // 128 KB

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int PROF_>
__device__ __forceinline__ void bar_b() {      // barrier in front of an MFMA segment (ablation 8 drops it)
  if (PROF_ == 8) return;
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int PROF_>
__device__ __forceinline__ void bar_d() {      // barrier behind an MFMA segment (ablations 7 and 8 drop it)
  if (PROF_ == 7 || PROF_ == 8) return;
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void bar() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int V>
struct IntC {
  static constexpr int value = V;
};

template <int EPI, int PROF = 0, int T1A = 4, int T1B = 4>
__global__ __launch_bounds__(512, 2) void igemm8_kernel(const IgemmParams p, const SkParams sk) {
  constexpr int WM = 2, WN = 4, TM = 8, TN = 4;
  constexpr int BM = 128 + 16 * (T1A + T1B);              // pixel-tile height (kBM for the full tile)
  constexpr int WB1 = 64 + 16 * T1A;                       // first tile row of the second wave half
  constexpr bool VAR = BM != kBM;
  static_assert(T1A >= 1 && T1A <= 4 && T1B >= 1 && T1B <= 4 && T1A >= T1B, "X half 1 has 1..4 tiles per wave half");
  static_assert(!VAR || PROF == 0, "the diagnostic builds exist for the full tile only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid0 = threadIdx.x;

  const int ntn = p.CoutPad / kBN, nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    // blocks b and b + 8 share an XCD: give each XCD a contiguous run of tiles (L2 reuse of pixel and weight tiles); bijective for any grid size
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int Ktot = p.T * p.Cin;
  const int ksteps = Ktot / kBK, cin_steps = p.Cin / kBK;

  constexpr bool SK = false;      // (the stream-K form is gone; the segment variables below keep its (tile, k0, k1) vocabulary)
  unsigned u = (unsigned)bid * (unsigned)ksteps;
  const unsigned u_end = u + (unsigned)ksteps;

  unsigned long long pk_t0 = 0, pk_t1 = 0, pk_t2 = 0;
  if (PROF == 2 && !SK) pk_t0 = __builtin_amdgcn_s_memrealtime();      // workgroup start (100 MHz)
  while (u < u_end) {
    const int tile = (int)(u / (unsigned)ksteps);
    const int k0 = (int)(u - (unsigned)tile * (unsigned)ksteps);
    const int k1 = min(ksteps, k0 + (int)(u_end - u));
    u += (unsigned)(k1 - k0);
    // lane-derived values are re-derived per segment from an opaque copy of the thread index: kept live across the main loop and
    // the epilogue of every segment they cost registers the stream-K form does not have (the accumulators fill half the file)
    int tid = tid0;
    if (SK) asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;     // waves 0-3 = pixel half 0, waves 4-7 = pixel half 1 (SIMD partners)
    const int lrow = lane >> 3, cpos = lane & 7;
    const int fr = lane & 15, fq = lane >> 4;
    const int fsw = (fr >> 1) & 7;
    int mt = tile / ntn, nt = tile - mt * ntn;
    if (ntn > 8) {
      // very wide outputs (the 1204-class head: 43 channel tiles): the 32 workgroups an XCD runs at once would be ONE pixel tile x 32 weight
      // tiles (33 operand tiles through its L2 per round).  Order the tiles so that 32 consecutive ones form an 8 x 4 block (12 operand
      // tiles): channel-tile groups of 4 (the last group may be narrower), inside a group panels of 8 pixel tiles.
      const int gmt = (p.M + BM - 1) / BM;
      const int full = gmt * 4, ngrp = (ntn + 3) / 4;
      int g = tile / full;
      if (g > ngrp - 1) g = ngrp - 1;
      const int t2 = tile - g * full;
      const int w = (g == ngrp - 1 && (ntn & 3)) ? (ntn & 3) : 4;
      const int panel = t2 / (8 * w), r = t2 - panel * 8 * w;
      const int rr = __builtin_amdgcn_readfirstlane(r / w);
      mt = panel * 8 + rr;
      nt = g * 4 + (r - rr * w);
    }
    const int m0 = mt * BM, n0 = nt * kBN;

    const int n_first = (int)fdiv(fdiv((unsigned)m0, p.dMW), p.dMH);
    const srd_t rsrc_x = make_srd(p.x + (long long)n_first * p.Hin * p.Win * p.ldin - p.tap_pad, 0x7FFFFFF0u);
    const srd_t rsrc_w = make_srd(p.w + (long long)n0 * Ktot, 0x7FFFFFF0u);

    // ---- rows this lane stages: per half h two pieces (8 rows of 128 B each); LDS row r' of a half <-> tile row
    int a_voff[2][2], b_voff[2][2];
    unsigned a_valid[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r2 = (2 * wid + i) * 8 + lrow;                  // row inside the half, 0..127
        const int sw = ((h * 128 + r2) >> 1) & 7;
        {
          // pixel row of the tile: wave half (r2 >> 6) owns tile rows [0, WB1) / [WB1, BM), its X half h the rows h * 64 + 0..63 of those
          const int row = VAR ? ((r2 >> 6) ? WB1 : 0) + h * 64 + (r2 & 63) : (r2 >> 6) * 128 + h * 64 + (r2 & 63);
          const bool have = !VAR || h == 0 || (r2 & 63) < 16 * ((r2 >> 6) ? T1B : T1A);      // rows of X half 1 that exist
          const int m = have ? m0 + row : 0x7FFFFFFF;
          unsigned vm = 0;
          int voff = OOB_VOFF;
          if (p.lin_in) {
            if (m < p.M) {
              voff = ((m - n_first * p.Hin * p.Win) * p.ldin + (cpos ^ sw) * 8) * 2;
              vm = 1u;
            }
          } else if (m < p.M) {
            const int t1 = (int)fdiv((unsigned)m, p.dMW), xx = m - t1 * p.MW;
            const int n = (int)fdiv((unsigned)t1, p.dMH), yy = t1 - n * p.MH;
            const int iy0 = yy * p.sin, ix0 = xx * p.sin;
            voff = ((((n - n_first) * p.Hin + iy0) * p.Win + ix0) * p.ldin + (cpos ^ sw) * 8) * 2;
#pragma unroll
            for (int t = 0; t < MAX_TAPS; ++t) {
              if (t >= p.T) break;
              const int dyt = (int)((p.dy_pack >> (4 * t)) & 0xF) - 2, dxt = (int)((p.dx_pack >> (4 * t)) & 0xF) - 2;
              const bool ok = (unsigned)(iy0 + dyt) < (unsigned)p.Hin && (unsigned)(ix0 + dxt) < (unsigned)p.Win;
              vm |= ok ? (1u << t) : 0u;
            }
          }
          a_voff[h][i] = voff;
          a_valid[h][i] = vm;
        }
        {
          const int row = (r2 >> 5) * 64 + h * 32 + (r2 & 31);    // channel row of the tile
          b_voff[h][i] = (row * Ktot + (cpos ^ sw) * 8) * 2;
        }
      }

    // ---- fragment read addresses (byte offsets into a k-step buffer, k-substep 0; substep 1 = ^64)
    int xrd = (wm * 64 + fr) * kROWB + ((fq ^ fsw) << 4);                  // + h*kHALF + jj*16*kROWB
    int wrd = 2 * kHALF + (wn * 32 + fr) * kROWB + ((fq ^ fsw) << 4);      // + h*kHALF + ii*16*kROWB

    // scalar state of the k-steps being prefetched: tile t+1 (slot 1) and t+2 (slot 2)
    auto tap_off = [&](int tap) {
      const int dyt = (int)((p.dy_pack >> (4 * tap)) & 0xF) - 2, dxt = (int)((p.dx_pack >> (4 * tap)) & 0xF) - 2;
      return ((dyt * p.Win + dxt) * p.ldin + p.tap_pad) * 2;
    };
    int pf_s = k0, pf_t = k0 / cin_steps, pf_c = k0 - pf_t * cin_steps;   // (k-step, tap, cin-step) of the NEXT k-step to set up
    struct Slot { int tap, soffx, soffw, live; };
    auto next_slot = [&]() {
      Slot s;
      s.live = pf_s < k1;
      s.tap = s.live ? pf_t : 31;                   // bit 31 of the tap masks is never set: a dead k-step stages zeros
      s.soffx = s.live ? tap_off(pf_t) + pf_c * (kBK * 2) : 0;
      s.soffw = s.live ? pf_s * (kBK * 2) : 0;
      ++pf_s;
      if (++pf_c == cin_steps) {
        pf_c = 0;
        ++pf_t;
      }
      return s;
    };
    // (measured and removed: issuing the second piece of every half-tile between the MFMAs of the same phase, with vmcnt(7), is 3-8 %
    //  slower on every shape - an LDS-DMA piece among MFMAs stalls the matrix pipe for longer than it shortens the load segment)
    // ablation builds (PROF 3 / 4 / 5, results are garbage, timing only): the loop without its LDS-DMA, without its fragment reads, without
    // its MFMAs (tools/prof_ig8.py)
    auto issue_x = [&](const Slot& s, int h, int buf) {
      if (PROF == 3) return;
      char* dst = smem + buf * kSTAGE + h * kHALF + (2 * wid) * 1024;
#pragma unroll
      for (int i = 0; i < 2; ++i) bufld16(rsrc_x, dst + i * 1024, ((a_valid[h][i] >> s.tap) & 1u) ? a_voff[h][i] : OOB_VOFF, s.soffx);
    };
    auto issue_w = [&](const Slot& s, int h, int buf) {
      if (PROF == 3) return;
      char* dst = smem + buf * kSTAGE + (2 + h) * kHALF + (2 * wid) * 1024;
#pragma unroll
      for (int i = 0; i < 2; ++i) bufld16(rsrc_w, dst + i * 1024, s.live ? b_voff[h][i] : OOB_VOFF, s.soffw);
    };

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = {0.f, 0.f, 0.f, 0.f};

    // ---- prologue: X0(0) W0(0) W1(0) X1(0) X0(1) W0(1), the steady-state issue order
    Slot s1 = next_slot();          // k-step k0
    issue_x(s1, 0, 0);
    issue_w(s1, 0, 0);
    issue_w(s1, 1, 0);
    issue_x(s1, 1, 0);
    s1 = next_slot();               // k0 + 1
    issue_x(s1, 0, 1);
    issue_w(s1, 0, 1);
    Slot s2 = next_slot();          // k0 + 2
    wait_vm<6>();                   // X0, W0 and W1 of the first k-step: the leading half reads W1 before the lagging half's next counted wait
    bar();
    if (wm == 1) bar();             // the second pixel half runs one barrier behind

    st16x8_t xf[2][4][2], wf[2][2][2];    // [half][tile][k-substep]
    // (measured and removed: software-pipelining the quarters by one phase - phase 1 multiplying the (W1, X1) fragments of the previous
    //  k-step so that every fragment read has a whole barrier interval to land - keeps all 96 fragment VGPRs live, hits the 256-VGPR cap
    //  with spills and is equal within noise: fragment-read latency is not what the loop waits for)
    auto read_x = [&](int h) {
      if (PROF == 4) return;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) xf[h][jj][ks] = *(const st16x8_t*)(smem + (xrd ^ (ks << 6)) + h * kHALF + jj * 16 * kROWB);
    };
    auto read_w = [&](int h) {
      if (PROF == 4) return;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) wf[h][ii][ks] = *(const st16x8_t*)(smem + (wrd ^ (ks << 6)) + h * kHALF + ii * 16 * kROWB);
    };
    // fragment reads at (half, k-substep) granularity: 2 W tiles / 4 X tiles of 16 rows each; `nxt` addresses the other k-step buffer
    auto rdw = [&](int h, int ks, bool nxt) {
      if (PROF == 4) return;
      const int base = nxt ? (wrd ^ kSTAGE) : wrd;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) wf[h][ii][ks] = *(const st16x8_t*)(smem + (base ^ (ks << 6)) + h * kHALF + ii * 16 * kROWB);
    };
    auto rdx = [&](int h, int ks, bool nxt, auto ntc) {
      if (PROF == 4) return;
      constexpr int NT = decltype(ntc)::value;            // 16-pixel tiles of this X half (4, or T1 for half 1)
      const int base = nxt ? (xrd ^ kSTAGE) : xrd;
#pragma unroll
      for (int jj = 0; jj < NT; ++jj) xf[h][jj][ks] = *(const st16x8_t*)(smem + (base ^ (ks << 6)) + h * kHALF + jj * 16 * kROWB);
    };
    // the 2 * NT MFMAs of one (W half, X half, k-substep)
    auto mf = [&](int hw, int hx, int ks, auto ntc) {
      if (PROF == 5) return;
      constexpr int NT = decltype(ntc)::value;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < NT; ++jj)
          acc[hw * 2 + ii][hx * 4 + jj] = MI355_MFMA_16x16x32(wf[hw][ii][ks], xf[hx][jj][ks], acc[hw * 2 + ii][hx * 4 + jj]);
    };
    // scheduling hint for a segment of `nm` MFMAs and `nr` independent fragment reads written before it: one read after each of the
    // first MFMAs (a ds_read_b128 fits the 16-cycle shadow of an MFMA), the rest of the MFMAs behind
#define IG8_INTERLEAVE(nm, nr)                                                     \
  do {                                                                            \
    _Pragma("unroll") for (int q_ = 0; q_ < ((nr) < (nm) ? (nr) : (nm)); ++q_) {   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
    }                                                                             \
    if constexpr ((nm) > (nr)) __builtin_amdgcn_sched_group_barrier(0x008, (nm) > (nr) ? (nm) - (nr) : 0, 0); \
    if constexpr ((nr) > (nm)) __builtin_amdgcn_sched_group_barrier(0x100, (nr) > (nm) ? (nr) - (nm) : 0, 0); \
  } while (0)
    auto mfma_q = [&](int hw, int hx) {
      if (PROF == 5) return;
      if (PROF == 6) {
        // timing-only ablation: the same FLOPs as 8 v_mfma_f32_32x32x16_bf16 (32 cycles each, the vector issue port held for 8 of them)
        // instead of 16 v_mfma_f32_16x16x32_bf16 (16 cycles, 8 held); operands and accumulator views are NOT a valid product
        typedef __attribute__((ext_vector_type(16))) float f32x16_t;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int jj = 0; jj < 4; jj += 2) {
              f32x16_t* a16 = (f32x16_t*)&acc[hw * 2 + ii][hx * 4 + (jj & 2) * 2];
              *a16 = MI355_MFMA_32x32x16(wf[hw][ii][ks], xf[hx][jj + ii][ks], *a16);      // every fragment read stays live
            }
        __builtin_amdgcn_s_setprio(0);
        return;
      }
      __builtin_amdgcn_s_setprio(1);      // measured: no priority, priority on the load segments instead, or waves 4-7 raised for the whole loop are all within +-1 %
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            acc[hw * 2 + ii][hx * 4 + jj] = MI355_MFMA_16x16x32(wf[hw][ii][ks], xf[hx][jj][ks], acc[hw * 2 + ii][hx * 4 + jj]);
      __builtin_amdgcn_s_setprio(0);
    };

    // ---- diagnostic build (PROF, non-stream-K): s_memtime around the four segments of every phase - (A) fragment reads + LDS-DMA
    //      issue, (B) counted vmcnt wait + barrier, (C) the 16 MFMAs, (D) trailing barrier - summed over the tile per wave, plus the 17
    //      absolute stamps of ONE k-step, so the timelines of the two waves of a SIMD can be laid side by side (tools/prof_ig8.py).
    //      Each stamp drains lgkmcnt, i.e. segment A includes the LDS read latency that the release build leaves in flight.
    constexpr bool PH = PROF == 1 && !SK;      // PROF == 2: only the start of 12 consecutive k-steps (near-release timing)
    constexpr bool PK = PROF == 2 && !SK;
    unsigned ph_sum[4] = {0, 0, 0, 0}, ph_cap[17], ph_prev = 0;      // low 32 bits of the counter: spans are far below 2^32 cycles
    unsigned long long ph_now = 0, ph_rt0 = 0, ph_rt1 = 0;
    if (PROF == 2 && !SK) pk_t1 = __builtin_amdgcn_s_memrealtime();    // main loop starts (prologue loads issued and landed)
#pragma unroll
    for (int i = 0; i < 17; ++i) ph_cap[i] = 0;
    const int ph_t = k0 + (PROF == 2 ? 4 : 6);
#define PH_STAMP(seg, capi)                                                                                     \
  do {                                                                                                          \
    if (PH) {                                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_now)::"memory");                            \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      if ((seg) >= 0) ph_sum[(seg) < 0 ? 0 : (seg)] += (unsigned)ph_now - ph_prev;                               \
      if (t == ph_t) ph_cap[capi] = (unsigned)ph_now;                                                           \
      ph_prev = (unsigned)ph_now;                                                                               \
    }                                                                                                           \
  } while (0)
    // The fragment reads live INSIDE the MFMA segments (a k-substep ahead of their use), not in the load segments: with the partner
    // wave streaming MFMAs, every instruction of a load segment costs the wave ~16 cycles, and reads + LDS-DMA together made that
    // segment longer than the 256 cycles of MFMAs it runs beside (ablation: the loop without its reads is MFMA-bound).  Order per k-step
    // (quarters Q1 = W0 X0, Q2 = W1 X0, Q3 = W1 X1, Q4 = W0 X1; k0 / k1 = the two k-substeps):
    //   phase 1: Q1k0 + reads W0k1, W1k0 | Q1k1 + reads W1k1        phase 2: Q2k0 + reads X1k0 | Q2k1 + reads X1k1
    //   phase 3: Q3                                                    phase 4: Q4k0 + reads NEXT X0k0 | Q4k1 + reads NEXT W0k0, X0k1
    // A fragment register set is re-loaded only after its last use, so at most 72 fragment VGPRs are live (64 before).  The reads of the
    // leading half now come one barrier earlier than the lagging half's counted wait used to cover, hence the vmcnt(6) in front of every
    // trailing barrier (the pieces of the last three phases may stay in flight); no region is read later than before.
    // The loop body per X-half-1 tile count: the two wave halves of a 208-pixel tile have 3 and 2 tiles there, so each half runs its own
    // instantiation (wave-uniform branch; the full tile has one).
    auto main_loop = [&](auto t1c) {
    constexpr int T1 = decltype(t1c)::value;
    const IntC<4> N4{};
    const IntC<T1> N1{};
    rdw(0, 0, false);
    rdx(0, 0, false, N4);
    rdx(0, 1, false, N4);
    int buf = 0;
    for (int t = k0; t < k1; ++t) {
      PH_STAMP(-1, 0);
      if (PK && (unsigned)(t - ph_t) < 12u) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_now)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        ph_cap[t - ph_t] = (unsigned)ph_now;
        if (t == ph_t) ph_rt0 = __builtin_amdgcn_s_memrealtime();           // 100 MHz wall clock: calibrates the cycle counter under load
        if (t == ph_t + 11) ph_rt1 = __builtin_amdgcn_s_memrealtime();
      }
      // ---- phase 1
      issue_w(s1, 1, buf ^ 1);
      PH_STAMP(0, 1);
      wait_vm<8>();
      bar_b<PROF>();
      PH_STAMP(1, 2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      rdw(0, 1, false);
      rdw(1, 0, false);
      mf(0, 0, 0, N4);
      IG8_INTERLEAVE(8, 4);
      rdw(1, 1, false);
      mf(0, 0, 1, N4);
      IG8_INTERLEAVE(8, 2);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      PH_STAMP(2, 3);
      wait_vm<6>();                 // the partner half reads, right after this barrier, fragments of tiles this wave helped to stage
      bar_d<PROF>();
      PH_STAMP(3, 4);
      // ---- phase 2
      issue_x(s1, 1, buf ^ 1);
      PH_STAMP(0, 5);
      wait_vm<8>();
      bar_b<PROF>();
      PH_STAMP(1, 6);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      rdx(1, 0, false, N1);
      mf(1, 0, 0, N4);
      IG8_INTERLEAVE(8, T1);
      rdx(1, 1, false, N1);
      mf(1, 0, 1, N4);
      IG8_INTERLEAVE(8, T1);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      PH_STAMP(2, 7);
      wait_vm<6>();                 // the partner half reads, right after this barrier, fragments of tiles this wave helped to stage
      bar_d<PROF>();
      PH_STAMP(3, 8);
      // ---- phase 3
      issue_x(s2, 0, buf);
      PH_STAMP(0, 9);
      wait_vm<8>();
      bar_b<PROF>();
      PH_STAMP(1, 10);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      mf(1, 1, 0, N1);
      mf(1, 1, 1, N1);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      PH_STAMP(2, 11);
      wait_vm<6>();                 // the partner half reads, right after this barrier, fragments of tiles this wave helped to stage
      bar_d<PROF>();
      PH_STAMP(3, 12);
      // ---- phase 4 (its wait publishes W0 / X0 of the next k-step: their fragments are fetched here, from the other buffer)
      issue_w(s2, 0, buf);
      PH_STAMP(0, 13);
      wait_vm<8>();
      bar_b<PROF>();
      PH_STAMP(1, 14);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      rdx(0, 0, true, N4);
      mf(0, 1, 0, N1);
      IG8_INTERLEAVE(2 * T1, 4);
      rdw(0, 0, true);
      rdx(0, 1, true, N4);
      mf(0, 1, 1, N1);
      IG8_INTERLEAVE(2 * T1, 6);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      PH_STAMP(2, 15);
      wait_vm<6>();                 // the partner half reads, right after this barrier, fragments of tiles this wave helped to stage
      bar_d<PROF>();
      PH_STAMP(3, 16);
      s1 = s2;
      s2 = next_slot();
      buf ^= 1;
      xrd ^= kSTAGE;
      wrd ^= kSTAGE;
    }
    };
    if constexpr (T1A == T1B) {
      main_loop(IntC<T1A>{});
    } else {
      if (wm == 0) main_loop(IntC<T1A>{});
      else main_loop(IntC<T1B>{});
    }
#undef PH_STAMP
#undef IG8_INTERLEAVE
    if (PK) pk_t2 = __builtin_amdgcn_s_memrealtime();                  // main loop done
    if ((PH || PK) && sk.dbg && lane == 0 && bid < 64) {
      unsigned long long* d = sk.dbg + ((size_t)bid * 8 + wid) * 24;
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i] = ph_sum[i];
      d[4] = (unsigned long long)(k1 - k0);
#pragma unroll
      for (int i = 0; i < 17; ++i) d[5 + i] = ph_cap[i];
      d[22] = ph_rt0;
      d[23] = ph_rt1;
      if (PK) {
        d[17] = pk_t0;
        d[18] = pk_t1;
        d[19] = pk_t2;
      }
    }
    if (wm == 0) bar();             // re-align the two halves
    wait_vm<0>();                   // the zero-fill pieces of the dead k-steps
    __builtin_amdgcn_s_waitcnt(0xC07F);
    bar();                          // the epilogue reuses smem

    igemm_epilogue<WM, WN, TM, TN, EPI, VAR ? BM : 0>(p, acc, smem, tid, 512, true, wm, wn, lane, mt, n0, m0, 0, wm ? WB1 : 0, 4 + (wm ? T1B : T1A));
    if (PROF == 2 && !SK && sk.dbg && lane == 0 && bid < 64) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's stores have left
      sk.dbg[((size_t)bid * 8 + wid) * 24 + 20] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

unsigned long long* g_sk_dbg = nullptr;
int g_sk_dbg_mode = 1;
template <int EPI>
int launch8(const IgemmParams& p, hipStream_t st) {
  const int gm = (p.M + kBM - 1) / kBM, gn = p.CoutPad / kBN;
  SkParams sk{};
  if (EPI == EPI_STATS && g_sk_dbg) {
    // diagnostic builds (tools/prof_ig8.py): [64 workgroups][8 waves][24] u64; mode 1 = phase stamps, 2 = k-step starts only
    sk.dbg = g_sk_dbg;
    static DeviceOnce attr_done_p;
    attr_done_p.once([&] {
      (void)hipFuncSetAttribute((const void*)igemm8_kernel<EPI_STATS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLDS);
      (void)hipFuncSetAttribute((const void*)igemm8_kernel<EPI_STATS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, kLDS);
    });
    if (g_sk_dbg_mode >= 3) {
      static DeviceOnce attr_done_a;
      attr_done_a.once([&] {
        (void)hipFuncSetAttribute((const void*)igemm8_kernel<EPI_STATS, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLDS);
        (void)hipFuncSetAttribute((const void*)igemm8_kernel<EPI_STATS, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kLDS);
        (void)hipFuncSetAttribute((const void*)igemm8_kernel<EPI_STATS, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, kLDS);
        (void)hipFuncSetAttribute((const void*)igemm8_kernel<EPI_STATS, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, kLDS);
        (void)hipFuncSetAttribute((const void*)igemm8_kernel<EPI_STATS, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, kLDS);
        (void)hipFuncSetAttribute((const void*)igemm8_kernel<EPI_STATS, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, kLDS);
      });
      if (g_sk_dbg_mode == 3) hipLaunchKernelGGL((igemm8_kernel<EPI_STATS, 3>), dim3(gm * gn), dim3(512), kLDS, st, p, sk);
      else if (g_sk_dbg_mode == 4) hipLaunchKernelGGL((igemm8_kernel<EPI_STATS, 4>), dim3(gm * gn), dim3(512), kLDS, st, p, sk);
      else if (g_sk_dbg_mode == 6) hipLaunchKernelGGL((igemm8_kernel<EPI_STATS, 6>), dim3(gm * gn), dim3(512), kLDS, st, p, sk);
      else if (g_sk_dbg_mode == 7) hipLaunchKernelGGL((igemm8_kernel<EPI_STATS, 7>), dim3(gm * gn), dim3(512), kLDS, st, p, sk);
      else if (g_sk_dbg_mode == 8) hipLaunchKernelGGL((igemm8_kernel<EPI_STATS, 8>), dim3(gm * gn), dim3(512), kLDS, st, p, sk);
      else hipLaunchKernelGGL((igemm8_kernel<EPI_STATS, 5>), dim3(gm * gn), dim3(512), kLDS, st, p, sk);
      return check_launch("igemm8_ablation");
    }
    if (g_sk_dbg_mode == 2) hipLaunchKernelGGL((igemm8_kernel<EPI_STATS, 2>), dim3(gm * gn), dim3(512), kLDS, st, p, sk);
    else hipLaunchKernelGGL((igemm8_kernel<EPI_STATS, 1>), dim3(gm * gn), dim3(512), kLDS, st, p, sk);
    return check_launch("igemm8_prof");
  }
  auto k = igemm8_kernel<EPI>;
  static DeviceOnce attr_done;
  attr_done.once([&] {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLDS);
  });
  hipLaunchKernelGGL(k, dim3(gm * gn), dim3(512), kLDS, st, p, sk);
  return check_launch("igemm8");
}

// the 224 / 208-pixel tiles (T1A, T1B) = (3, 3) / (3, 2)
template <int EPI, int T1A, int T1B>
int launch8v(const IgemmParams& p, hipStream_t st) {
  constexpr int BM = 128 + 16 * (T1A + T1B);
  const int gm = (p.M + BM - 1) / BM, gn = p.CoutPad / kBN;
  SkParams sk{};
  auto k = igemm8_kernel<EPI, 0, T1A, T1B>;
  static DeviceOnce attr_done;
  attr_done.once([&] {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLDS);
  });
  hipLaunchKernelGGL(k, dim3(gm * gn), dim3(512), kLDS, st, p, sk);
  return check_launch("igemm8v");
}

template <int EPI>
int launch8_rows(const IgemmParams& p, hipStream_t st, int rows) {
  switch (rows) {
    case 256: return launch8<EPI>(p, st);
    case 224: return launch8v<EPI, 3, 3>(p, st);
    case 208: return launch8v<EPI, 3, 2>(p, st);
    default: break;
  }
  return fail(MI355DET_EINVAL, "%s: pixel-tile height %lld not built (256, 224, 208)", "igemm8", rows);
}

}  // namespace

bool igemm8_applicable(const IgemmParams& p) {
  if (!(p.CoutPad % kBN == 0 && p.Cin % kBK == 0 && p.T >= 1 && p.T <= MAX_TAPS)) return false;
  const long long units = (long long)((p.M + kBM - 1) / kBM) * (p.CoutPad / kBN) * (p.T * p.Cin / kBK);
  return units < (1ll << 31);      // 32-bit unit arithmetic in the kernel
}

void igemm8_set_dbg(unsigned long long* ptr) { g_sk_dbg = ptr; }
void igemm8_set_dbg_mode(int mode) { g_sk_dbg_mode = mode; }

int igemm8_launch(int epi, const IgemmParams& p, hipStream_t st, int rows) {
  if (!igemm8_applicable(p)) return fail(MI355DET_EINVAL, "%s: shape not supported by the phase-staggered kernel", "igemm8");
  switch (epi) {
    case EPI_STATS: return launch8_rows<EPI_STATS>(p, st, rows);
    case EPI_PLAIN: return launch8_rows<EPI_PLAIN>(p, st, rows);
    case EPI_RES: return launch8_rows<EPI_RES>(p, st, rows);
    case EPI_AFF: return launch8_rows<EPI_AFF>(p, st, rows);
    case EPI_F32: return rows == 256 ? launch8<EPI_F32>(p, st) : fail(MI355DET_EINVAL, "%s: the fp32 head epilogue is built for the 256-pixel tile only", "igemm8");      // (the 1204-class cls_logits: Cout padded to 256)
    default: break;
  }
  return fail(MI355DET_EINVAL, "%s: epilogue not built for the phase-staggered kernel", "igemm8");
}
