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
#include "igemm_common.h"

namespace {

constexpr int kBM = 256, kBN = 256, kBK = 64, kROWB = 128;
constexpr int kHALF = 128 * kROWB;            // 16 KB: one half-tile
constexpr int kSTAGE = 4 * kHALF;             // 64 KB: one k-step
constexpr int kLDS = 2 * kSTAGE;              // 128 KB

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void bar() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void igemm8_kernel(const IgemmParams p) {
  constexpr int WM = 2, WN = 4, TM = 8, TN = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;     // waves 0-3 = pixel half 0, waves 4-7 = pixel half 1 (SIMD partners)

  const int ntn = p.CoutPad / kBN, nblk = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int mt = bid / ntn, nt = bid - mt * ntn;
  const int m0 = mt * kBM, n0 = nt * kBN;

  const int n_first = (int)fdiv(fdiv((unsigned)m0, p.dMW), p.dMH);
  const int Ktot = p.T * p.Cin;
  const srd_t rsrc_x = make_srd(p.x + (long long)n_first * p.Hin * p.Win * p.ldin - p.tap_pad, 0x7FFFFFF0u);
  const srd_t rsrc_w = make_srd(p.w + (long long)n0 * Ktot, 0x7FFFFFF0u);

  // ---- rows this lane stages: per half h two pieces (8 rows of 128 B each); LDS row r' of a half <-> tile row
  const int lrow = lane >> 3, cpos = lane & 7;
  int a_voff[2][2], b_voff[2][2];
  unsigned a_valid[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r2 = (2 * wid + i) * 8 + lrow;                  // row inside the half, 0..127
      const int sw = ((h * 128 + r2) >> 1) & 7;
      {
        const int row = (r2 >> 6) * 128 + h * 64 + (r2 & 63);   // pixel row of the tile
        const int m = m0 + row;
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
  const int fr = lane & 15, fq = lane >> 4;
  const int fsw = (fr >> 1) & 7;
  int xrd = (wm * 64 + fr) * kROWB + ((fq ^ fsw) << 4);                  // + h*kHALF + jj*16*kROWB
  int wrd = 2 * kHALF + (wn * 32 + fr) * kROWB + ((fq ^ fsw) << 4);      // + h*kHALF + ii*16*kROWB

  const int ksteps = Ktot / kBK, cin_steps = p.Cin / kBK;

  // scalar state of the k-steps being prefetched: tile t+1 (slot 1) and t+2 (slot 2)
  auto tap_off = [&](int tap) {
    const int dyt = (int)((p.dy_pack >> (4 * tap)) & 0xF) - 2, dxt = (int)((p.dx_pack >> (4 * tap)) & 0xF) - 2;
    return ((dyt * p.Win + dxt) * p.ldin + p.tap_pad) * 2;
  };
  int pf_t = 0, pf_c = 0, pf_s = 0;               // (tap, cin-step, k-step) of the NEXT tile to set up
  struct Slot { int tap, soffx, soffw, live; };
  auto next_slot = [&]() {
    Slot s;
    s.live = pf_s < ksteps;
    s.tap = s.live ? pf_t : 31;                   // bit 31 of the tap masks is never set: a dead tile stages zeros
    s.soffx = s.live ? tap_off(pf_t) + pf_c * (kBK * 2) : 0;
    s.soffw = s.live ? pf_s * (kBK * 2) : 0;
    ++pf_s;
    if (++pf_c == cin_steps) {
      pf_c = 0;
      ++pf_t;
    }
    return s;
  };
  // one half-tile (2 pieces of this wave): which = 0 pixels / 1 weights
  auto issue_x = [&](const Slot& s, int h, int buf) {
    char* dst = smem + buf * kSTAGE + h * kHALF + (2 * wid) * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i) bufld16(rsrc_x, dst + i * 1024, ((a_valid[h][i] >> s.tap) & 1u) ? a_voff[h][i] : OOB_VOFF, s.soffx);
  };
  auto issue_w = [&](const Slot& s, int h, int buf) {
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
  Slot s1 = next_slot();          // tile 0
  issue_x(s1, 0, 0);
  issue_w(s1, 0, 0);
  issue_w(s1, 1, 0);
  issue_x(s1, 1, 0);
  s1 = next_slot();               // tile 1
  issue_x(s1, 0, 1);
  issue_w(s1, 0, 1);
  Slot s2 = next_slot();          // tile 2
  wait_vm<8>();
  bar();
  if (wm == 1) bar();             // the second pixel half runs one barrier behind

  bf16x8_t xf[2][4][2], wf[2][2][2];    // [half][tile][k-substep]
  auto read_x = [&](int h) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) xf[h][jj][ks] = *(const bf16x8_t*)(smem + (xrd ^ (ks << 6)) + h * kHALF + jj * 16 * kROWB);
  };
  auto read_w = [&](int h) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) wf[h][ii][ks] = *(const bf16x8_t*)(smem + (wrd ^ (ks << 6)) + h * kHALF + ii * 16 * kROWB);
  };
  auto mfma_q = [&](int hw, int hx) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          acc[hw * 2 + ii][hx * 4 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[hw][ii][ks], xf[hx][jj][ks], acc[hw * 2 + ii][hx * 4 + jj], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  int buf = 0;
  for (int t = 0; t < ksteps; ++t) {
    // ---- phase 1
    read_w(0);
    __builtin_amdgcn_sched_barrier(0);
    read_x(0);
    issue_w(s1, 1, buf ^ 1);
    wait_vm<8>();
    bar();
    __builtin_amdgcn_sched_barrier(0);
    mfma_q(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    bar();
    // ---- phase 2
    read_w(1);
    issue_x(s1, 1, buf ^ 1);
    wait_vm<8>();
    bar();
    __builtin_amdgcn_sched_barrier(0);
    mfma_q(1, 0);
    __builtin_amdgcn_sched_barrier(0);
    bar();
    // ---- phase 3
    read_x(1);
    issue_x(s2, 0, buf);
    wait_vm<8>();
    bar();
    __builtin_amdgcn_sched_barrier(0);
    mfma_q(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    bar();
    // ---- phase 4
    issue_w(s2, 0, buf);
    wait_vm<8>();
    bar();
    __builtin_amdgcn_sched_barrier(0);
    mfma_q(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    bar();
    s1 = s2;
    s2 = next_slot();
    buf ^= 1;
    xrd ^= kSTAGE;
    wrd ^= kSTAGE;
  }
  if (wm == 0) bar();             // re-align the two halves
  wait_vm<0>();                   // the zero-fill pieces of the dead tiles
  __builtin_amdgcn_s_waitcnt(0xC07F);
  bar();                          // the epilogue reuses smem

  igemm_epilogue<WM, WN, TM, TN, EPI>(p, acc, smem, tid, 512, true, wm, wn, lane, mt, n0, m0);
}

template <int EPI>
int launch8(const IgemmParams& p, hipStream_t st) {
  const int gm = (p.M + kBM - 1) / kBM, gn = p.CoutPad / kBN;
  auto k = igemm8_kernel<EPI>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLDS);
    attr_done = true;
  }
  hipLaunchKernelGGL(k, dim3(gm * gn), dim3(512), kLDS, st, p);
  return check_launch("igemm8");
}

}  // namespace

bool igemm8_applicable(const IgemmParams& p) { return p.CoutPad % kBN == 0 && p.Cin % kBK == 0 && p.T >= 1 && p.T <= MAX_TAPS; }

int igemm8_launch(int epi, const IgemmParams& p, hipStream_t st) {
  if (!igemm8_applicable(p)) return fail(MI355DET_EINVAL, "%s: shape not supported by the phase-staggered kernel", "igemm8");
  switch (epi) {
    case EPI_STATS: return launch8<EPI_STATS>(p, st);
    case EPI_PLAIN: return launch8<EPI_PLAIN>(p, st);
    case EPI_RES: return launch8<EPI_RES>(p, st);
    case EPI_AFF: return launch8<EPI_AFF>(p, st);
    default: break;
  }
  return fail(MI355DET_EINVAL, "%s: epilogue not built for the phase-staggered kernel", "igemm8");
}
