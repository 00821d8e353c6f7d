// Data gradient of a 3x3 / stride-2 / pad-1 convolution with FEW channels (Cin 32 or 64, Cout 64: the largest feature map of
// Darknet-53), all four output-parity classes in ONE launch.
//
//   dx[n, 2yy+py, 2xx+px, ci] = sum over the taps (kh,kw) of class (py,px) and co of dy[n, yy+oy, xx+ox, co] * W[co, ci, kh, kw]
//   (py=0: kh=1, oy=0;  py=1: kh=0, oy=1 and kh=2, oy=0;  same in x)
//
// The general path runs one implicit GEMM per parity class: four launches that each re-stage the same dy tile per tap and whose
// per-tile set-up outweighs their 1-4 k-steps (0.72 ms for 32->64 @320 against 0.25 ms at the HBM floor; 0.36 ms with this kernel).  Here a
// persistent workgroup keeps all nine weight taps resident in LDS, stages ONE dy tile (TP lattice pixels of row yy and of row
// yy+1, plus one pixel to the right) per output tile with LDS-DMA - double-buffered across tiles - and the nine taps read it at
// row offsets 0/+1; a tile never straddles a row (TP divides Wo), so every bounds decision is wave-uniform.  The four classes of a
// tile are written as two whole output rows (2*TP consecutive pixels each) through an fp32 LDS transpose.
// Replaces, for these shapes, the four mi355det_conv_dgrad class launches (reference: the autograd of nn.Conv2d in
// nets/darknet.py:43-49, the stride-2 down-sampling convolutions).
#include "common.h"

using namespace mi355;

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) int srd_t;

struct DgradS2Params {
  const bf16_t* dy;   // [n, Ho, Wo, K] pitch lddy
  const bf16_t* w;    // dgrad pack of mi355det_pack_weights: per class [cin_pad][T_c * K]
  bf16_t* dx;         // [n, 2Ho, 2Wo, cin] pitch lddx
  const bf16_t* res;  // optional residual added before rounding (same indexing as dx, pitch ldres)
  int n, Ho, Wo, lddy, lddx, ldres, K, cin_pad;
  int tiles_per_row, ntiles;
};

namespace {

__device__ __forceinline__ srd_t make_srd(const void* base, unsigned num_records) {
  const unsigned long long a = (unsigned long long)base;
  srd_t r;
  r[0] = (int)(unsigned)a;
  r[1] = (int)((unsigned)(a >> 32) & 0xFFFFu);
  r[2] = (int)num_records;
  r[3] = 0x00020000;
  return r;
}
// inline assembly on purpose: see conv_kernels.hip (the compiler would drain vmcnt before every later LDS read)
__device__ __forceinline__ void bufld16(srd_t rsrc, const void* lds_dst_uniform, int voffset, int soffset) {
  const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)lds_dst_uniform);
  const int so = __builtin_amdgcn_readfirstlane(soffset);   // uniform by construction; integer divisions leave it in a VGPR
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voffset), "s"(rsrc), "s"(so) : "memory");
}
#define OOB_VOFF ((int)0x80000000)
// workgroup barrier for LDS hand-offs: this wave's LDS traffic has completed, and the compiler moves no memory access across it
// (a bare s_barrier builtin carries no memory semantics; __syncthreads() would also drain the global stores and the prefetch)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// taps in pack order (class-major): class, index inside the class, dy-lattice offsets
__device__ constexpr int kCls[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
__device__ constexpr int kJ[9] = {0, 0, 1, 0, 1, 0, 1, 2, 3};
__device__ constexpr int kOy[9] = {0, 0, 0, 1, 0, 1, 1, 0, 0};
__device__ constexpr int kOx[9] = {0, 1, 0, 0, 0, 1, 0, 1, 0};
__device__ constexpr int kT[4] = {1, 2, 2, 4};
__device__ constexpr int kWbase[4] = {0, 1, 3, 5};   // pack offset of a class in units of cin_pad * K

// TP lattice pixels per tile (divides Wo), NCH = K / 64.  256 threads = PXW waves along pixels x WC along the 32 channels.
template <int TP, int NCH, bool RES>
__global__ __launch_bounds__(256, NCH == 1 ? 2 : 1) void dgrad_s2_kernel(const DgradS2Params p) {
  constexpr int AR = (TP + 1 + 7) / 8 * 8;          // LDS rows per plane
  constexpr int ABUF = 2 * NCH * AR * 128;          // one tile: planes yy / yy+1, NCH 64-channel chunks
  constexpr int WBYTES = 9 * NCH * 32 * 128;
  constexpr int PXW = TP / 16 >= 4 ? 4 : TP / 16, WC = 4 / PXW, CHF = 2 / WC;
  static_assert(TP % 16 == 0 && PXW * WC == 4 && CHF >= 1, "wave split");
  static_assert(2 * TP * 32 * 4 <= ABUF && 2 * 2 * TP * 32 * 2 <= ABUF, "row staging must fit the dead pixel tile");
  static_assert((2 * TP * 4) % 256 == 0, "whole passes of the workgroup over a row");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wl = smem;
  char* const abase = smem + WBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wid % PXW, wc = wid / PXW;
  const int cobase = blockIdx.y * 32;               // the 32 output channels (conv input channels) of this workgroup
  const int K = p.K;

  const srd_t rsrc_dy = make_srd(p.dy, 0x7FFFFFF0u);
  const srd_t rsrc_w = make_srd(p.w, 0x7FFFFFF0u);
  const int lrow = lane >> 3, cpos = lane & 7;

  // ---- weights: once per workgroup
  for (int ii = wid; ii < 9 * NCH * 4; ii += 4) {
    const int t = ii / (NCH * 4), q = (ii / 4) % NCH, rb = ii % 4;
    const int ci = rb * 8 + lrow;
    const int c = kCls[t];
    const int voff = (kWbase[c] * p.cin_pad * K + (cobase + ci) * kT[c] * K + kJ[t] * K + q * 64 + ((cpos ^ ((ci >> 1) & 7)) << 3)) * 2;
    bufld16(rsrc_w, wl + ((t * NCH + q) * 32 + rb * 8) * 128, voff, 0);
  }

  // ---- pixel tile of one output tile: lane offsets are tile-independent, the tile enters through the scalar offset
  constexpr int RB = AR / 8, NI = 2 * NCH * RB;
  auto issue_tile = [&](int tile, char* ab) {
    const int row = tile / p.tiles_per_row, xt = tile - row * p.tiles_per_row;
    const int n = row / p.Ho, yy = row - n * p.Ho;
    const bool last_x = xt == p.tiles_per_row - 1, last_y = yy == p.Ho - 1;
    for (int ii = wid; ii < NI; ii += 4) {
      const int plane = ii / (NCH * RB), q = (ii / RB) % NCH, rb = ii % RB;
      const int r = rb * 8 + lrow;
      const bool ok = r <= TP && !(r == TP && last_x) && !(plane == 1 && last_y);
      const int voff = (r * p.lddy + q * 64 + ((cpos ^ lrow) << 3)) * 2;      // swizzle phase r & 7 == lrow
      const int soff = (((n * p.Ho + yy + plane) * p.Wo + xt * TP) * p.lddy) * 2;
      bufld16(rsrc_dy, ab + ((plane * NCH + q) * AR + rb * 8) * 128, ok ? voff : OOB_VOFF, (plane == 1 && last_y) ? 0 : soff);
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  int tile = blockIdx.x;
  int buf = 0;
  if (tile < p.ntiles) issue_tile(tile, abase);
  for (; tile < p.ntiles; tile += gridDim.x, buf ^= 1) {
    char* const ab = abase + buf * ABUF;
    // the pixel tile was issued BEFORE the previous tile's output stores (vmcnt retires in order): those NST stores per thread may
    // stay in flight (with a residual the compiler's own waits for the residual loads have already drained the older DMA)
    constexpr int NST = 2 * 2 * TP * 4 / 256;
    if (tile == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    lds_barrier();                                  // this tile (and the weights) landed; everyone is done with the other buffer
    const int next = tile + gridDim.x;
    if (next < p.ntiles) issue_tile(next, abase + (buf ^ 1) * ABUF);

    f32x4_t acc[4][CHF];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int h = 0; h < CHF; ++h) acc[c][h] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        st16x8_t af[2][2];
#pragma unroll
        for (int oy = 0; oy < 2; ++oy)
#pragma unroll
          for (int ox = 0; ox < 2; ++ox) {
            const int r = wp * 16 + fr + ox;
            af[oy][ox] = *(const st16x8_t*)(ab + ((oy * NCH + q) * AR + r) * 128 + (((ks * 4 + fq) ^ (r & 7)) << 4));
          }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
          for (int h = 0; h < CHF; ++h) {
            const int r = (wc * CHF + h) * 16 + fr;
            const st16x8_t wf = *(const st16x8_t*)(wl + ((t * NCH + q) * 32 + r) * 128 + (((ks * 4 + fq) ^ ((r >> 1) & 7)) << 4));
            acc[kCls[t]][h] = MI355_MFMA_16x16x32(wf, af[kOy[t]][kOx[t]], acc[kCls[t]][h]);
          }
        }
      }

    // ---- epilogue: the two output rows 2yy / 2yy+1 of this tile, each 2*TP consecutive pixels x 32 channels
    const int row = tile / p.tiles_per_row, xt = tile - row * p.tiles_per_row;
    const int n = row / p.Ho, yy = row - n * p.Ho;
    const long long opix0 = ((long long)(n * 2 * p.Ho + 2 * yy) * (2 * p.Wo) + 2 * xt * TP);   // first pixel of row 2yy; row 2yy+1 is 2*Wo further
    if (!RES) {
      // no residual: round in registers and transpose both rows in one pass through the dead pixel tile (bf16, 64 B per pixel)
      bf16_t* const stg = (bf16_t*)ab;
      lds_barrier();                                // every wave is done reading the pixel tile
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int h = 0; h < CHF; ++h) {
          const int pix = (c >> 1) * (2 * TP) + 2 * (wp * 16 + fr) + (c & 1);
          const int ch = (wc * CHF + h) * 16 + fq * 4;
          uint2 o;
          o.x = (unsigned)f2s(acc[c][h][0]) | ((unsigned)f2s(acc[c][h][1]) << 16);
          o.y = (unsigned)f2s(acc[c][h][2]) | ((unsigned)f2s(acc[c][h][3]) << 16);
          *(uint2*)(stg + pix * 32 + ch) = o;
        }
      lds_barrier();
#pragma unroll
      for (int e0 = 0; e0 < 2 * 2 * TP * 4; e0 += 256) {
        const int e = e0 + tid;
        const int pix = e >> 2, part = e & 3;
        const int py = pix >= 2 * TP, pl = pix - py * (2 * TP);
        const uint4 o = *(const uint4*)(stg + pix * 32 + part * 8);
        *(uint4*)(p.dx + (opix0 + (long long)py * (2 * p.Wo) + pl) * p.lddx + cobase + part * 8) = o;
      }
    } else {
      float* const stg = (float*)ab;
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        lds_barrier();
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int h = 0; h < CHF; ++h) {
            const int pix = 2 * (wp * 16 + fr) + px;
            const int ch = (wc * CHF + h) * 16 + fq * 4;
            *(f32x4_t*)(stg + pix * 32 + ch) = acc[py * 2 + px][h];
          }
        lds_barrier();
        const long long opix = opix0 + (long long)py * (2 * p.Wo);
#pragma unroll
        for (int e0 = 0; e0 < 2 * TP * 4; e0 += 256) {
          const int e = e0 + tid;
          const int pix = e >> 2, part = e & 3;
          const f32x4_t a = *(const f32x4_t*)(stg + pix * 32 + part * 8), b = *(const f32x4_t*)(stg + pix * 32 + part * 8 + 4);
          float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
          const uint4 ru = *(const uint4*)(p.res + (opix + pix) * p.ldres + cobase + part * 8);
          const unsigned rr[4] = {ru.x, ru.y, ru.z, ru.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[2 * k] += s2f((bf16_t)(rr[k] & 0xFFFF));
            v[2 * k + 1] += s2f((bf16_t)(rr[k] >> 16));
          }
          uint4 o;
          o.x = (unsigned)f2s(v[0]) | ((unsigned)f2s(v[1]) << 16);
          o.y = (unsigned)f2s(v[2]) | ((unsigned)f2s(v[3]) << 16);
          o.z = (unsigned)f2s(v[4]) | ((unsigned)f2s(v[5]) << 16);
          o.w = (unsigned)f2s(v[6]) | ((unsigned)f2s(v[7]) << 16);
          *(uint4*)(p.dx + (opix + pix) * p.lddx + cobase + part * 8) = o;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int TP, int NCH>
int launch(const DgradS2Params& p, int cin, hipStream_t st) {
  constexpr int AR = (TP + 1 + 7) / 8 * 8;
  constexpr int lds = 9 * NCH * 32 * 128 + 2 * (2 * NCH * AR * 128);
  const int per_cu = NCH == 1 ? 2 : 1;
  const int gx = min(p.ntiles, max(1, 256 * per_cu / (cin / 32)));   // persistent: one resident round over both channel halves
  auto go = [&](auto kern) {
    static DeviceOnce attr_done;
    attr_done.once([&] {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    hipLaunchKernelGGL(kern, dim3(gx, cin / 32), dim3(256), lds, st, p);
  };
  if (p.res) go(dgrad_s2_kernel<TP, NCH, true>);
  else go(dgrad_s2_kernel<TP, NCH, false>);
  return check_launch("dgrad_s2");
}

}  // namespace

int g_dgrad_s2_off = 0;   // diagnostic (mi355det_debug_set(2, 1)): always the four class launches (tests compare the two)

// 1 = handled, 0 = shape not covered (caller takes the general path), < 0 = error
int mi355det_internal_dgrad_s2(const mi355det_conv_shape* s, const void* dy, const void* wt, void* dx, const void* residual, int32_t residual_ld,
                               void* stream) {
  if (g_dgrad_s2_off || s->ksize != 3 || s->stride != 2 || s->pad != 1 || (s->h & 1) || (s->w & 1)) return 0;
  // Cout 128 (two 64-channel chunks: 74 KB of weights, one workgroup per CU) measured SLOWER than the four class launches
  // (518 vs 485 us on 64->128 @160): only the single-chunk case is dispatched
  if (!(s->cin == 32 || s->cin == 64) || s->cout != 64) return 0;
  if (s->ho * 2 != s->h || s->wo * 2 != s->w || s->out_ld % 8 || s->in_ld % 8 || (residual && residual_ld % 8)) return 0;
  const int nch = s->cout / 64;
  const int tp = nch == 1 ? (s->wo % 64 == 0 ? 64 : (s->wo % 32 == 0 ? 32 : 0)) : (s->wo % 32 == 0 ? 32 : 0);
  if (!tp) return 0;
  if ((long long)s->n * s->ho * s->wo * s->out_ld * 2 >= 0x7FFFFFF0ll) return 0;   // 31-bit byte offsets into dy
  DgradS2Params p{};
  p.dy = (const bf16_t*)dy;
  p.w = (const bf16_t*)wt;
  p.dx = (bf16_t*)dx;
  p.res = (const bf16_t*)residual;
  p.n = s->n; p.Ho = s->ho; p.Wo = s->wo;
  p.lddy = s->out_ld; p.lddx = s->in_ld; p.ldres = residual_ld;
  p.K = s->cout;
  p.cin_pad = (s->cin + 31) / 32 * 32;
  p.tiles_per_row = s->wo / tp;
  p.ntiles = s->n * s->ho * p.tiles_per_row;
  int e;
  if (nch == 1) e = tp == 64 ? launch<64, 1>(p, s->cin, S(stream)) : launch<32, 1>(p, s->cin, S(stream));
  else return 0;
  return e < 0 ? e : 1;
}
