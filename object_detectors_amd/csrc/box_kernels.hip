// Box kernels: NMS family (majority-vote NMS of the YOLO path, torchvision nms / batched_nms),
// pairwise box_iou, fused IoU+Matcher, BoxCoder, anchor grid, sigmoid focal loss.
// Wavefront-reduction / bit-mask kernels, HBM/latency-bound; -ffp-contract=off for bit-exact
// threshold decisions against the reference's unfused float32 arithmetic.
#include "common.h"

using namespace mi355;

namespace {

#define NMS_MAX_N 131072       // the n*n/8-byte suppression mask is 2 GiB per image at this size
#define NMS_SORT_CHUNK 16384    // boxes one workgroup sorts in LDS; larger inputs: chunk sorts + merge by rank (nms_merge_kernel)
#define SORT_THREADS 1024
#define RADIX_MIN_N 4096      // padded sizes from here on are sorted by the radix path of nms_sort_kernel
#define RADIX_CAP 16384        // = NMS_MAX_N

struct NmsWs {   // per-image workspace carve (all offsets in bytes, 16-B aligned)
  size_t sorted_idx, sbox, scls, mask, kept, remover, misc, diagt, keptbits, ckey, csrc, stride;
  int words;
};

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

__host__ __device__ inline NmsWs nms_layout(int max_n) {
  NmsWs w;
  w.words = (max_n + 63) / 64;
  size_t o = 0;
  w.sorted_idx = o; o = align16(o + sizeof(int) * (size_t)max_n);
  w.sbox = o;       o = align16(o + sizeof(float4) * (size_t)max_n);
  w.scls = o;       o = align16(o + sizeof(int) * (size_t)max_n);
  w.kept = o;       o = align16(o + sizeof(int) * (size_t)max_n);
  w.remover = o;    o = align16(o + sizeof(int) * (size_t)max_n);
  w.misc = o;       o = align16(o + 64);
  w.diagt = o;      o = align16(o + sizeof(unsigned long long) * (size_t)w.words * 64);   // transposed diagonal blocks: word j = earlier boxes of j's tile that drop j
  w.keptbits = o;   o = align16(o + sizeof(unsigned long long) * (size_t)w.words);        // per tile: which of its 64 boxes the scan kept
  w.ckey = o;       o = align16(o + (max_n > NMS_SORT_CHUNK ? sizeof(unsigned) * (size_t)max_n : 0));      // chunk-sorted keys / source rows (large inputs only)
  w.csrc = o;       o = align16(o + (max_n > NMS_SORT_CHUNK ? sizeof(int) * (size_t)max_n : 0));
  w.mask = o;       o = align16(o + sizeof(unsigned long long) * (size_t)w.words * 64 * w.words);   // [column block][row padded to 64]
  w.stride = o;
  return w;
}

// ---- 1. sort by score (single block bitonic in LDS), gather sorted boxes -------------------
// MODE 0: nms_majority rows [n,6]; order = reverse of a stable ascending argsort (helper.py:308,320)
// MODE 1: torchvision nms; descending score, ties -> lower index first; optional category offsets
// CHUNK: blockIdx.x = chunk of NMS_SORT_CHUNK consecutive positions of the initial order (index order for MODE 1, reverse index order for
// MODE 0); the workgroup radix-sorts its chunk and leaves (key, source row) in global memory for nms_merge_kernel.
template <int MODE, bool CHUNK = false>
__global__ __launch_bounds__(SORT_THREADS) void nms_sort_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                                 const long long* __restrict__ idxs, const int* __restrict__ count,
                                                                 int n_fixed, int max_n, char* __restrict__ ws_base, NmsWs L) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  __shared__ float smax[SORT_THREADS / WAVE];
  const int b = CHUNK ? blockIdx.y : blockIdx.x;
  const int n_total = count ? min(count[b], max_n) : n_fixed;
  const int p_base = CHUNK ? blockIdx.x * NMS_SORT_CHUNK : 0;
  const int n = CHUNK ? max(0, min(NMS_SORT_CHUNK, n_total - p_base)) : n_total;
  char* ws = ws_base + (size_t)b * L.stride;
  if (CHUNK && blockIdx.x == 0 && threadIdx.x == 0) ((int*)(ws + L.misc))[0] = n_total;
  if (CHUNK && n == 0) return;
  int npad = CHUNK ? RADIX_MIN_N : 64;
  while (npad < n) npad <<= 1;
  const float* P = boxes + (size_t)b * max_n * (MODE == 0 ? 6 : 4);
  if (MODE == 1) {            // batched form (mi355det_nms_batch): image b owns rows [b*max_n, (b+1)*max_n) of scores / idxs too
    scores += (size_t)b * max_n;
    if (idxs) idxs += (size_t)b * max_n;
  }
  for (int i = threadIdx.x; i < npad; i += SORT_THREADS) {
    unsigned long long k = 0;
    if (i < n) {
      const float sc = MODE == 0 ? P[(size_t)i * 6 + 4] : scores[i];
      const unsigned tie = MODE == 0 ? (unsigned)i : 0xFFFFFFFFu - (unsigned)i;
      k = ((unsigned long long)f2ord(sc) << 32) | tie;
    }
    keys[i] = k;
  }
  // batched_nms offset: idxs * (max coordinate + 1)   (torchvision batched_nms)
  float off_scale = 0.f;
  if (!CHUNK && MODE == 1 && idxs) {
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n * 4; i += SORT_THREADS) m = fmaxf(m, P[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x / WAVE] = m;
    __syncthreads();
    m = smax[0];
    for (int w = 1; w < SORT_THREADS / WAVE; ++w) m = fmaxf(m, smax[w]);
    off_scale = m + 1.0f;
  }
  __syncthreads();
  const bool radix = npad >= RADIX_MIN_N;
  if (radix) {
    // ---- LSD radix sort, 4-bit digits, 8 passes, one workgroup.  Position p starts with (key' = ~ordered(score), source index); MODE 1
    //      starts in index order, MODE 0 in reverse index order, so that the stable sort leaves equal scores in the order the reference's
    //      argsort does.  Every thread owns a contiguous run of `items` positions, counts its digits (packed 8-bit counters), the
    //      [digit][thread] count table is scanned once per pass, and the run is scattered in order.  ~25 us for 10 000 boxes against
    //      ~280 us for the bitonic network below, which is kept for small inputs.
    unsigned* rk = (unsigned*)keys;                                      // [RADIX_CAP] keys
    unsigned short* ri = (unsigned short*)(rk + RADIX_CAP);               // [RADIX_CAP] source indices
    unsigned short* cnt = ri + RADIX_CAP;                                 // [16][SORT_THREADS] digit counts, then exclusive offsets
    __shared__ unsigned s_wsum[SORT_THREADS / WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const int items = (n + SORT_THREADS - 1) / SORT_THREADS;              // <= 16
    const int total = items * SORT_THREADS;
    for (int p = tid; p < total; p += SORT_THREADS) {
      unsigned k = 0xFFFFFFFFu;
      unsigned short src = 0;
      if (p < n) {
        const int i = MODE == 0 ? n_total - 1 - (p_base + p) : p_base + p;
        const float sc = MODE == 0 ? P[(size_t)i * 6 + 4] : scores[i];
        k = ~f2ord(sc);
        src = (unsigned short)(CHUNK ? p : i);      // chunk form: position inside the chunk (the row index may exceed 16 bits)
      }
      rk[p] = k;
      ri[p] = src;
    }
    __syncthreads();
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 4 * pass;
      unsigned k[16];
      unsigned short ix[16];
      unsigned long long c0 = 0, c1 = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j < items) {
          k[j] = rk[tid * items + j];
          ix[j] = ri[tid * items + j];
          const unsigned d = (k[j] >> shift) & 15u;
          if (d < 8) c0 += 1ull << (8 * d);
          else c1 += 1ull << (8 * (d - 8));
        }
#pragma unroll
      for (int d = 0; d < 16; ++d) cnt[d * SORT_THREADS + tid] = (unsigned short)(((d < 8 ? c0 >> (8 * d) : c1 >> (8 * (d - 8)))) & 0xFFull);
      __syncthreads();
      // exclusive scan of the table in (digit, thread) order: thread t owns 16 consecutive entries
      unsigned v[16], sum = 0;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        v[e] = cnt[tid * 16 + e];
        sum += v[e];
      }
      unsigned inc = sum;
#pragma unroll
      for (int o = 1; o < WAVE; o <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)inc, o, WAVE);
        if (lane >= o) inc += up;
      }
      if (lane == WAVE - 1) s_wsum[wid] = inc;
      __syncthreads();
      unsigned run = inc - sum;
      for (int w = 0; w < wid; ++w) run += s_wsum[w];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        cnt[tid * 16 + e] = (unsigned short)run;
        run += v[e];
      }
      __syncthreads();
      // scatter the run in order; the thread's 16 offsets live in its own column of the table
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j < items) {
          const unsigned d = (k[j] >> shift) & 15u;
          const unsigned pos = cnt[d * SORT_THREADS + tid];
          cnt[d * SORT_THREADS + tid] = (unsigned short)(pos + 1);
          rk[pos] = k[j];
          ri[pos] = ix[j];
        }
      __syncthreads();
    }
  } else {
  for (int k = 2; k <= npad; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < npad; i += SORT_THREADS) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long a = keys[i], c = keys[ixj];
            const bool desc = (i & k) == 0;   // descending overall
            if (desc ? a < c : a > c) {
              keys[i] = c;
              keys[ixj] = a;
            }
          }
        }
        __syncthreads();
      }
    }
  }
  if (CHUNK) {
    const unsigned* rk = (const unsigned*)keys;
    const unsigned short* ri = (const unsigned short*)(rk + RADIX_CAP);
    unsigned* ckey = (unsigned*)(ws + L.ckey);
    int* csrc = (int*)(ws + L.csrc);
    for (int j = threadIdx.x; j < n; j += SORT_THREADS) {
      const int gp = p_base + (int)ri[j];
      ckey[p_base + j] = rk[j];
      csrc[p_base + j] = MODE == 0 ? n_total - 1 - gp : gp;
    }
    return;
  }
  int* sorted_idx = (int*)(ws + L.sorted_idx);
  float4* sbox = (float4*)(ws + L.sbox);
  int* scls = (int*)(ws + L.scls);
  const unsigned short* ri_sorted = (const unsigned short*)((const unsigned*)keys + RADIX_CAP);
  for (int i = threadIdx.x; i < n; i += SORT_THREADS) {
    const unsigned tie = (unsigned)(keys[i] & 0xFFFFFFFFull);
    const int src = radix ? (int)ri_sorted[i] : (MODE == 0 ? (int)tie : (int)(0xFFFFFFFFu - tie));
    sorted_idx[i] = src;
    if (MODE == 0) {
      const float* r = P + (size_t)src * 6;
      sbox[i] = make_float4(r[0], r[1], r[2], r[3]);
      scls[i] = (int)r[5];   // (P[:,5]).int()
    } else {
      float4 v = *(const float4*)(P + (size_t)src * 4);
      if (idxs) {
        const float o = (float)idxs[src] * off_scale;
        v.x += o; v.y += o; v.z += o; v.w += o;
      }
      sbox[i] = v;
    }
  }
  if (threadIdx.x == 0) ((int*)(ws + L.misc))[0] = n;
}

// Large inputs (n > NMS_SORT_CHUNK).  The stable LSD sort orders by (key, position in the initial order); keys of different chunks are
// merged by RANK: element j of chunk c lands at j + sum over the other chunks d of the number of their elements that precede it - those
// with key <= its key for d < c (earlier initial positions win ties), key < its key for d > c: two binary searches per other chunk, no
// serial merge.  Then the same gather as the single-workgroup sort (sorted boxes, classes, batched_nms category offsets).
__global__ __launch_bounds__(256) void nms_maxcoord_kernel(const float* __restrict__ boxes, const int* __restrict__ count, int n_fixed, int max_n,
                                                           char* __restrict__ ws_base, NmsWs L) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const int n = count ? min(count[b], max_n) : n_fixed;
  const float* P = boxes + (size_t)b * max_n * 4;
  float m = -INFINITY;
  for (long long i = threadIdx.x; i < (long long)n * 4; i += 256) m = fmaxf(m, P[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x / WAVE] = m;
  __syncthreads();
  if (threadIdx.x == 0) ((float*)(ws_base + (size_t)b * L.stride + L.misc))[1] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) + 1.0f;
}

template <int MODE>
__global__ __launch_bounds__(256) void nms_merge_kernel(const float* __restrict__ boxes, const long long* __restrict__ idxs, int max_n,
                                                        char* __restrict__ ws_base, NmsWs L) {
  const int b = blockIdx.y;
  char* ws = ws_base + (size_t)b * L.stride;
  const int n = ((const int*)(ws + L.misc))[0];
  const float off_scale = (MODE == 1 && idxs) ? ((const float*)(ws + L.misc))[1] : 0.f;
  const unsigned* ckey = (const unsigned*)(ws + L.ckey);
  const int* csrc = (const int*)(ws + L.csrc);
  int* sorted_idx = (int*)(ws + L.sorted_idx);
  float4* sbox = (float4*)(ws + L.sbox);
  int* scls = (int*)(ws + L.scls);
  const float* P = boxes + (size_t)b * max_n * (MODE == 0 ? 6 : 4);
  const int chunks = (n + NMS_SORT_CHUNK - 1) / NMS_SORT_CHUNK;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int c = e / NMS_SORT_CHUNK;
    const unsigned k = ckey[e];
    int rank = e - c * NMS_SORT_CHUNK;
    for (int d = 0; d < chunks; ++d) {
      if (d == c) continue;
      const unsigned* a = ckey + (size_t)d * NMS_SORT_CHUNK;
      int lo = 0, hi = min(NMS_SORT_CHUNK, n - d * NMS_SORT_CHUNK);
      while (lo < hi) {                                   // d < c: first element > k; d > c: first element >= k
        const int mid = (lo + hi) >> 1;
        const unsigned v = a[mid];
        if (d < c ? v <= k : v < k) lo = mid + 1;
        else hi = mid;
      }
      rank += lo;
    }
    const int src = csrc[e];
    sorted_idx[rank] = src;
    if (MODE == 0) {
      const float* r = P + (size_t)src * 6;
      sbox[rank] = make_float4(r[0], r[1], r[2], r[3]);
      scls[rank] = (int)r[5];
    } else {
      float4 v = *(const float4*)(P + (size_t)src * 4);
      if (idxs) {
        const float o = (float)idxs[(size_t)b * max_n + src] * off_scale;
        v.x += o; v.y += o; v.z += o; v.w += o;
      }
      sbox[rank] = v;
    }
  }
}

// IoU variants with the reference's exact operation order
template <int MODE>
__device__ __forceinline__ float nms_iou(const float4 s, const float4 r) {   // s = kept box (S), r = remaining
  const float xx1 = fmaxf(r.x, s.x), yy1 = fmaxf(r.y, s.y), xx2 = fminf(r.z, s.z), yy2 = fminf(r.w, s.w);
  const float w = fmaxf(xx2 - xx1, 0.0f), h = fmaxf(yy2 - yy1, 0.0f);
  const float inter = w * h;
  const float as = (s.z - s.x) * (s.w - s.y), ar = (r.z - r.x) * (r.w - r.y);
  if (MODE == 0) return inter / ((ar - inter) + as);   // helper.py:362-365
  return inter / (as + ar - inter);                    // torchvision nms
}

// ---- 2. bit mask: bit (i,j) j>i set when box j is dropped by box i; stored column-block major, word [tj][i] holds
//         columns tj*64..tj*64+63 of row i, so the scan's per-tile gather over rows is one contiguous, coalesced read
// MODE 0: drop = !(IoU < thr) (helper.py:368 keeps IoU<thr);  MODE 1: drop = IoU > thr
template <int MODE>
__global__ __launch_bounds__(WAVE) void nms_mask_kernel(char* __restrict__ ws_base, NmsWs L, float thr) {
  __shared__ float4 cb[WAVE];
  const int b = blockIdx.z;
  char* ws = ws_base + (size_t)b * L.stride;
  const int n = ((const int*)(ws + L.misc))[0];
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (tj < ti || ti * 64 >= n || tj * 64 >= n) return;
  const float4* sbox = (const float4*)(ws + L.sbox);
  const int lane = threadIdx.x;
  const int j = tj * 64 + lane;
  cb[lane] = j < n ? sbox[j] : make_float4(0, 0, 0, 0);
  __syncthreads();
  const int i = ti * 64 + lane;
  if (i >= n) return;
  const float4 s = sbox[i];
  unsigned long long bits = 0;
  const int cn = min(64, n - tj * 64);
  // The verdict is the reference's comparison on the IEEE quotient inter / den.  Away from the threshold (|inter - thr*den| beyond
  // a 2^-20 guard band, i.e. > 8 ulp of slack) the sign of inter - thr*den decides it without the division; inside the band and
  // for degenerate boxes (den <= 0, NaN) the exact quotient is formed.  Bit-identical keep masks, ~40 % fewer VALU ops.
  const float as = (s.z - s.x) * (s.w - s.y);
  for (int c = (ti == tj ? lane + 1 : 0); c < cn; ++c) {
    const float4 r = cb[c];
    const float xx1 = fmaxf(r.x, s.x), yy1 = fmaxf(r.y, s.y), xx2 = fminf(r.z, s.z), yy2 = fminf(r.w, s.w);
    const float w = fmaxf(xx2 - xx1, 0.0f), h = fmaxf(yy2 - yy1, 0.0f);
    const float inter = w * h;
    const float ar = (r.z - r.x) * (r.w - r.y);
    const float den = MODE == 0 ? (ar - inter) + as : as + ar - inter;
    const float t = thr * den;
    bool drop;
    if (den > 0.0f && thr > 0.0f && inter < t * (1.0f - 9.5367431640625e-7f)) drop = false;           // quotient < thr for sure
    else if (den > 0.0f && thr > 0.0f && inter > t * (1.0f + 9.5367431640625e-7f)) drop = true;       // quotient > thr for sure
    else {
      const float v = inter / den;
      drop = MODE == 0 ? !(v < thr) : (v > thr);
    }
    if (drop) bits |= 1ull << c;
  }
  ((unsigned long long*)(ws + L.mask))[(size_t)tj * (L.words * 64) + i] = bits;   // column-block major: [tj][row]
  if (ti == tj) {
    // transposed diagonal block for the scan's round-based resolve: lane c collects column c = the earlier boxes of this tile that
    // drop box c.  Built from the SAME bits by ballots (not recomputed: the quotient is not symmetric in its rounding)
    unsigned long long col = 0;
    for (int c = 0; c < cn; ++c) {
      const unsigned long long rows = __ballot((bits >> c) & 1ull);
      if (lane == c) col = rows;
    }
    ((unsigned long long*)(ws + L.diagt))[i] = col;
  }
}

// ---- 3. greedy scan: one workgroup per image ------------------------------------------------------
// Tile t (64 sorted boxes) needs R_t = OR over all boxes kept so far of their mask word for column block t, then a 64-step chain over
// its 64x64 diagonal block.  Only the chain is serial; everything else is pushed ahead of it:
//   * wave 0 ("resolver") resolves tile t from R[t] (LDS) | the rows of tile t-1 in column block t (prefetched one tile earlier, masked
//     with tile t-1's kept bits), and prefetches the two 64-word pieces of tile t+1 before it starts the chain;
//   * waves 1..15 ("pushers"), while tile t is being resolved, OR the kept rows of tile t-1 into R[u] of EVERY later column block
//     u >= t+1: one coalesced 512-byte read per column block (the 64 rows of a tile are contiguous in a column block), eight in flight,
//     and an LDS atomic OR per kept row.  By the time tile u is resolved R[u] holds the rows of all tiles <= u-2; tile u-1 comes in through the prefetch.
//   * one barrier per tile.  Sequential cost per 64 boxes: the readlane chain + one barrier.
#define SCAN_THREADS 1024
__device__ __forceinline__ unsigned op_or_u(unsigned a, unsigned b) { return a | b; }
// OR over the wave, result valid in lane 63 (quad / row rotations, then row broadcasts)
__device__ __forceinline__ unsigned long long wave_or64_lane63(unsigned long long v) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#define NMS_OR_STEP(x, ctrl, rmask) \
  x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rmask, 0xF, false)
  NMS_OR_STEP(lo, 0xB1, 0xF); NMS_OR_STEP(hi, 0xB1, 0xF);
  NMS_OR_STEP(lo, 0x4E, 0xF); NMS_OR_STEP(hi, 0x4E, 0xF);
  NMS_OR_STEP(lo, 0x124, 0xF); NMS_OR_STEP(hi, 0x124, 0xF);
  NMS_OR_STEP(lo, 0x128, 0xF); NMS_OR_STEP(hi, 0x128, 0xF);
  NMS_OR_STEP(lo, 0x142, 0xA); NMS_OR_STEP(hi, 0x142, 0xA);
  NMS_OR_STEP(lo, 0x143, 0xC); NMS_OR_STEP(hi, 0x143, 0xC);
#undef NMS_OR_STEP
  return ((unsigned long long)hi << 32) | lo;
}
__global__ __launch_bounds__(SCAN_THREADS) void nms_scan_kernel(char* __restrict__ ws_base, NmsWs L, int* __restrict__ out_count) {
  extern __shared__ unsigned long long s_scan[];        // [words] kept bits per tile, then [words] removed bits per column block
  const int b = blockIdx.x;
  char* ws = ws_base + (size_t)b * L.stride;
  const int n = ((const int*)(ws + L.misc))[0];
  const unsigned long long* mask = (const unsigned long long*)(ws + L.mask);
  const size_t NP = (size_t)L.words * 64;
  int* kept = (int*)(ws + L.kept);
  int* remover = (int*)(ws + L.remover);
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
  const int words = (n + 63) / 64;
  unsigned long long* skeptbits = s_scan;
  unsigned long long* sR = s_scan + L.words;
  for (int i = tid; i < words; i += SCAN_THREADS) sR[i] = 0ull;
  __syncthreads();
  constexpr int NPUSH = SCAN_THREADS / WAVE - 1;
  // resolver state (wave 0): rows of the previous tile in this column block, this tile's transposed diagonal block (lane j: the earlier
  // boxes of the tile that drop j), the previous tile's kept bits
  const unsigned long long* diagt = (const unsigned long long*)(ws + L.diagt);
  unsigned long long curA = 0, curT = 0, kept_prev = 0;
  if (wid == 0 && lane < n) curT = diagt[lane];
  int kc = 0;                                           // boxes kept in tiles < t (resolver only)
  for (int t = 0; t < words; ++t) {
    if (wid == 0) {
      const int row = t * 64 + lane;
      unsigned long long nextA = 0, nextT = 0;
      if (t + 1 < words) {
        if (row < n) nextA = mask[(size_t)(t + 1) * NP + row];
        if (row + 64 < n) nextT = diagt[row + 64];
      }
      const unsigned long long q = wave_or64_lane63(((kept_prev >> lane) & 1ull) ? curA : 0ull);
      const unsigned long long Rt = sR[t];
      const int cn = min(64, n - t * 64);
      const unsigned long long valid = cn == 64 ? ~0ull : ((1ull << cn) - 1ull);
      const unsigned long long Ru = (((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(q >> 32), 63) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)q, 63)) |
                                    (((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(Rt >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)Rt));
      // rounds instead of a 64-step chain: an undecided box that no undecided earlier box of the tile drops is kept (all its possible
      // droppers are decided, and a kept one would already have removed it); the newly kept ones then remove what they drop.  The lowest
      // undecided box always qualifies, so every round decides at least one; the number of rounds is the depth of the dependency chain
      // (a handful), not the number of kept boxes.  Same keep set as the sequential greedy scan.
      unsigned long long alive = ~Ru & valid;
      unsigned long long kept_bits = 0;
      while (alive) {
        const bool me = (alive >> lane) & 1ull;
        const unsigned long long K = __ballot(me && (curT & alive) == 0ull);
        kept_bits |= K;
        const unsigned long long gone = __ballot(me && (curT & K) != 0ull);
        alive &= ~(K | gone);
      }
      const bool mine = (kept_bits >> lane) & 1ull;
      if (row < n) remover[row] = mine ? -1 : -2;      // -2: removed, the remover is resolved by nms_remover_kernel
      if (mine) kept[kc + __popcll(kept_bits & ((1ull << lane) - 1ull))] = row;
      kc += __popcll(kept_bits);
      if (lane == 0) {
        skeptbits[t] = kept_bits;
        ((unsigned long long*)(ws + L.keptbits))[t] = kept_bits;
      }
      curA = nextA;
      curT = nextT;
      kept_prev = kept_bits;
    } else if (t >= 1) {
      const unsigned long long kb = skeptbits[t - 1];   // written before the last barrier
      if ((kb >> lane) & 1ull) {                        // only the kept rows of tile t-1 take part
        const unsigned long long* rows = mask + (size_t)(t - 1) * 64 + lane;      // + u * NP: this row's word in column block u
        for (int u = t + 1 + (wid - 1); u < words; u += 8 * NPUSH) {               // eight independent loads in flight, then LDS atomics
          unsigned long long w[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) w[i] = (u + i * NPUSH < words) ? rows[(size_t)(u + i * NPUSH) * NP] : 0ull;
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (w[i] != 0ull) atomicOr(&sR[u + i * NPUSH], w[i]);
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    ((int*)(ws + L.misc))[1] = kc;
    out_count[b] = kc;
  }
}

// remover[c] = the first kept box (in keep order = ascending sorted row) whose mask row has bit c: the box that suppressed c
// (helper.py:355-368).  One workgroup per column block of 64 boxes; its 16 waves split the row tiles v <= w round-robin.  Per row tile a
// wave reads the 64 row words of the column block in one coalesced load and walks the tile's KEPT rows in ascending order on the scalar
// unit (readlane broadcast of the row's word, lane c tests its own bit); each lane keeps its first hit, an LDS min over the waves gives
// the first kept row overall.  Work: (row tiles) x (kept rows per tile) scalar steps per wave instead of a walk over the whole kept
// list per 64 boxes (1.16 ms -> ~20 us at n = 10 000).
#define REMOVER_WAVES 16
__global__ __launch_bounds__(REMOVER_WAVES* WAVE) void nms_remover_kernel(char* __restrict__ ws_base, NmsWs L) {
  __shared__ int s_best[REMOVER_WAVES][WAVE];
  const int b = blockIdx.y;
  char* ws = ws_base + (size_t)b * L.stride;
  const int n = ((const int*)(ws + L.misc))[0];
  const int words = (n + 63) / 64;
  const int w = blockIdx.x;
  if (w >= words) return;
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const size_t NP = (size_t)L.words * 64;
  const unsigned long long* col = (const unsigned long long*)(ws + L.mask) + (size_t)w * NP;
  const unsigned long long* keptbits = (const unsigned long long*)(ws + L.keptbits);
  int* remover = (int*)(ws + L.remover);
  const int c = w * 64 + lane;
  const bool removed = c < n && remover[c] == -2;
  int best = 0x7FFFFFFF;
  for (int v = wid; v <= w; v += REMOVER_WAVES) {
    if (!__any(removed && best == 0x7FFFFFFF)) break;
    unsigned long long kb = keptbits[v];
    if (kb == 0ull) continue;
    const int row = v * 64 + lane;
    const unsigned long long mine = row < n ? col[row] : 0ull;
    const unsigned mlo = (unsigned)mine, mhi = (unsigned)(mine >> 32);
    while (kb) {
      const int i = __builtin_ctzll(kb);
      kb &= kb - 1ull;
      const unsigned long long word = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mhi, i) << 32) | (unsigned)__builtin_amdgcn_readlane((int)mlo, i);
      if (((word >> lane) & 1ull) && best == 0x7FFFFFFF) best = v * 64 + i;
    }
  }
  s_best[wid][lane] = best;
  __syncthreads();
  if (wid == 0 && removed) {
    int m = s_best[0][lane];
#pragma unroll
    for (int q = 1; q < REMOVER_WAVES; ++q) m = min(m, s_best[q][lane]);
    remover[c] = m;
  }
}

// ---- 4. majority vote + output (helper.py:369-375) --------------------------------------------
// One wave per kept box: votes = boxes removed BY this box with IoU > thr; if they span more than
// one class, relabel to the modal class (count ties -> smallest class id).
#define VOTE_WAVES 4
__global__ __launch_bounds__(VOTE_WAVES* WAVE) void nms_vote_kernel(const float* __restrict__ boxes, char* __restrict__ ws_base, NmsWs L,
                                                                     int max_n, float thr, int num_classes,
                                                                     float* __restrict__ out_rows, int* __restrict__ out_idx) {
  extern __shared__ int hist_all[];
  const int b = blockIdx.y;
  char* ws = ws_base + (size_t)b * L.stride;
  const int n = ((const int*)(ws + L.misc))[0];
  const int kc = ((const int*)(ws + L.misc))[1];
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  int* hist = hist_all + (size_t)wid * num_classes;
  const float4* sbox = (const float4*)(ws + L.sbox);
  const int* scls = (const int*)(ws + L.scls);
  const int* kept = (const int*)(ws + L.kept);
  const int* remover = (const int*)(ws + L.remover);
  const int* sorted_idx = (const int*)(ws + L.sorted_idx);
  for (int c = lane; c < num_classes; c += WAVE) hist[c] = 0;
  for (int k = blockIdx.x * VOTE_WAVES + wid; k < kc; k += gridDim.x * VOTE_WAVES) {
    const int pos = kept[k];
    const float4 s = sbox[pos];
    int nvote = 0;
    // the boxes this one removed are among the set bits of its mask row: lanes run over the column blocks, each walks its word's bits
    // (was: a scan of every later box for remover == pos)
    const unsigned long long* mrow = (const unsigned long long*)(ws + L.mask) + pos;
    const size_t NPm = (size_t)L.words * 64;
    for (int u = (pos >> 6) + lane; u < (n + 63) / 64; u += WAVE) {
      unsigned long long word = mrow[(size_t)u * NPm];
      while (word) {
        const int j = u * 64 + __builtin_ctzll(word);
        word &= word - 1ull;
        if (j < n && remover[j] == pos) {
          const float v = nms_iou<0>(s, sbox[j]);
          if (v > thr) {
            const int c = scls[j];
            if (c >= 0 && c < num_classes) atomicAdd(&hist[c], 1);
            ++nvote;
          }
        }
      }
    }
    nvote = (int)wave_sum((float)nvote);
    __threadfence_block();
    const int src = sorted_idx[pos];
    const float* r = boxes + ((size_t)b * max_n + src) * 6;
    float label = r[5];
    if (nvote > 0) {
      int best_cnt = 0, best_cls = 0x7FFFFFFF, distinct = 0;
      for (int c = lane; c < num_classes; c += WAVE) {
        const int h = hist[c];
        if (h > 0) {
          ++distinct;
          if (h > best_cnt) {   // ascending c within a lane: first maximum = smallest id
            best_cnt = h;
            best_cls = c;
          }
          hist[c] = 0;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int oc = __shfl_xor(best_cnt, o, WAVE), ok = __shfl_xor(best_cls, o, WAVE);
        distinct += __shfl_xor(distinct, o, WAVE);
        if (oc > best_cnt || (oc == best_cnt && ok < best_cls)) {
          best_cnt = oc;
          best_cls = ok;
        }
      }
      if (distinct > 1) label = (float)best_cls;
    }
    __threadfence_block();
    if (lane < 6) out_rows[((size_t)b * max_n + k) * 6 + lane] = lane == 5 ? label : r[lane];
    if (lane == 0) out_idx[(size_t)b * max_n + k] = src;
  }
}

__global__ void nms_keep_kernel(char* __restrict__ ws_base, NmsWs L, long long* __restrict__ keep, int* __restrict__ keep_count, int max_n) {
  const int b = blockIdx.y;                                   // batched form: image b writes keep[b*max_n ...], keep_count[b]
  char* ws = ws_base + (size_t)b * L.stride;
  const int kc = ((const int*)(ws + L.misc))[1];
  const int* kept = (const int*)(ws + L.kept);
  const int* sorted_idx = (const int*)(ws + L.sorted_idx);
  keep += (size_t)b * max_n;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < kc; k += gridDim.x * blockDim.x) keep[k] = sorted_idx[kept[k]];
  if (blockIdx.x == 0 && threadIdx.x == 0) keep_count[b] = kc;
}

// ---- torchvision box_iou ----------------------------------------------------------------------
__device__ __forceinline__ float tv_iou(const float4 a, const float4 b) {
  const float area_a = (a.z - a.x) * (a.w - a.y), area_b = (b.z - b.x) * (b.w - b.y);
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.0f), h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.0f);
  const float inter = w * h;
  return inter / (area_a + area_b - inter);
}

__global__ void box_iou_kernel(const float* __restrict__ b1, const float* __restrict__ b2, float* __restrict__ out, long long m, long long n) {
  const long long total = m * n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / n, c = i - r * n;
    out[i] = tv_iou(*(const float4*)(b1 + 4 * r), *(const float4*)(b2 + 4 * c));
  }
}

// ---- fused box_iou + Matcher (tvision/_utils.py:271-344) ----------------------------------------
// pass 1: per anchor max over GT (first index on ties) + thresholds; per-GT row max via atomicMax
// pass 2: low-quality rescue — anchors attaining any GT's row maximum get their argmax back.
// Algorithmic bytes: N*16 (anchors) + M*16 read, N*8 written (x2 anchors read when rescue is on).
#define MATCH_THREADS 256
#define MATCH_MAX_GT 1024
__global__ __launch_bounds__(MATCH_THREADS) void match_pass1_kernel(const float* __restrict__ gt, const float* __restrict__ anchors, int m,
                                                                     long long n, float hi, float lo, unsigned* __restrict__ gt_best,
                                                                     long long* __restrict__ matches) {
  extern __shared__ float4 sg[];
  // per-GT maximum over this workgroup's anchors with LDS atomics, then ONE global atomic per GT per workgroup
  // (a global atomicMax per wave per GT = 13k same-address atomics cost 170 us at N=120k, M=7)
  unsigned* sbest = (unsigned*)(sg + m);
  for (int g = threadIdx.x; g < m; g += MATCH_THREADS) {
    sg[g] = *(const float4*)(gt + 4 * (size_t)g);
    sbest[g] = 0u;
  }
  __syncthreads();
  const long long stride = (long long)gridDim.x * MATCH_THREADS;
  const long long n_round = (n + MATCH_THREADS - 1) / MATCH_THREADS * MATCH_THREADS;
  for (long long i = blockIdx.x * (long long)MATCH_THREADS + threadIdx.x; i < n_round; i += stride) {
    const bool live = i < n;
    const float4 a = live ? *(const float4*)(anchors + 4 * i) : make_float4(0, 0, 0, 0);
    float best = -INFINITY;
    int arg = 0;
    for (int g = 0; g < m; ++g) {
      const float v = tv_iou(sg[g], a);
      if (g == 0 || v > best) {   // first maximum, as torch.max(dim=0)
        best = v;
        arg = g;
      }
      unsigned o = live ? f2ord(v) : 0u;
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) o = max(o, (unsigned)__shfl_xor((int)o, s, WAVE));
      if ((threadIdx.x & 63) == 0) atomicMax(sbest + g, o);
    }
    if (live) {
      long long r = arg;
      if (best < lo) r = -1;
      else if (best < hi) r = -2;
      matches[i] = r;
    }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < m; g += MATCH_THREADS) atomicMax(gt_best + g, sbest[g]);
}

__global__ __launch_bounds__(MATCH_THREADS) void match_pass2_kernel(const float* __restrict__ gt, const float* __restrict__ anchors, int m,
                                                                     long long n, const unsigned* __restrict__ gt_best,
                                                                     long long* __restrict__ matches) {
  extern __shared__ float4 sg[];
  unsigned* sb = (unsigned*)(sg + m);
  for (int g = threadIdx.x; g < m; g += MATCH_THREADS) {
    sg[g] = *(const float4*)(gt + 4 * (size_t)g);
    sb[g] = gt_best[g];
  }
  __syncthreads();
  const long long i = blockIdx.x * (long long)MATCH_THREADS + threadIdx.x;
  if (i >= n) return;
  const float4 a = *(const float4*)(anchors + 4 * i);
  float best = -INFINITY;
  int arg = 0;
  bool rescue = false;
  for (int g = 0; g < m; ++g) {
    const float v = tv_iou(sg[g], a);
    if (g == 0 || v > best) {
      best = v;
      arg = g;
    }
    rescue = rescue || (v == ord2f(sb[g]));
  }
  if (rescue) matches[i] = arg;
}

// ---- BoxCoder (tvision/_utils.py:79-125, 190-223) -------------------------------------------------
__global__ void box_encode_kernel(const float* __restrict__ ref, const float* __restrict__ prop, float* __restrict__ out, long long n, float wx,
                                  float wy, float ww, float wh) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float4 p = *(const float4*)(prop + 4 * i), r = *(const float4*)(ref + 4 * i);
    const float ew = p.z - p.x, eh = p.w - p.y, ecx = p.x + 0.5f * ew, ecy = p.y + 0.5f * eh;
    const float gw = r.z - r.x, gh = r.w - r.y, gcx = r.x + 0.5f * gw, gcy = r.y + 0.5f * gh;
    *(float4*)(out + 4 * i) = make_float4(wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh, ww * logf(gw / ew), wh * logf(gh / eh));
  }
}

__global__ void box_decode_kernel(const float* __restrict__ codes, const float* __restrict__ boxes, float* __restrict__ out, long long n, int k,
                                  float wx, float wy, float ww, float wh, float clip) {
  const long long total = n * k;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / k;
    const float4 b = *(const float4*)(boxes + 4 * r), c = *(const float4*)(codes + 4 * i);
    const float w = b.z - b.x, h = b.w - b.y, cx = b.x + 0.5f * w, cy = b.y + 0.5f * h;
    const float dx = c.x / wx, dy = c.y / wy, dw = fminf(c.z / ww, clip), dh = fminf(c.w / wh, clip);
    const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
    *(float4*)(out + 4 * i) = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
  }
}

// ---- AnchorGenerator.grid_anchors, one level (anchor_utils.py:98-134) ----------------------------
__global__ void anchor_grid_kernel(const float* __restrict__ cell, int a, int gh, int gw, int sh, int sw, float* __restrict__ out) {
  const long long total = (long long)gh * gw * a;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ai = (int)(i % a);
    const long long p = i / a;
    const int x = (int)(p % gw), y = (int)(p / gw);
    const float fx = (float)x * (float)sw, fy = (float)y * (float)sh;
    const float4 c = *(const float4*)(cell + 4 * ai);
    *(float4*)(out + 4 * i) = make_float4(fx + c.x, fy + c.y, fx + c.z, fy + c.w);
  }
}

// ---- sigmoid focal loss fused fwd+bwd (torchvision.ops.sigmoid_focal_loss; retinanet.py:137-141) ---
// The kernel is VALU bound (not HBM bound) on the transcendental chain: one hardware exp, one hardware log and ONE reciprocal
// per element (p and 1-p share it); the IEEE division and log1pf of the first version cost ~100 lane-ops per element
// (0.84 ms for 16 x 120087 x 91 logits).  Absolute error of log(1+e) vs log1p(e) is < 6e-8, far inside the 1e-4 loss bar.
__device__ __forceinline__ void sfl(float x, float t, float alpha, float gamma, float& loss, float& grad) {
  const float e = __expf(-fabsf(x));
  const float inv = __builtin_amdgcn_rcpf(1.0f + e);
  const float ce = fmaxf(x, 0.0f) - x * t + __logf(1.0f + e);
  const float p = x >= 0.0f ? inv : e * inv;
  const float p_t = p * t + (1.0f - p) * (1.0f - t);
  const float q = 1.0f - p_t;
  const float mf = gamma == 2.0f ? q * q : powf(q, gamma);
  const float dmf = gamma == 2.0f ? 2.0f * q : (q > 0.0f ? gamma * powf(q, gamma - 1.0f) : 0.0f);
  const float dpt = (2.0f * t - 1.0f) * p * (1.0f - p);
  loss = ce * mf;
  grad = (p - t) * mf - ce * dmf * dpt;
  if (alpha >= 0.0f) {
    const float a_t = alpha * t + (1.0f - alpha) * (1.0f - t);
    loss *= a_t;
    grad *= a_t;
  }
}

// TGT_MODE 0: dense target tensor t; 1: target from Matcher output (matched>=0 -> class gt_labels[matched])
#define FOCAL_THREADS 1024     // 16 waves fold their sums through LDS: ONE atomic per workgroup on the loss word, <= 512 workgroups (atomics to
                               // one address retire at ~90 per microsecond: 2048 of them were 23 of the kernel's 56 us at K = 91)
// optional bf16 gradient output in the layout the head convolution's backward reads: per pyramid level an NHWC buffer [n, h*w, ld] whose
// channel a*k + c is anchor a, class c of that pixel (row r of the level-concatenated [n, sum HWA, k] logits = pixel r / A, anchor r % A)
struct FocalDiv {             // n / d for n < 2^31 (Granlund-Montgomery round-up form, as conv_kernels.hip)
  unsigned mul, shift, d;
};
static FocalDiv focal_div(unsigned d) {
  FocalDiv f;
  f.d = d;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.shift = l;
  f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ unsigned fdivu(unsigned n, const FocalDiv f) { return (__umulhi(f.mul, n) + n) >> f.shift; }

struct FocalLevels {
  FocalDiv dk, drpi, da;      // divisions by k, rows_per_image, A without the ~25-instruction hardware sequence (fast path below)
  int fast;                   // 1: rows and rows*k < 2^31, k % 4 == 0: the four elements of a group share a row
  int nlev, A;
  long long start[8];       // first row of level q inside one image's rows
  long long pixels[8];      // h*w of level q
  bf16_t* dst[8];
  int ld[8];
};

template <int TGT_MODE, int NT = FOCAL_THREADS>
__global__ __launch_bounds__(NT) void focal_kernel(const float* __restrict__ x, const float* __restrict__ t, const long long* __restrict__ matched,
                                                    const long long* __restrict__ gt_labels, const float* __restrict__ scale,
                                                    const unsigned char* __restrict__ valid, long long rows, int k, float alpha, float gamma,
                                                    float gscale, float* __restrict__ loss_sum, float* __restrict__ grad,
                                                    const float* __restrict__ nfg = nullptr, long long rows_per_image = 0,
                                                    const int* __restrict__ gt_off = nullptr, float inv_images = 1.f, const FocalLevels lv = FocalLevels{}) {
  __shared__ float red[NT / WAVE];
  const long long total = rows * k;
  const bool vec = (total & 3) == 0;      // 4 consecutive elements per lane: 16-byte loads/stores of logits and gradients
  float acc = 0.f;
  const long long nvec = vec ? total / 4 : total;
  for (long long q = blockIdx.x * (long long)NT + threadIdx.x; q < nvec; q += (long long)gridDim.x * NT) {
    const long long i0 = vec ? q * 4 : q;
    const bool level_out = TGT_MODE == 1 && lv.nlev > 0 && lv.fast && vec && !grad;      // (wave-uniform) bf16 level output, 32-bit index arithmetic
    float g4[4] = {0.f, 0.f, 0.f, 0.f};
    bf16_t* da[4] = {nullptr, nullptr, nullptr, nullptr};                                  // destination of each of the four gradients
    if (level_out) {
      // ---- training fast path (mi355det_retina_loss_lv): 32-bit index arithmetic with multiply-shift divisions, the t = 0 form of the loss
      //      for groups that do not contain a label (all but one group in 301 at K = 1204); the general path below spends ~100 instructions
      //      per element on 64-bit divisions and per-element bookkeeping.  k % 4 != 0 (the 91-class head): a group may straddle two rows
      //      (13 % of the groups at k = 91); those lanes fetch the state of the second row too, and ALL lanes run the same four loss
      //      evaluations - as a separate branch the straddling groups made every wave execute both branches (1.05 ms for 175 M logits
      //      against 2.2 ms for the 1.16 G of the 1204-class head).
      const unsigned iu = (unsigned)i0;
      const unsigned r0 = fdivu(iu, lv.dk), c0 = iu - r0 * (unsigned)k;
      const bool straddles = lv.fast == 2 && c0 + 3u >= (unsigned)k;
      const float4 v = *(const float4*)(x + i0);
      const float xs[4] = {v.x, v.y, v.z, v.w};
      bf16_t* drow[2];
      long long mi2[2], lab2[2];
      float wimg2[2];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half == 1 && !straddles) {
          drow[1] = drow[0]; mi2[1] = mi2[0]; lab2[1] = lab2[0]; wimg2[1] = wimg2[0];
          break;
        }
        const unsigned ru = r0 + half;
        const unsigned bu = fdivu(ru, lv.drpi), rl = ru - bu * lv.drpi.d;
        unsigned st0 = 0, px = (unsigned)lv.pixels[0];
        bf16_t* base = lv.dst[0];
        int ldq = lv.ld[0];
#pragma unroll
        for (int l = 1; l < 8; ++l)
          if (l < lv.nlev && rl >= (unsigned)lv.start[l]) {
            st0 = (unsigned)lv.start[l]; px = (unsigned)lv.pixels[l]; base = lv.dst[l]; ldq = lv.ld[l];
          }
        const unsigned local = rl - st0, pix = fdivu(local, lv.da);
        drow[half] = base + ((size_t)bu * px + pix) * (size_t)ldq + (size_t)(local - pix * lv.da.d) * k;
        mi2[half] = matched[ru];
        wimg2[half] = nfg ? inv_images / fmaxf(1.f, nfg[bu]) : 1.f;
        lab2[half] = mi2[half] >= 0 ? gt_labels[mi2[half] + (gt_off ? gt_off[bu] : 0)] : -1;
      }
      // a label inside the group's column range of either row (BETWEEN_THRESHOLDS rows, mi == -2, have lab == -1)
      const bool has_label = (lab2[0] >= (long long)c0 && lab2[0] < (long long)c0 + 4) ||
                             (straddles && lab2[1] >= 0 && lab2[1] + (long long)k < (long long)c0 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool second = c0 + e >= (unsigned)k;             // (never for groups inside one row)
        const unsigned cu = second ? c0 + e - (unsigned)k : c0 + e;
        const long long mi = second ? mi2[1] : mi2[0], lab = second ? lab2[1] : lab2[0];
        const float wimg = second ? wimg2[1] : wimg2[0];
        da[e] = (second ? drow[1] : drow[0]) + cu;
        if (mi == -2) continue;                                // BETWEEN_THRESHOLDS rows are ignored (retinanet.py:135): gradient 0
        const float sc = scale ? scale[cu] : 1.0f;
        float l, g;
        if (has_label || alpha < 0.0f || gamma != 2.0f) {
          sfl(sc * xs[e], lab == (long long)cu ? 1.0f : 0.0f, alpha, gamma, l, g);
        } else {
          // t = 0, gamma = 2: loss = (1 - alpha) p^2 softplus(x); d/dx = (1 - alpha) p^2 (2 (1 - p) softplus(x) + p).  Same operations as
          // sfl() would execute with t = 0, in the same order: bit-identical.
          const float xx = sc * xs[e];
          const float ee = __expf(-fabsf(xx));
          const float inv = __builtin_amdgcn_rcpf(1.0f + ee);
          const float ce = fmaxf(xx, 0.0f) - xx * 0.0f + __logf(1.0f + ee);
          const float p = xx >= 0.0f ? inv : ee * inv;
          const float p_t = p * 0.0f + (1.0f - p) * (1.0f - 0.0f);
          const float q = 1.0f - p_t;
          const float mf = q * q, dmf = 2.0f * q;
          const float dpt = (2.0f * 0.0f - 1.0f) * p * (1.0f - p);
          const float a_t = alpha * 0.0f + (1.0f - alpha) * (1.0f - 0.0f);
          l = ce * mf * a_t;
          g = ((p - 0.0f) * mf - ce * dmf * dpt) * a_t;
        }
        acc += l * wimg;
        g4[e] = g * (sc * gscale * wimg);
      }
    }
    if (level_out) {
      // ---- stores.  k % 4 == 0: every group is an aligned 8-byte word.  Otherwise (k = 91) three groups in four start at an odd 2- or
      //      4-byte boundary: when the wave's 256 gradients form ONE contiguous run (no pixel / level / image boundary inside: the pitch of a
      //      pixel is padded), lane L writes the aligned word that starts inside its group - its own tail and the head of lane L+1's group -
      //      and only the first and the last lane add 2-byte stores; 2-byte stores for every element cost 3x the kernel's arithmetic.
      const unsigned long long v64 = (unsigned long long)f2bf(g4[0]) | ((unsigned long long)f2bf(g4[1]) << 16) |
                                     ((unsigned long long)f2bf(g4[2]) << 32) | ((unsigned long long)f2bf(g4[3]) << 48);
      const bool own_run = da[1] == da[0] + 1 && da[2] == da[0] + 2 && da[3] == da[0] + 3;
      if (lv.fast == 1) {
        *(unsigned long long*)da[0] = v64;
      } else {
        const int lane = threadIdx.x & (WAVE - 1);
        const unsigned long long a0 = (unsigned long long)da[0];
        const unsigned long long b0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a0 >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)a0);
        const bool full = __ballot(1) == ~0ull;
        const bool run = full && __all(own_run && a0 == b0 + 8ull * (unsigned long long)lane);
        if (run) {
          const int s = (int)((8 - (b0 & 7)) & 7) >> 1;          // first element of the run that starts an aligned word (0..3)
          if (s == 0) {
            *(unsigned long long*)da[0] = v64;
          } else {
            const unsigned nlo = (unsigned)__shfl_down((int)(unsigned)v64, 1, WAVE), nhi = (unsigned)__shfl_down((int)(unsigned)(v64 >> 32), 1, WAVE);
            const unsigned long long nxt = ((unsigned long long)nhi << 32) | nlo;
            if (lane < WAVE - 1) *(unsigned long long*)(da[0] + s) = (v64 >> (16 * s)) | (nxt << (16 * (4 - s)));
            if (lane == 0)
              for (int e = 0; e < s; ++e) da[e][0] = f2bf(g4[e]);
            if (lane == WAVE - 1)
              for (int e = s; e < 4; ++e) da[e][0] = f2bf(g4[e]);
          }
        } else if (own_run && (a0 & 7ull) == 0) {
          *(unsigned long long*)da[0] = v64;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) da[e][0] = f2bf(g4[e]);
        }
      }
      continue;
    }
    long long r = total < (1ll << 31) ? (long long)((unsigned)i0 / (unsigned)k) : i0 / k;
    int c = (int)(i0 - r * k);
    float xv[4], tv4[4], gv[4];
    const int cnt = vec ? 4 : 1;
    if (vec) {
      const float4 v = *(const float4*)(x + i0);
      xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w;
      if (TGT_MODE == 0) {
        const float4 w = *(const float4*)(t + i0);
        tv4[0] = w.x; tv4[1] = w.y; tv4[2] = w.z; tv4[3] = w.w;
      }
    } else {
      xv[0] = x[i0];
      if (TGT_MODE == 0) tv4[0] = t[i0];
    }
    long long mi = 0, lab = -1;
    bool ok = true;
    bool fresh = true;
    float wimg = 1.f;          // batched form: 1 / max(1, num_foreground of the row's image) / num_images (retinanet.py:141-143)
    // bf16 level output: `drow` = channel 0 of the current row's (pixel, anchor); a group of 4 that stays inside one row and is 8-byte
    // aligned (always for k % 4 == 0) is written with one store
    auto locate = [&](long long row) -> bf16_t* {
      // 32-bit divisions where the row index allows (always, in practice): the 64-bit forms cost ~100 instructions each, three per group
      // of four elements - the 1204-class loss ran at 2.2 TB/s
      long long b, rl;
      if (rows < (1ll << 31)) {
        const unsigned bu = (unsigned)row / (unsigned)rows_per_image;
        b = bu;
        rl = (unsigned)row - bu * (unsigned)rows_per_image;
      } else {
        b = row / rows_per_image;
        rl = row - b * rows_per_image;
      }
      long long st0 = 0, px = lv.pixels[0];
      bf16_t* base = lv.dst[0];
      int ldq = lv.ld[0];
#pragma unroll
      for (int l = 1; l < 8; ++l)
        if (l < lv.nlev && rl >= lv.start[l]) {
          st0 = lv.start[l]; px = lv.pixels[l]; base = lv.dst[l]; ldq = lv.ld[l];
        }
      const long long local = rl - st0;
      const long long pix = rows < (1ll << 31) ? (long long)((unsigned)local / (unsigned)lv.A) : local / lv.A;
      return base + (b * px + pix) * ldq + (local - pix * lv.A) * k;
    };
    bf16_t* drow = lv.nlev > 0 ? locate(r) : nullptr;
    bf16_t* const dfirst = drow ? drow + c : nullptr;
    const bool packed = drow && vec && c + 3 < k && (((unsigned long long)dfirst) & 7ull) == 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e >= cnt) break;
      if (fresh && e > 0 && lv.nlev > 0) drow = locate(r);
      if (fresh) {
        if (TGT_MODE == 0) ok = !valid || valid[r];
        else {
          mi = matched[r];
          ok = mi != -2;                                   // BETWEEN_THRESHOLDS rows are ignored (retinanet.py:135)
          int b = 0;
          if (nfg) {
            b = rows < (1ll << 31) ? (int)((unsigned)r / (unsigned)rows_per_image) : (int)(r / rows_per_image);
            wimg = inv_images / fmaxf(1.f, nfg[b]);
          }
          lab = mi >= 0 ? gt_labels[mi + (gt_off ? gt_off[b] : 0)] : -1;
        }
        fresh = false;
      }
      const float tv = TGT_MODE == 0 ? tv4[e] : (lab == c ? 1.0f : 0.0f);
      float g = 0.f;
      if (ok) {
        const float sc = scale ? scale[c] : 1.0f;
        float l;
        sfl(sc * xv[e], tv, alpha, gamma, l, g);
        acc += l * wimg;
        g *= sc * gscale * wimg;
      }
      gv[e] = g;
      if (lv.nlev > 0 && !packed) drow[c] = f2bf(g);
      if (++c == k) {
        c = 0;
        ++r;
        fresh = true;
      }
    }
    if (grad) {
      if (vec) *(float4*)(grad + i0) = make_float4(gv[0], gv[1], gv[2], gv[3]);
      else grad[i0] = gv[0];
    }
    if (packed) {
      uint2 o;
      o.x = f2bf(gv[0]) | ((unsigned)f2bf(gv[1]) << 16);
      o.y = f2bf(gv[2]) | ((unsigned)f2bf(gv[3]) << 16);
      *(uint2*)dfirst = o;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x / WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NT / WAVE; ++w) t += red[w];
    atomicAdd(loss_sum, t);
  }
}

// ---- RetinaNet head losses for a whole batch (retinanet.py:56-62,107-143,196-223) ---------------------------------
// nfg[b] = number of foreground anchors (matched >= 0) of image b
__global__ __launch_bounds__(256) void count_fg_kernel(const long long* __restrict__ matched, long long rows_per_image, float* __restrict__ nfg) {
  __shared__ float red[4];
  const long long* m = matched + (long long)blockIdx.y * rows_per_image;
  float c = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < rows_per_image; i += (long long)gridDim.x * 256) c += m[i] >= 0 ? 1.f : 0.f;
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x / WAVE] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(nfg + blockIdx.y, red[0] + red[1] + red[2] + red[3]);   // integer-valued floats: order independent
}

// L1 regression loss on the foreground anchors against BoxCoder.encode_single targets (weights 1,1,1,1; _utils.py:130-163)
__global__ __launch_bounds__(256) void retina_reg_kernel(const float* __restrict__ pred, const float* __restrict__ anchors,
                                                         const long long* __restrict__ matched, const float* __restrict__ gt_boxes,
                                                         const int* __restrict__ gt_off, const float* __restrict__ nfg, int n_images,
                                                         long long rows_per_image, float wx, float wy, float ww, float wh, float gscale,
                                                         float* __restrict__ loss_sum, float* __restrict__ grad) {
  __shared__ float red[4];
  const long long total = (long long)n_images * rows_per_image;
  float acc = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long mi = matched[i];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mi >= 0) {
      const int b = (int)(i / rows_per_image);
      const long long r = i - (long long)b * rows_per_image;
      const float wimg = 1.f / fmaxf(1.f, nfg[b]) / (float)n_images;
      const float4 a = *(const float4*)(anchors + r * 4);
      const float4 q = *(const float4*)(gt_boxes + (mi + gt_off[b]) * 4);
      const float4 pv = *(const float4*)(pred + i * 4);
      const float ew = a.z - a.x, eh = a.w - a.y, ecx = a.x + 0.5f * ew, ecy = a.y + 0.5f * eh;
      const float gw = q.z - q.x, gh = q.w - q.y, gcx = q.x + 0.5f * gw, gcy = q.y + 0.5f * gh;
      const float t0 = wx * (gcx - ecx) / ew, t1 = wy * (gcy - ecy) / eh, t2 = ww * logf(gw / ew), t3 = wh * logf(gh / eh);
      const float d0 = pv.x - t0, d1 = pv.y - t1, d2 = pv.z - t2, d3 = pv.w - t3;
      acc += (fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3)) * wimg;
      const float gs = gscale * wimg;
      g = make_float4(d0 > 0.f ? gs : (d0 < 0.f ? -gs : 0.f), d1 > 0.f ? gs : (d1 < 0.f ? -gs : 0.f), d2 > 0.f ? gs : (d2 < 0.f ? -gs : 0.f),
                      d3 > 0.f ? gs : (d3 < 0.f ? -gs : 0.f));
    }
    if (grad) *(float4*)(grad + i * 4) = g;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x / WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
}

int launch_nms_common(int mode, const float* boxes, const float* scores, const long long* idxs, const int* count, int n_fixed, int bs,
                      int max_n, float thr, void* workspace, size_t workspace_bytes, int* out_count, hipStream_t st, NmsWs& L) {
  if (max_n <= 0 || max_n > NMS_MAX_N) return fail(MI355DET_EINVAL, "%s: n must be in [1,%lld]", "nms", NMS_MAX_N);
  L = nms_layout(max_n);
  if (workspace_bytes < L.stride * (size_t)bs) return fail(MI355DET_EWORKSPACE, "%s: workspace too small (%lld needed)", "nms", (long long)(L.stride * bs));
  if (max_n > NMS_SORT_CHUNK) {
    // chunk sorts (one workgroup per 16 384 boxes) + merge by rank
    const size_t lds = (size_t)RADIX_CAP * 6 + (size_t)16 * SORT_THREADS * 2;
    const int chunks = (max_n + NMS_SORT_CHUNK - 1) / NMS_SORT_CHUNK;
    if (mode == 0) {
      (void)hipFuncSetAttribute((const void*)nms_sort_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((nms_sort_kernel<0, true>), dim3(chunks, bs), dim3(SORT_THREADS), lds, st, boxes, scores, idxs, count, n_fixed, max_n, (char*)workspace, L);
      hipLaunchKernelGGL(nms_merge_kernel<0>, dim3(min(256, (max_n + 255) / 256), bs), dim3(256), 0, st, boxes, idxs, max_n, (char*)workspace, L);
    } else {
      (void)hipFuncSetAttribute((const void*)nms_sort_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((nms_sort_kernel<1, true>), dim3(chunks, bs), dim3(SORT_THREADS), lds, st, boxes, scores, idxs, count, n_fixed, max_n, (char*)workspace, L);
      if (idxs) hipLaunchKernelGGL(nms_maxcoord_kernel, dim3(bs), dim3(256), 0, st, boxes, count, n_fixed, max_n, (char*)workspace, L);
      hipLaunchKernelGGL(nms_merge_kernel<1>, dim3(min(256, (max_n + 255) / 256), bs), dim3(256), 0, st, boxes, idxs, max_n, (char*)workspace, L);
    }
  } else {
  int npad = 64;
  while (npad < max_n) npad <<= 1;
  // bitonic: one 64-bit key per padded slot; radix (npad >= RADIX_MIN_N): u32 keys + u16 indices + the [16][1024] u16 count table
  const size_t lds = npad >= RADIX_MIN_N ? (size_t)RADIX_CAP * 6 + (size_t)16 * SORT_THREADS * 2 : (size_t)npad * 8;
  if (mode == 0) {
    (void)hipFuncSetAttribute((const void*)nms_sort_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nms_sort_kernel<0>, dim3(bs), dim3(SORT_THREADS), lds, st, boxes, scores, idxs, count, n_fixed, max_n, (char*)workspace, L);
  } else {
    (void)hipFuncSetAttribute((const void*)nms_sort_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nms_sort_kernel<1>, dim3(bs), dim3(SORT_THREADS), lds, st, boxes, scores, idxs, count, n_fixed, max_n, (char*)workspace, L);
  }
  }
  const int tiles = (max_n + 63) / 64;
  if (mode == 0) hipLaunchKernelGGL(nms_mask_kernel<0>, dim3(tiles, tiles, bs), dim3(WAVE), 0, st, (char*)workspace, L, thr);
  else hipLaunchKernelGGL(nms_mask_kernel<1>, dim3(tiles, tiles, bs), dim3(WAVE), 0, st, (char*)workspace, L, thr);
  const size_t scan_lds = 2 * sizeof(unsigned long long) * (size_t)L.words;      // kept bits per tile + removed bits per column block
  (void)hipFuncSetAttribute((const void*)nms_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scan_lds);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(bs), dim3(SCAN_THREADS), scan_lds, st, (char*)workspace, L, out_count);
  if (mode == 0) hipLaunchKernelGGL(nms_remover_kernel, dim3((max_n + 63) / 64, bs), dim3(REMOVER_WAVES * WAVE), 0, st, (char*)workspace, L);
  return 0;
}

}  // namespace

extern "C" {

size_t mi355det_nms_workspace(int32_t bs, int32_t max_n) {
  if (max_n <= 0 || bs <= 0) return 0;
  return nms_layout(max_n).stride * (size_t)bs;
}

int mi355det_nms_majority(const float* boxes, const int32_t* count, int32_t bs, int32_t max_n, float thresh_iou, int32_t num_classes,
                          float* out_rows, int32_t* out_idx, int32_t* out_count, void* workspace, size_t workspace_bytes, void* stream) {
  if (bs <= 0 || num_classes <= 0 || num_classes > 8192) return fail(MI355DET_EINVAL, "%s: bad bs / num_classes (1..8192)", "nms_majority");
  NmsWs L;
  if (int e = launch_nms_common(0, boxes, nullptr, nullptr, count, 0, bs, max_n, thresh_iou, workspace, workspace_bytes, out_count, S(stream), L))
    return e;
  const size_t lds = (size_t)VOTE_WAVES * num_classes * sizeof(int);
  (void)hipFuncSetAttribute((const void*)nms_vote_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int vb = max(1, min(64, (max_n + VOTE_WAVES - 1) / VOTE_WAVES));
  hipLaunchKernelGGL(nms_vote_kernel, dim3(vb, bs), dim3(VOTE_WAVES * WAVE), lds, S(stream), boxes, (char*)workspace, L, max_n, thresh_iou,
                     num_classes, out_rows, out_idx);
  return check_launch("nms_majority");
}

int mi355det_nms(const float* boxes, const float* scores, const int64_t* idxs, int32_t n, float iou_thr, int64_t* keep, int32_t* keep_count,
                 void* workspace, size_t workspace_bytes, void* stream) {
  if (n == 0) {
    if (hipMemsetAsync(keep_count, 0, sizeof(int32_t), S(stream)) != hipSuccess) return fail(MI355DET_ELAUNCH, "%s: memset failed", "nms");
    return 0;
  }
  NmsWs L;
  if (int e = launch_nms_common(1, boxes, scores, (const long long*)idxs, nullptr, n, 1, n, iou_thr, workspace, workspace_bytes, keep_count,
                                S(stream), L))
    return e;
  hipLaunchKernelGGL(nms_keep_kernel, dim3(min(64, (n + 255) / 256), 1), dim3(256), 0, S(stream), (char*)workspace, L, (long long*)keep, keep_count, n);
  return check_launch("nms");
}

int mi355det_nms_batch(const float* boxes, const float* scores, const int64_t* idxs, int32_t bs, int32_t n, float iou_thr, int64_t* keep,
                       int32_t* keep_count, void* workspace, size_t workspace_bytes, void* stream) {
  if (bs <= 0 || n <= 0) return fail(MI355DET_EINVAL, "%s: bs and n must be positive", "nms_batch");
  if (!boxes || !scores || !keep || !keep_count) return fail(MI355DET_EINVAL, "%s: null argument", "nms_batch");
  NmsWs L;
  if (int e = launch_nms_common(1, boxes, scores, (const long long*)idxs, nullptr, n, bs, n, iou_thr, workspace, workspace_bytes, keep_count,
                                S(stream), L))
    return e;
  hipLaunchKernelGGL(nms_keep_kernel, dim3(min(64, (n + 255) / 256), bs), dim3(256), 0, S(stream), (char*)workspace, L, (long long*)keep, keep_count, n);
  return check_launch("nms_batch");
}

int mi355det_box_iou(const float* boxes1, const float* boxes2, float* out, int64_t m, int64_t n, void* stream) {
  if (m < 0 || n < 0) return fail(MI355DET_EINVAL, "%s: bad shape", "box_iou");
  if (m * n == 0) return 0;
  const int blocks = (int)min((long long)4096, (long long)((m * n + 255) / 256));
  hipLaunchKernelGGL(box_iou_kernel, dim3(blocks), dim3(256), 0, S(stream), boxes1, boxes2, out, (long long)m, (long long)n);
  return check_launch("box_iou");
}

int mi355det_match_anchors(const float* gt, const float* anchors, int32_t m, int64_t n, float high_thr, float low_thr, int allow_low_quality,
                           uint32_t* gt_best, int64_t* matches, void* stream) {
  // reference raises ValueError on empty inputs (tvision/_utils.py:282-291): the Python mirror does that
  if (m <= 0 || n <= 0) return fail(MI355DET_EINVAL, "%s: empty ground truth or proposals", "match_anchors");
  if (m > MATCH_MAX_GT) return fail(MI355DET_EINVAL, "%s: more than %lld GT boxes per image", "match_anchors", MATCH_MAX_GT);
  if (hipMemsetAsync(gt_best, 0, sizeof(uint32_t) * (size_t)m, S(stream)) != hipSuccess) return fail(MI355DET_ELAUNCH, "%s: memset failed", "match");
  const int blocks = (int)((n + MATCH_THREADS - 1) / MATCH_THREADS);
  hipLaunchKernelGGL(match_pass1_kernel, dim3(min(blocks, 256)), dim3(MATCH_THREADS), (sizeof(float4) + sizeof(unsigned)) * m, S(stream), gt, anchors,
                     m, (long long)n, high_thr, low_thr, gt_best, (long long*)matches);
  if (allow_low_quality)
    hipLaunchKernelGGL(match_pass2_kernel, dim3(blocks), dim3(MATCH_THREADS), (sizeof(float4) + sizeof(unsigned)) * m, S(stream), gt, anchors, m,
                       (long long)n, gt_best, (long long*)matches);
  return check_launch("match_anchors");
}

int mi355det_box_encode(const float* reference_boxes, const float* proposals, float* out, int64_t n, float wx, float wy, float ww, float wh,
                        void* stream) {
  if (n < 0) return fail(MI355DET_EINVAL, "%s: bad shape", "box_encode");
  if (n == 0) return 0;
  hipLaunchKernelGGL(box_encode_kernel, dim3((int)min((long long)2048, (long long)((n + 255) / 256))), dim3(256), 0, S(stream), reference_boxes,
                     proposals, out, (long long)n, wx, wy, ww, wh);
  return check_launch("box_encode");
}

int mi355det_box_decode(const float* codes, const float* boxes, float* out, int64_t n, int32_t k, float wx, float wy, float ww, float wh, float clip,
                        void* stream) {
  if (n < 0 || k <= 0) return fail(MI355DET_EINVAL, "%s: bad shape", "box_decode");
  if (n == 0) return 0;
  hipLaunchKernelGGL(box_decode_kernel, dim3((int)min((long long)2048, (long long)((n * k + 255) / 256))), dim3(256), 0, S(stream), codes, boxes,
                     out, (long long)n, k, wx, wy, ww, wh, clip);
  return check_launch("box_decode");
}

int mi355det_anchor_grid(const float* cell, int32_t a, int32_t gh, int32_t gw, int32_t stride_h, int32_t stride_w, float* out, void* stream) {
  if (a <= 0 || gh <= 0 || gw <= 0) return fail(MI355DET_EINVAL, "%s: bad shape", "anchor_grid");
  const long long total = (long long)a * gh * gw;
  hipLaunchKernelGGL(anchor_grid_kernel, dim3((int)min((long long)2048, (total + 255) / 256)), dim3(256), 0, S(stream), cell, a, gh, gw, stride_h,
                     stride_w, out);
  return check_launch("anchor_grid");
}

// Launch geometry of the focal kernels.  Small problems (K = 91: 11 M elements, 87 MB) finish in a few iterations per thread, and the ONE
// atomic each workgroup adds to the loss word then matters (atomics to one address retire at ~90 per microsecond): 512 workgroups of 1024
// threads.  Large ones (K = 1204: 145 M elements) are bandwidth / VALU bound and balance better over many small workgroups.
// reduction = 'none' (torchvision's default): the unreduced loss (and d loss / d x) per element, the same evaluation `sfl` as the summed form
__global__ __launch_bounds__(256) void focal_elem_kernel(const float* __restrict__ x, const float* __restrict__ t, long long n, float alpha, float gamma,
                                                         float* __restrict__ loss, float* __restrict__ grad) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float l, g;
    sfl(x[i], t[i], alpha, gamma, l, g);
    loss[i] = l;
    if (grad) grad[i] = g;
  }
}

#define FOCAL_LAUNCH(MODE, total_elems, ...)                                                                                                   \
  do {                                                                                                                                         \
    const long long q4_ = (total_elems) / 4 + 1;                                                                                               \
    if ((total_elems) >= (64ll << 20)) {                                                                                                       \
      const int blocks_ = (int)min((long long)256 * 32, (q4_ + 255) / 256);                                                                    \
      hipLaunchKernelGGL((focal_kernel<MODE, 256>), dim3(blocks_), dim3(256), 0, __VA_ARGS__);                                                 \
    } else {                                                                                                                                   \
      const int blocks_ = (int)min((long long)512, (q4_ + FOCAL_THREADS - 1) / FOCAL_THREADS);                                                 \
      hipLaunchKernelGGL((focal_kernel<MODE, FOCAL_THREADS>), dim3(blocks_), dim3(FOCAL_THREADS), 0, __VA_ARGS__);                             \
    }                                                                                                                                          \
  } while (0)

int mi355det_sigmoid_focal_loss_elem(const float* x, const float* t, int64_t n, float alpha, float gamma, float* loss, float* grad, void* stream) {
  if (n == 0) return 0;
  if (n < 0 || !x || !t || !loss) return fail(MI355DET_EINVAL, "%s: bad arguments", "sigmoid_focal_loss_elem");
  hipLaunchKernelGGL(focal_elem_kernel, dim3((int)min((long long)4096, (long long)((n + 255) / 256))), dim3(256), 0, S(stream), x, t, (long long)n, alpha, gamma,
                     loss, grad);
  return check_launch("sigmoid_focal_loss_elem");
}

int mi355det_sigmoid_focal_loss(const float* x, const float* t, const float* scale, const uint8_t* valid, int64_t rows, int32_t k, float alpha,
                                float gamma, float grad_scale, float* loss_sum, float* grad, void* stream) {
  if (rows < 0 || k <= 0) return fail(MI355DET_EINVAL, "%s: bad shape", "sigmoid_focal_loss");
  if (rows == 0) return 0;
  FOCAL_LAUNCH(0, (long long)rows * k, S(stream), x, t, (const long long*)nullptr, (const long long*)nullptr, scale, valid, (long long)rows, k, alpha,
               gamma, grad_scale, loss_sum, grad, (const float*)nullptr, 0ll, (const int*)nullptr, 1.f);
  return check_launch("sigmoid_focal_loss");
}

int mi355det_retina_cls_loss(const float* logits, const int64_t* matched, const int64_t* gt_labels, const float* scale, int64_t rows, int32_t k,
                             float alpha, float gamma, float grad_scale, float* loss_sum, float* grad, void* stream) {
  if (rows < 0 || k <= 0) return fail(MI355DET_EINVAL, "%s: bad shape", "retina_cls_loss");
  if (rows == 0) return 0;
  FOCAL_LAUNCH(1, (long long)rows * k, S(stream), logits, (const float*)nullptr, (const long long*)matched, (const long long*)gt_labels, scale,
               (const unsigned char*)nullptr, (long long)rows, k, alpha, gamma, grad_scale, loss_sum, grad, (const float*)nullptr, 0ll,
               (const int*)nullptr, 1.f);
  return check_launch("retina_cls_loss");
}

int mi355det_retina_loss(const float* cls_logits, const float* bbox_regression, const float* anchors, const int64_t* matched,
                         const float* gt_boxes, const int64_t* gt_labels, const int32_t* gt_offsets, const float* class_scale, int32_t n_images,
                         int64_t rows_per_image, int32_t k, float alpha, float gamma, float grad_scale, float* num_fg, float* losses,
                         float* grad_logits, float* grad_regression, void* stream) {
  if (n_images <= 0 || rows_per_image <= 0 || k <= 0) return fail(MI355DET_EINVAL, "%s: bad shape", "retina_loss");
  if (!cls_logits || !bbox_regression || !anchors || !matched || !gt_boxes || !gt_labels || !gt_offsets || !num_fg || !losses)
    return fail(MI355DET_EINVAL, "%s: null argument", "retina_loss");
  hipStream_t st = S(stream);
  (void)hipMemsetAsync(num_fg, 0, sizeof(float) * n_images, st);
  (void)hipMemsetAsync(losses, 0, sizeof(float) * 2, st);
  hipLaunchKernelGGL(count_fg_kernel, dim3((int)min((long long)64, (long long)((rows_per_image + 255) / 256)), n_images), dim3(256), 0, st,
                     (const long long*)matched, (long long)rows_per_image, num_fg);
  const long long rows = (long long)n_images * rows_per_image;
  FOCAL_LAUNCH(1, rows * k, st, cls_logits, (const float*)nullptr, (const long long*)matched, (const long long*)gt_labels, class_scale,
               (const unsigned char*)nullptr, rows, k, alpha, gamma, grad_scale, losses, grad_logits, (const float*)num_fg, (long long)rows_per_image,
               (const int*)gt_offsets, 1.0f / (float)n_images);
  hipLaunchKernelGGL(retina_reg_kernel, dim3((int)min((long long)2048, (rows + 255) / 256)), dim3(256), 0, st, bbox_regression, anchors,
                     (const long long*)matched, gt_boxes, (const int*)gt_offsets, (const float*)num_fg, n_images, (long long)rows_per_image, 1.f, 1.f,
                     1.f, 1.f, grad_scale, losses + 1, grad_regression);
  return check_launch("retina_loss");
}

int mi355det_retina_loss_lv(const float* cls_logits, const float* bbox_regression, const float* anchors, const int64_t* matched,
                            const float* gt_boxes, const int64_t* gt_labels, const int32_t* gt_offsets, const float* class_scale, int32_t n_images,
                            int64_t rows_per_image, int32_t k, float alpha, float gamma, float grad_scale, float* num_fg, float* losses,
                            const mi355det_level_grads* cls_levels, float* grad_regression, void* stream) {
  if (n_images <= 0 || rows_per_image <= 0 || k <= 0) return fail(MI355DET_EINVAL, "%s: bad shape", "retina_loss_lv");
  if (!cls_logits || !bbox_regression || !anchors || !matched || !gt_boxes || !gt_labels || !gt_offsets || !num_fg || !losses || !cls_levels)
    return fail(MI355DET_EINVAL, "%s: null argument", "retina_loss_lv");
  if (cls_levels->n_levels < 1 || cls_levels->n_levels > 8 || cls_levels->anchors_per_pixel < 1)
    return fail(MI355DET_EINVAL, "%s: 1..8 levels and >= 1 anchor per pixel", "retina_loss_lv");
  FocalLevels lv{};
  lv.nlev = cls_levels->n_levels;
  lv.A = cls_levels->anchors_per_pixel;
  long long at = 0;
  for (int q = 0; q < lv.nlev; ++q) {
    if (!cls_levels->grad[q] || cls_levels->pixels[q] <= 0 || cls_levels->grad_ld[q] < lv.A * k)
      return fail(MI355DET_EINVAL, "%s: level %d: null buffer, no pixels or pitch < anchors * classes", "retina_loss_lv", q);
    lv.start[q] = at;
    lv.pixels[q] = cls_levels->pixels[q];
    lv.dst[q] = (bf16_t*)cls_levels->grad[q];
    lv.ld[q] = cls_levels->grad_ld[q];
    at += cls_levels->pixels[q] * lv.A;
  }
  if (at != rows_per_image) return fail(MI355DET_EINVAL, "%s: the levels hold %lld rows per image, the logits %lld", "retina_loss_lv", at, (long long)rows_per_image);
  {
    const long long rows_all = (long long)n_images * rows_per_image;
    // 1: every group of four lies in one row and is 8-byte aligned in the level buffers; 2: any k (groups that straddle a row take the general
    // path, unaligned groups store element by element)
    lv.fast = rows_all * k < (1ll << 31) ? ((k % 4 == 0 && lv.A * k % 4 == 0) ? 1 : (k >= 4 ? 2 : 0)) : 0;      // 2: a group of four spans at most two rows
    for (int q = 0; q < lv.nlev; ++q)
      if (lv.fast == 1 && (lv.ld[q] % 4 != 0 || (((uintptr_t)lv.dst[q]) & 7) != 0)) lv.fast = 2;
    lv.dk = focal_div((unsigned)k);
    lv.drpi = focal_div((unsigned)rows_per_image);
    lv.da = focal_div((unsigned)lv.A);
  }
  hipStream_t st = S(stream);
  (void)hipMemsetAsync(num_fg, 0, sizeof(float) * n_images, st);
  (void)hipMemsetAsync(losses, 0, sizeof(float) * 2, st);
  hipLaunchKernelGGL(count_fg_kernel, dim3((int)min((long long)64, (long long)((rows_per_image + 255) / 256)), n_images), dim3(256), 0, st,
                     (const long long*)matched, (long long)rows_per_image, num_fg);
  const long long rows = (long long)n_images * rows_per_image;
  FOCAL_LAUNCH(1, rows * k, st, cls_logits, (const float*)nullptr, (const long long*)matched, (const long long*)gt_labels, class_scale,
               (const unsigned char*)nullptr, rows, k, alpha, gamma, grad_scale, losses, (float*)nullptr, (const float*)num_fg, (long long)rows_per_image,
               (const int*)gt_offsets, 1.0f / (float)n_images, lv);
  hipLaunchKernelGGL(retina_reg_kernel, dim3((int)min((long long)2048, (rows + 255) / 256)), dim3(256), 0, st, bbox_regression, anchors,
                     (const long long*)matched, gt_boxes, (const int*)gt_offsets, (const float*)num_fg, n_images, (long long)rows_per_image, 1.f, 1.f,
                     1.f, 1.f, grad_scale, losses + 1, grad_regression);
  return check_launch("retina_loss_lv");
}

}  // extern "C"
