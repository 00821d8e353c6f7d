// YOLO criterion kernels (wavefront-reduction family, HBM-bound; no MFMA — not contractions).
// Replaces the per-image Python loops of yolo/nets/yolo_forw.py and yolo/utilities/helper.py.
// Compiled with -ffp-contract=off: index decisions (argmax, thresholds) must see the same IEEE
// float32 values as the reference's unfused torch ops.
#include "common.h"

using namespace mi355;

namespace {

struct Box {
  float x1, y1, x2, y2;
};

// helper.get_abs_coord (helper.py:203-217)
__device__ __forceinline__ Box abs_coord(float xc, float yc, float w, float h) {
  Box b;
  b.x1 = xc - w / 2.0f;
  b.y1 = yc - h / 2.0f;
  b.x2 = xc + w / 2.0f;
  b.y2 = yc + h / 2.0f;
  return b;
}

// helper.bbox_iou (helper.py:245-277), same operation order, float32
__device__ __forceinline__ float iou_metric(const Box a, const Box b, int iou_type) {
  float inter = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.0f) * fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.0f);
  float w1 = a.x2 - a.x1, h1 = a.y2 - a.y1;
  float w2 = b.x2 - b.x1, h2 = b.y2 - b.y1;
  float uni = (w1 * h1 + 1e-16f) + w2 * h2 - inter;
  float iou = inter / uni;
  if (iou_type == 0) return iou;
  float cw = fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1);
  float ch = fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1);
  if (iou_type == 1) {
    float c_area = cw * ch + 1e-16f;
    return iou - (c_area - uni) / c_area;
  }
  float c2 = cw * cw + ch * ch + 1e-16f;
  float dx = (b.x1 + b.x2) - (a.x1 + a.x2), dy = (b.y1 + b.y2) - (a.y1 + a.y2);
  float rho2 = dx * dx / 4.0f + dy * dy / 4.0f;
  if (iou_type == 2) return iou - rho2 / c2;
  float d = atanf(w2 / h2) - atanf(w1 / h1);
  float v = 0.40528473456935109f * (d * d);
  float alpha = v / (1.0f - iou + v);
  return iou - (rho2 / c2 + v * alpha);
}

struct Geom {  // LDS copy with dynamic indexing
  int num_scales, na, num_classes, attrs;
  float img_size, ignore_thr;
  int iou_type;
  int grid[MI355DET_MAX_SCALES];
  int off[MI355DET_MAX_SCALES + 1];
  float aw[MI355DET_MAX_SCALES][MI355DET_MAX_ANCHORS];
  float ah[MI355DET_MAX_SCALES][MI355DET_MAX_ANCHORS];
};

__device__ __forceinline__ void load_geom(Geom& s, const mi355det_yolo_geom& g) {
  if (threadIdx.x == 0) {
    s.num_scales = g.num_scales;
    s.na = g.na;
    s.num_classes = g.num_classes;
    s.attrs = g.num_classes + 5;
    s.img_size = g.img_size;
    s.ignore_thr = g.ignore_thr;
    s.iou_type = g.iou_type;
#pragma unroll
    for (int k = 0; k < MI355DET_MAX_SCALES; ++k) {
      s.grid[k] = g.grid[k];
      s.off[k] = g.off[k];
#pragma unroll
      for (int a = 0; a < MI355DET_MAX_ANCHORS; ++a) {
        s.aw[k][a] = g.anchor_w[k][a];
        s.ah[k][a] = g.anchor_h[k][a];
      }
    }
    s.off[MI355DET_MAX_SCALES] = g.off[MI355DET_MAX_SCALES];
  }
  __syncthreads();
}

struct Anchor {
  float cx, cy, aw, ah, gridf;
  int scale, pix, a;
};

// yolo_forw.py:104-114 — anchor n of one image: index off[k] + (y*W+x)*na + a
__device__ __forceinline__ Anchor anchor_at(const Geom& g, int n) {
  Anchor r;
  int k = 0;
  while (k + 1 < g.num_scales && n >= g.off[k + 1]) ++k;
  int local = n - g.off[k];
  r.scale = k;
  r.a = local % g.na;
  r.pix = local / g.na;
  int W = g.grid[k];
  int y = r.pix / W, x = r.pix - y * W;
  r.gridf = (float)W;
  r.cx = ((float)x + 0.5f) / r.gridf;
  r.cy = ((float)y + 0.5f) / r.gridf;
  r.aw = g.aw[k][r.a];
  r.ah = g.ah[k][r.a];
  return r;
}

// ------------------------------------------------------------------------------------------
// helper.bbox_iou standalone
__global__ void bbox_iou_kernel(const float* __restrict__ bb1, const float* __restrict__ bb2, float* __restrict__ out,
                                long long m, long long n, int iou_type, int xcycwh, int paired) {
  long long total = paired ? n : m * n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long r = paired ? i : i / n, c = paired ? i : i % n;
    const float4 p = *(const float4*)(bb1 + 4 * r);
    const float4 q = *(const float4*)(bb2 + 4 * c);
    Box a, b;
    if (xcycwh) {
      a = abs_coord(p.x, p.y, p.z, p.w);
      b = abs_coord(q.x, q.y, q.z, q.w);
    } else {
      a = {p.x, p.y, p.z, p.w};
      b = {q.x, q.y, q.z, q.w};
    }
    out[i] = iou_metric(a, b, iou_type);
  }
}

// ------------------------------------------------------------------------------------------
// get_target, pass 1 (yolo_forw.py:186-187,200): one thread per (image, anchor); every GT of the
// image is scored against the anchor generated on the fly (no [M,N] matrix, no anchor table in HBM).
// Algorithmic bytes per image: M*16 (GT) + N (noobj byte) written.
#define ASSIGN_THREADS 256
#define GT_CHUNK 64
__global__ __launch_bounds__(ASSIGN_THREADS) void yolo_assign_kernel(mi355det_yolo_geom geom, const float* __restrict__ gt_box,
                                                                      const int* __restrict__ gt_off,
                                                                      unsigned long long* __restrict__ best_key,
                                                                      unsigned char* __restrict__ noobj, int n_total) {
  __shared__ Geom g;
  __shared__ float4 sgt[GT_CHUNK];
  __shared__ unsigned long long swave[ASSIGN_THREADS / WAVE][GT_CHUNK];
  load_geom(g, geom);
  const int b = blockIdx.y;
  const int n = blockIdx.x * ASSIGN_THREADS + threadIdx.x;
  const int g0 = gt_off[b], g1 = gt_off[b + 1];
  const bool live = n < n_total;
  Anchor an = anchor_at(g, live ? n : 0);
  const Box ab = abs_coord(an.cx, an.cy, an.aw, an.ah);
  const float thr = g.ignore_thr;
  const int iou_type = g.iou_type;
  bool below = true;
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  for (int c0 = g0; c0 < g1; c0 += GT_CHUNK) {
    const int cn = min(GT_CHUNK, g1 - c0);
    __syncthreads();
    if (threadIdx.x < cn) sgt[threadIdx.x] = *(const float4*)(gt_box + 4 * (size_t)(c0 + threadIdx.x));
    __syncthreads();
    for (int j = 0; j < cn; ++j) {
      const float4 t = sgt[j];
      const Box gb = abs_coord(t.x, t.y, t.z, t.w);
      const float v = iou_metric(gb, ab, iou_type);
      below = below && (v < thr);
      unsigned long long key = live ? (((unsigned long long)f2ord(v) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)n)) : 0ull;
      key = wave_max_u64(key);
      if (lane == 0) swave[wid][j] = key;
    }
    __syncthreads();
    if (threadIdx.x < cn) {
      unsigned long long k = swave[0][threadIdx.x];
#pragma unroll
      for (int w = 1; w < ASSIGN_THREADS / WAVE; ++w) k = max(k, swave[w][threadIdx.x]);
      atomicMax(best_key + c0 + threadIdx.x, k);
    }
  }
  if (live) noobj[(size_t)b * n_total + n] = below ? 1 : 0;
}

// get_target, pass 2 (yolo_forw.py:187-201): decode argmax, targets, clear noobj at the matched anchors
__global__ void yolo_targets_kernel(mi355det_yolo_geom geom, const float* __restrict__ gt_box, const int* __restrict__ gt_off,
                                    const unsigned long long* __restrict__ best_key, long long* __restrict__ obj_idx,
                                    float* __restrict__ tgt, unsigned char* __restrict__ noobj, int bs, int n_total) {
  __shared__ Geom g;
  load_geom(g, geom);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int G = gt_off[bs];
  if (i >= G) return;
  int b = 0;
  while (b + 1 < bs && i >= gt_off[b + 1]) ++b;
  const unsigned n = 0xFFFFFFFFu - (unsigned)(best_key[i] & 0xFFFFFFFFull);
  obj_idx[i] = (long long)n;
  const Anchor an = anchor_at(g, (int)n);
  const float4 t = *(const float4*)(gt_box + 4 * (size_t)i);
  float px = t.x * an.gridf, py = t.y * an.gridf;
  float gx = px - truncf(px), gy = py - truncf(py);
  gx = fminf(fmaxf(gx, 0.0001f), 0.9999f);
  gy = fminf(fmaxf(gy, 0.0001f), 0.9999f);
  float gw = logf(t.z / an.aw + 1e-16f);
  float gh = logf(t.w / an.ah + 1e-16f);
  *(float4*)(tgt + 4 * (size_t)i) = make_float4(gx, gy, gw, gh);
  noobj[(size_t)b * n_total + n] = 0;
}

// ------------------------------------------------------------------------------------------
// custom.FocalLoss on BCE-with-logits (custom.py:50-60): value and d/dx
__device__ __forceinline__ void focal_bce(float x, float t, float alpha, float gamma, float& loss, float& grad, float& prob) {
  const float e = __expf(-fabsf(x));
  const float bce = fmaxf(x, 0.0f) - x * t + log1pf(e);
  const float p = x >= 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
  const float p_t = t * p + (1.0f - t) * (1.0f - p);
  const float af = t * alpha + (1.0f - t) * (1.0f - alpha);
  const float q = 1.0f - p_t;
  const float mf = (gamma == 1.0f) ? q : (gamma == 2.0f ? q * q : powf(q, gamma));
  loss = bce * (af * mf);
  const float dpt = (2.0f * t - 1.0f) * p * (1.0f - p);
  float dmf;
  if (gamma == 1.0f) dmf = -dpt;
  else if (gamma == 2.0f) dmf = -2.0f * q * dpt;
  else dmf = q > 0.0f ? -gamma * powf(q, gamma - 1.0f) * dpt : 0.0f;
  grad = af * ((p - t) * mf + bce * dmf);
  prob = p;
}

// nn.BCEWithLogitsLoss(pos_weight=pw) on one element (ATen: (1-y)*x + (1+(pw-1)*y) * softplus(-x)), optionally inside custom.EQLoss
// (custom.py:83-99: * alpha_factor * (1-p_t)^gamma * clamp(eq_mask + y, 0, 1)); value and d/dx
__device__ __forceinline__ void bce_class(float x, float y, float pw, bool eql, float eq_mask, float alpha, float gamma, float& loss,
                                          float& grad) {
  const float e = __expf(-fabsf(x));
  const float lw = 1.0f + (pw - 1.0f) * y;
  const float bce = (1.0f - y) * x + lw * (log1pf(e) + fmaxf(-x, 0.0f));
  const float p = x >= 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
  const float dbce = (1.0f - y) - lw * (1.0f - p);
  if (!eql) {
    loss = bce;
    grad = dbce;
    return;
  }
  const float p_t = y * p + (1.0f - y) * (1.0f - p);
  const float af = y * alpha + (1.0f - y) * (1.0f - alpha);
  const float q = 1.0f - p_t;
  const float mf = (gamma == 1.0f) ? q : (gamma == 2.0f ? q * q : powf(q, gamma));
  const float dpt = (2.0f * y - 1.0f) * p * (1.0f - p);
  float dmf;
  if (gamma == 1.0f) dmf = -dpt;
  else if (gamma == 2.0f) dmf = -2.0f * q * dpt;
  else dmf = q > 0.0f ? -gamma * powf(q, gamma - 1.0f) * dpt : 0.0f;
  const float w = fminf(fmaxf(eq_mask + y, 0.0f), 1.0f);
  loss = bce * (af * mf) * w;
  grad = w * af * (dbce * mf + bce * dmf);
}

// gradient view format GF: 0 = fp32, 1 = bf16, 2 = IEEE fp16 (the engine's fp16 storage: yolo/procedures/initialize.py:44-45, apex O2)
__device__ __forceinline__ bf16_t f2h(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ float h2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
template <int GF>
__device__ __forceinline__ void grad_store(void* base, long long off, float v) {
  if (GF == 1) ((bf16_t*)base)[off] = f2bf(v);
  else if (GF == 2) ((bf16_t*)base)[off] = f2h(v);
  else ((float*)base)[off] = v;
}
template <int GF>
__device__ __forceinline__ void grad_add(void* base, long long off, float v) {
  if (GF == 1) ((bf16_t*)base)[off] = f2bf(bf2f(((bf16_t*)base)[off]) + v);
  else if (GF == 2) ((bf16_t*)base)[off] = f2h(h2f(((bf16_t*)base)[off]) + v);
  else ((float*)base)[off] += v;
}

struct Views {
  mi355det_head_view h[MI355DET_MAX_SCALES];
};

// reduction='mean' only: number of no-object elements of the batch (yolo_forw.py:148 `neg_conf_loss.mean()`), one workgroup, integer sum
__global__ __launch_bounds__(1024) void yolo_noobj_count_kernel(const unsigned char* __restrict__ noobj, long long n, int* __restrict__ count) {
  __shared__ int red[1024 / WAVE];
  int c = 0;
  for (long long i = threadIdx.x; i < n; i += 1024) c += noobj[i] ? 1 : 0;
  float cf = wave_sum((float)c);                             // <= 64 * n/1024 per wave: exact in fp32 up to 2^24 per wave
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = (int)cf;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < 1024 / WAVE; ++w) t += red[w];
    *count = t;
  }
}

// No-object confidence term (yolo_forw.py:131-133,144): one thread per (image, anchor), lanes run
// over pixels within an anchor plane.  Per-block partials {loss, sum sigmoid, count}.
#define DENSE_THREADS 256
template <int GF>
__global__ __launch_bounds__(DENSE_THREADS) void yolo_noobj_kernel(mi355det_yolo_geom geom, Views heads, Views grads, int has_grad,
                                                                    const unsigned char* __restrict__ noobj, float alpha,
                                                                    float gamma, float gscale, float* __restrict__ partials,
                                                                    int n_total, const int* __restrict__ noobj_count) {
  __shared__ Geom g;
  __shared__ float red[DENSE_THREADS / WAVE][3];
  load_geom(g, geom);
  if (noobj_count) gscale /= (float)max(*noobj_count, 1);   // reduction='mean': the no-object term is averaged over its own element count
  const int b = blockIdx.y;
  const int t = blockIdx.x * DENSE_THREADS + threadIdx.x;   // plane-major index: scale, anchor a, pixel
  float l = 0.f, s = 0.f, c = 0.f;
  if (t < n_total) {
    int k = 0;
    while (k + 1 < g.num_scales && t >= g.off[k + 1]) ++k;
    const int local = t - g.off[k];
    const int hw = g.grid[k] * g.grid[k];
    const int a = local / hw, pix = local - a * hw;
    const int n = g.off[k] + pix * g.na + a;
    const long long ch = (long long)a * g.attrs + 4;
    const mi355det_head_view hv = heads.h[k];
    const float x = ((const float*)hv.ptr)[b * hv.sb + ch * hv.sc + pix * hv.sp];
    if (noobj[(size_t)b * n_total + n]) {
      float loss, grad, p;
      focal_bce(x, 0.0f, alpha, gamma, loss, grad, p);
      l = loss;
      s = p;
      c = 1.0f;
      if (has_grad) {
        const mi355det_head_view gv = grads.h[k];
        grad_store<GF>(gv.ptr, b * gv.sb + ch * gv.sc + pix * gv.sp, grad * gscale);
      }
    }
  }
  l = wave_sum(l);
  s = wave_sum(s);
  c = wave_sum(c);
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  if (lane == 0) {
    red[wid][0] = l;
    red[wid][1] = s;
    red[wid][2] = c;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < DENSE_THREADS / WAVE; ++w) v += red[w][threadIdx.x];
    partials[((size_t)b * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = v;
  }
}

// forward-mode duals over the 4 raw box logits, for d(IoU metric)/d(raw)
struct D4 {
  float v, d[4];
};
__device__ __forceinline__ D4 dconst(float v) { return {v, {0, 0, 0, 0}}; }
__device__ __forceinline__ D4 operator+(D4 a, D4 b) { return {a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2], a.d[3] + b.d[3]}}; }
__device__ __forceinline__ D4 operator-(D4 a, D4 b) { return {a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2], a.d[3] - b.d[3]}}; }
__device__ __forceinline__ D4 operator*(D4 a, D4 b) {
  return {a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2], a.d[3] * b.v + a.v * b.d[3]}};
}
__device__ __forceinline__ D4 operator/(D4 a, D4 b) {
  const float q = a.v / b.v, ib = 1.0f / b.v;
  return {q, {(a.d[0] - q * b.d[0]) * ib, (a.d[1] - q * b.d[1]) * ib, (a.d[2] - q * b.d[2]) * ib, (a.d[3] - q * b.d[3]) * ib}};
}
__device__ __forceinline__ D4 dmin(D4 a, D4 b) { return a.v <= b.v ? a : b; }
__device__ __forceinline__ D4 dmax(D4 a, D4 b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ D4 dclamp0(D4 a) { return a.v > 0.0f ? a : dconst(fmaxf(a.v, 0.0f)); }
__device__ __forceinline__ D4 datan(D4 a) {
  const float s = 1.0f / (1.0f + a.v * a.v);
  return {atanf(a.v), {a.d[0] * s, a.d[1] * s, a.d[2] * s, a.d[3] * s}};
}

struct DBox {
  D4 x1, y1, x2, y2;
};
__device__ __forceinline__ D4 iou_metric_dual(const DBox a, const DBox b, int iou_type) {
  const D4 eps = dconst(1e-16f);
  D4 inter = dclamp0(dmin(a.x2, b.x2) - dmax(a.x1, b.x1)) * dclamp0(dmin(a.y2, b.y2) - dmax(a.y1, b.y1));
  D4 w1 = a.x2 - a.x1, h1 = a.y2 - a.y1, w2 = b.x2 - b.x1, h2 = b.y2 - b.y1;
  D4 uni = (w1 * h1 + eps) + w2 * h2 - inter;
  D4 iou = inter / uni;
  if (iou_type == 0) return iou;
  D4 cw = dmax(a.x2, b.x2) - dmin(a.x1, b.x1), ch = dmax(a.y2, b.y2) - dmin(a.y1, b.y1);
  if (iou_type == 1) {
    D4 ca = cw * ch + eps;
    return iou - (ca - uni) / ca;
  }
  D4 c2 = cw * cw + ch * ch + eps;
  D4 dx = (b.x1 + b.x2) - (a.x1 + a.x2), dy = (b.y1 + b.y2) - (a.y1 + a.y2);
  D4 rho2 = dx * dx / dconst(4.0f) + dy * dy / dconst(4.0f);
  if (iou_type == 2) return iou - rho2 / c2;
  D4 d = datan(w2 / h2) - datan(w1 / h1);
  D4 v = dconst(0.40528473456935109f) * (d * d);
  const float alpha = v.v / (1.0f - iou.v + v.v);   // no_grad in the reference (helper.py:273-274)
  return iou - (rho2 / c2 + v * dconst(alpha));
}

// Positive terms (yolo_forw.py:123-150 + transform_pred + get_stats): one wave per image walks its
// GTs in order (duplicate anchor assignments accumulate, like the reference's gather/scatter).
// partial layout per image: xy, wh, iou_loss, pos_conf, cls, iou_sum, pconf_sum, pcls_sum, sum of sigmoid(raw) over all classes
// (class_loss 0/2 statistics), sum of the target class weights (CrossEntropyLoss(weight, 'mean') divisor)
#define POS_NP 10
template <int GF>
__global__ __launch_bounds__(WAVE) void yolo_pos_kernel(mi355det_yolo_geom geom, mi355det_yolo_loss_cfg cfg, Views heads, Views grads,
                                                         int has_grad, const int* __restrict__ gt_off,
                                                         const long long* __restrict__ gt_label,
                                                         const long long* __restrict__ obj_idx, const float* __restrict__ tgt,
                                                         const float* __restrict__ idf, float inv_ng, int bs,
                                                         float* __restrict__ pos_partials) {
  __shared__ Geom g;
  load_geom(g, geom);
  const int b = blockIdx.x, lane = threadIdx.x;
  const int C = g.num_classes;
  float acc[POS_NP];
#pragma unroll
  for (int i = 0; i < POS_NP; ++i) acc[i] = 0.f;
  // gradient scales per term.  reduction='sum': every term / sum(M) (yolo_forw.py:158-160).  'mean': MSELoss over [G,2] elements,
  // FocalLoss / (1 - iou) over G, class term over its own divisor (CrossEntropyLoss: summed target weights; BCE forms: G*C)
  const bool mean = cfg.reduction_mean != 0;
  const float gs_xy = cfg.grad_scale * inv_ng * (mean ? 0.5f : 1.0f);
  const float gs = cfg.grad_scale * inv_ng;
  float gs_cls = gs;
  if (mean) {
    if (cfg.class_loss == 1) {
      float ws = 0.f;
      const int total = gt_off[bs];
      for (int i = lane; i < total; i += WAVE) ws += cfg.class_weights ? cfg.class_weights[(int)gt_label[i]] : 1.0f;
      gs_cls = cfg.grad_scale / wave_sum(ws);
    } else {
      gs_cls = gs / (float)C;
    }
  }
  for (int i = gt_off[b]; i < gt_off[b + 1]; ++i) {
    const int n = (int)obj_idx[i];
    const Anchor an = anchor_at(g, n);
    const mi355det_head_view hv = heads.h[an.scale];
    const float* hp = (const float*)hv.ptr + b * hv.sb + (long long)an.pix * hv.sp;
    const long long ch0 = (long long)an.a * g.attrs;
    float r[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) r[k] = hp[(ch0 + k) * hv.sc];
    const float4 t = *(const float4*)(tgt + 4 * (size_t)i);
    // --- box terms (every lane computes them redundantly; lane k<5 writes gradient k)
    const float sx = 1.0f / (1.0f + __expf(-r[0])), sy = 1.0f / (1.0f + __expf(-r[1]));
    const float stride = g.img_size / an.gridf;
    // transform_pred (yolo_forw.py:217-218,227-228): pixels
    D4 px = {(sx + an.cx * an.gridf - 0.5f) * stride, {sx * (1 - sx) * stride, 0, 0, 0}};
    D4 py = {(sy + an.cy * an.gridf - 0.5f) * stride, {0, sy * (1 - sy) * stride, 0, 0}};
    const float pwv = __expf(r[2]) * an.aw * an.gridf * stride, phv = __expf(r[3]) * an.ah * an.gridf * stride;
    D4 pw = {pwv, {0, 0, pwv, 0}}, ph = {phv, {0, 0, 0, phv}};
    const float gxv = (t.x + an.cx * an.gridf - 0.5f) * stride, gyv = (t.y + an.cy * an.gridf - 0.5f) * stride;
    const float gwv = __expf(t.z) * an.aw * an.gridf * stride, ghv = __expf(t.w) * an.ah * an.gridf * stride;
    const D4 two = dconst(2.0f);
    DBox pb = {px - pw / two, py - ph / two, px + pw / two, py + ph / two};
    DBox gb = {dconst(gxv - gwv / 2.0f), dconst(gyv - ghv / 2.0f), dconst(gxv + gwv / 2.0f), dconst(gyv + ghv / 2.0f)};
    const D4 iou = iou_metric_dual(pb, gb, g.iou_type);
    float pl, pg, pp;
    focal_bce(r[4], 1.0f, cfg.alpha, cfg.gamma, pl, pg, pp);
    const float dxy0 = sx - t.x, dxy1 = sy - t.y, dwh0 = r[2] - t.z, dwh1 = r[3] - t.w;
    acc[0] += cfg.lambda_xy * (dxy0 * dxy0 + dxy1 * dxy1);
    acc[1] += cfg.lambda_wh * (dwh0 * dwh0 + dwh1 * dwh1);
    acc[2] += cfg.lambda_iou * (1.0f - iou.v);
    acc[3] += cfg.lambda_conf * pl;
    acc[5] += iou.v;
    acc[6] += pp;
    // --- class term: CE(idf*logits, label) (yolo_forw.py:136); lanes over classes
    const int label = (int)gt_label[i];
    float m = -INFINITY, mr = -INFINITY;
    for (int c = lane; c < C; c += WAVE) {
      const float raw = hp[(ch0 + 5 + c) * hv.sc];
      const float z = idf ? idf[c] * raw : raw;
      m = fmaxf(m, z);
      mr = fmaxf(mr, raw);
    }
    m = wave_max(m);
    mr = wave_max(mr);
    float se = 0.f, ser = 0.f, zl = 0.f, rl = 0.f;
    for (int c = lane; c < C; c += WAVE) {
      const float raw = hp[(ch0 + 5 + c) * hv.sc];
      const float z = idf ? idf[c] * raw : raw;
      se += __expf(z - m);
      ser += __expf(raw - mr);
      if (c == label) {
        zl = z;
        rl = raw;
      }
    }
    se = wave_sum(se);
    ser = wave_sum(ser);
    zl = wave_sum(zl);
    rl = wave_sum(rl);
    const float wy = cfg.class_weights ? cfg.class_weights[label] : 1.0f;       // CrossEntropyLoss(weight=w): w[y] * nll
    if (cfg.class_loss == 1) {
      acc[4] += cfg.lambda_cls * wy * ((m + __logf(se)) - zl);
      acc[7] += __expf(rl - mr) / ser;   // softmax(raw)[label] (transform_pred :222, get_stats :243)
      acc[9] += wy;
    } else {
      // class_loss 0: BCEWithLogitsLoss(pos_weight) on the one-hot row; 2: the same inside custom.EQLoss (custom.py:83-99: focal factors and
      // the (rare-class mask + target) weights).  Statistics use sigmoid(raw) (transform_pred :223)
      float ls = 0.f, ss = 0.f;
      for (int c = lane; c < C; c += WAVE) {
        const float raw = hp[(ch0 + 5 + c) * hv.sc];
        const float w = idf ? idf[c] : 1.0f;
        const float y = c == label ? 1.0f : 0.0f;
        float l, gr;
        bce_class(w * raw, y, cfg.class_weights ? cfg.class_weights[c] : 1.0f, cfg.class_loss == 2, cfg.eq_mask ? cfg.eq_mask[c] : 0.0f,
                  cfg.alpha, cfg.gamma, l, gr);
        ls += l;
        ss += 1.0f / (1.0f + __expf(-raw));
        if (has_grad) {
          const mi355det_head_view gv = grads.h[an.scale];
          grad_add<GF>(gv.ptr, b * gv.sb + (long long)an.pix * gv.sp + (ch0 + 5 + c) * gv.sc, cfg.lambda_cls * w * gr * gs_cls);
        }
      }
      acc[4] += cfg.lambda_cls * wave_sum(ls);
      acc[7] += 1.0f / (1.0f + __expf(-rl));
      acc[8] += wave_sum(ss);
    }
    if (has_grad) {
      const mi355det_head_view gv = grads.h[an.scale];
      const long long gb0 = b * gv.sb + (long long)an.pix * gv.sp;
      if (lane < 5) {
        float gr;
        if (lane == 0) gr = cfg.lambda_xy * 2.0f * dxy0 * sx * (1 - sx) * gs_xy - cfg.lambda_iou * iou.d[0] * gs;
        else if (lane == 1) gr = cfg.lambda_xy * 2.0f * dxy1 * sy * (1 - sy) * gs_xy - cfg.lambda_iou * iou.d[1] * gs;
        else if (lane == 2) gr = cfg.lambda_wh * 2.0f * dwh0 * gs_xy - cfg.lambda_iou * iou.d[2] * gs;
        else if (lane == 3) gr = cfg.lambda_wh * 2.0f * dwh1 * gs_xy - cfg.lambda_iou * iou.d[3] * gs;
        else gr = cfg.lambda_conf * pg * gs;
        grad_add<GF>(gv.ptr, gb0 + (ch0 + lane) * gv.sc, gr);
      }
      if (cfg.class_loss == 1) {
        for (int c = lane; c < C; c += WAVE) {
          const float raw = hp[(ch0 + 5 + c) * hv.sc];
          const float w = idf ? idf[c] : 1.0f;
          const float z = w * raw;
          float gr = __expf(z - m) / se - (c == label ? 1.0f : 0.0f);
          grad_add<GF>(gv.ptr, gb0 + (ch0 + 5 + c) * gv.sc, cfg.lambda_cls * wy * w * gr * gs_cls);
        }
      }
    }
  }
  if (lane < POS_NP) pos_partials[(size_t)b * POS_NP + lane] = acc[lane];
}

// Final reduction in a fixed order (deterministic): out12 = loss, sub_losses[6], stats[5]
__global__ void yolo_reduce_kernel(const float* __restrict__ dense_partials, int n_dense, const float* __restrict__ pos_partials,
                                   int bs, float lambda_no_conf, float ng, int num_classes, int class_loss, int reduction_mean,
                                   float* __restrict__ out12) {
  __shared__ double sh[256][3];
  double a[3] = {0, 0, 0};
  for (int i = threadIdx.x; i < n_dense; i += blockDim.x)
    for (int k = 0; k < 3; ++k) a[k] += (double)dense_partials[(size_t)i * 3 + k];
  for (int k = 0; k < 3; ++k) sh[threadIdx.x][k] = a[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    double d[3] = {0, 0, 0};
    for (int t = 0; t < (int)blockDim.x; ++t)
      for (int k = 0; k < 3; ++k) d[k] += sh[t][k];
    double p[POS_NP];
    for (int k = 0; k < POS_NP; ++k) p[k] = 0;
    for (int b = 0; b < bs; ++b)
      for (int k = 0; k < POS_NP; ++k) p[k] += (double)pos_partials[(size_t)b * POS_NP + k];
    const double neg = lambda_no_conf * d[0];
    double sub[6] = {p[0], p[1], p[2], p[3], neg, p[4]};
    if (reduction_mean) {
      // every term averaged over its own element count, no final / sum(M) (yolo_forw.py:143-160 with reduction != 'sum')
      const double dcls = class_loss == 1 ? p[9] : ng * (double)num_classes;
      const double div[6] = {2.0 * ng, 2.0 * ng, ng, ng, d[2] > 0 ? d[2] : NAN, dcls};
      for (int k = 0; k < 6; ++k) sub[k] /= div[k];
    } else {
      for (int k = 0; k < 6; ++k) sub[k] /= ng;
    }
    double loss = 0;
    for (int k = 0; k < 6; ++k) loss += sub[k];
    out12[0] = (float)loss;
    for (int k = 0; k < 6; ++k) out12[1 + k] = (float)sub[k];
    out12[7] = (float)(p[5] / ng);                         // avg_iou
    out12[8] = (float)(p[6] / ng);                         // pos_conf
    out12[9] = (float)(d[2] > 0 ? d[1] / d[2] : NAN);      // no_obj_conf (mean of empty = nan, like torch)
    out12[10] = (float)(p[7] / ng);                        // pos_class
    // neg_class: mean of the class probabilities off the label; softmax rows sum to 1, sigmoid rows are summed explicitly
    const double off_label = class_loss == 1 ? ng - p[7] : p[8] - p[7];
    out12[11] = (float)(off_label / (ng * (double)(num_classes - 1)));
  }
}

// ------------------------------------------------------------------------------------------
// Inference decode (yolo_forw.py:163-176): one wave per (image, anchor) row, lanes over attributes.
// Algorithmic bytes: 2 * bs*N*attrs*4.
#define DEC_WAVES 4
__global__ __launch_bounds__(DEC_WAVES* WAVE) void yolo_decode_kernel(mi355det_yolo_geom geom, Views heads, const float* __restrict__ idf,
                                                                       int softmax_cls, float* __restrict__ out, int n_total,
                                                                       long long rows) {
  __shared__ Geom g;
  load_geom(g, geom);
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const int C = g.num_classes, attrs = g.attrs;
  for (long long row = (long long)blockIdx.x * DEC_WAVES + wid; row < rows; row += (long long)gridDim.x * DEC_WAVES) {
    const int b = (int)(row / n_total), n = (int)(row - (long long)b * n_total);
    const Anchor an = anchor_at(g, n);
    const mi355det_head_view hv = heads.h[an.scale];
    const float* hp = (const float*)hv.ptr + b * hv.sb + (long long)an.pix * hv.sp + (long long)an.a * attrs * hv.sc;
    float* op = out + row * attrs;
    const float stride = g.img_size / an.gridf;
    if (lane < 5) {
      const float x = hp[lane * hv.sc];
      float v;
      if (lane == 0) v = (1.0f / (1.0f + __expf(-x)) + an.cx * an.gridf - 0.5f) * stride;
      else if (lane == 1) v = (1.0f / (1.0f + __expf(-x)) + an.cy * an.gridf - 0.5f) * stride;
      else if (lane == 2) v = __expf(x) * an.aw * an.gridf * stride;
      else if (lane == 3) v = __expf(x) * an.ah * an.gridf * stride;
      else v = 1.0f / (1.0f + __expf(-x));
      op[lane] = v;
    }
    if (softmax_cls) {
      float m = -INFINITY;
      for (int c = lane; c < C; c += WAVE) {
        const float z = (idf ? idf[c] : 1.0f) * hp[(5 + c) * hv.sc];
        m = fmaxf(m, z);
      }
      m = wave_max(m);
      float se = 0.f;
      for (int c = lane; c < C; c += WAVE) se += __expf((idf ? idf[c] : 1.0f) * hp[(5 + c) * hv.sc] - m);
      se = wave_sum(se);
      const float inv = 1.0f / se;
      for (int c = lane; c < C; c += WAVE) op[5 + c] = __expf((idf ? idf[c] : 1.0f) * hp[(5 + c) * hv.sc] - m) * inv;
    } else {
      for (int c = lane; c < C; c += WAVE) op[5 + c] = 1.0f / (1.0f + __expf(-(idf ? idf[c] : 1.0f) * hp[(5 + c) * hv.sc]));
    }
  }
}

// Channels-last fast path of the decode (engine-native NHWC heads, sc == 1): a workgroup stages DEC_PIX pixels x
// (na*attrs) floats in LDS with coalesced 16-byte loads, one thread per (pixel, anchor) row does the box decode +
// sigmoid + softmax in place (row stride 85 floats: odd -> conflict-free), and the [rows,attrs] output — contiguous in
// the reference's flattened order — is written back with coalesced stores.  Optionally emits score = conf*max(cls) and
// the arg-max class (test_one_epoch.py:25,35) so the candidate filter never re-reads the 85-wide rows.
#define DEC_PIX 64
__global__ __launch_bounds__(256) void yolo_decode_cl_kernel(mi355det_yolo_geom geom, mi355det_head_view hv, int scale, const float* __restrict__ idf,
                                                             int softmax_cls, float* __restrict__ out, float* __restrict__ score_out,
                                                             int* __restrict__ label_out, int n_total) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ Geom g;
  load_geom(g, geom);
  const int W = g.grid[scale], hw = W * W, na = g.na, attrs = g.attrs, C = g.num_classes;
  const int chans = na * attrs;                       // floats used per pixel
  const int pitch = (int)hv.sp;                       // floats per pixel in memory (>= chans)
  const int b = blockIdx.y, pix0 = blockIdx.x * DEC_PIX, npix = min(DEC_PIX, hw - pix0);
  const float* src = (const float*)hv.ptr + (long long)b * hv.sb + (long long)pix0 * pitch;
  const int nfl = npix * pitch;
  if ((pitch & 3) == 0 && ((size_t)src & 15) == 0) {
    for (int i = threadIdx.x * 4; i < nfl; i += 256 * 4) *(float4*)(tile + i) = *(const float4*)(src + i);
  } else {
    for (int i = threadIdx.x; i < nfl; i += 256) tile[i] = src[i];
  }
  __syncthreads();
  const float gridf = (float)W, stride = g.img_size / gridf;
  for (int r = threadIdx.x; r < npix * na; r += 256) {
    const int pl = r / na, a = r - pl * na, pix = pix0 + pl;
    float* row = tile + pl * pitch + a * attrs;
    const int y = pix / W, x = pix - y * W;
    const float cx = ((float)x + 0.5f) / gridf, cy = ((float)y + 0.5f) / gridf;
    row[0] = (1.0f / (1.0f + __expf(-row[0])) + cx * gridf - 0.5f) * stride;
    row[1] = (1.0f / (1.0f + __expf(-row[1])) + cy * gridf - 0.5f) * stride;
    row[2] = __expf(row[2]) * g.aw[scale][a] * gridf * stride;
    row[3] = __expf(row[3]) * g.ah[scale][a] * gridf * stride;
    const float conf = 1.0f / (1.0f + __expf(-row[4]));
    row[4] = conf;
    float best = -INFINITY;
    int arg = 0;
    if (softmax_cls) {
      float m = -INFINITY;
      for (int c = 0; c < C; ++c) {
        const float z = (idf ? idf[c] : 1.0f) * row[5 + c];
        row[5 + c] = z;
        m = fmaxf(m, z);
      }
      float se = 0.f;
      for (int c = 0; c < C; ++c) {
        const float e = __expf(row[5 + c] - m);
        row[5 + c] = e;
        se += e;
      }
      const float inv = 1.0f / se;
      for (int c = 0; c < C; ++c) {
        const float pc = row[5 + c] * inv;
        row[5 + c] = pc;
        if (pc > best) {
          best = pc;
          arg = c;
        }
      }
    } else {
      for (int c = 0; c < C; ++c) {
        const float pc = 1.0f / (1.0f + __expf(-(idf ? idf[c] : 1.0f) * row[5 + c]));
        row[5 + c] = pc;
        if (pc > best) {
          best = pc;
          arg = c;
        }
      }
    }
    if (score_out) {
      const long long n = (long long)b * n_total + g.off[scale] + (long long)pix * na + a;
      score_out[n] = conf * best;
      label_out[n] = arg;
    }
  }
  __syncthreads();
  float* dst = out + ((long long)b * n_total + g.off[scale] + (long long)pix0 * na) * attrs;
  const int nout = npix * chans;
  for (int j = threadIdx.x; j < nout; j += 256) {
    const int pl = j / chans, rem = j - pl * chans;
    dst[j] = tile[pl * pitch + rem];
  }
}


// Channels-last decode, wave-per-pixel form (pixels with up to 64*KCH channels: COCO 3 x 85 = 255 -> KCH 4).  The LDS-tile form above
// gives each (pixel, anchor) row to ONE thread, which walks the 80 classes three times through LDS: 192 busy threads per workgroup and
// long dependent chains (1.5 TB/s, 19 % of the HBM roof).  Here a wave owns a pixel: lane l holds channels l, l+64, ..; every load and
// store instruction of the wave is one contiguous 256-byte segment; softmax max / first-argmax / sum per anchor are masked wave reductions;
// the largest class probability of a softmax row is exactly 1 * (1/sum) (its exponent is 0), so score and label need no second pass.
// Full-wave reductions on the DPP path (no LDS crossbar: a __shfl_xor butterfly is six dependent ds_bpermute round trips, ~600 cycles; the
// decode needs nine reductions per pixel).  Rows of 16 lanes by quad_perm / row_ror, then row_bcast15 / row_bcast31 carry the row totals
// to lane 63, which is read back as a scalar.  `idn` is the operation's identity (lanes a DPP step does not write keep it).
#define DPP_STEP(T, v, idn, op, ctrl, rmask)                                                                                                  \
  v = op(v, __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, idn), __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false)))
#define DPP_WAVE_REDUCE(T, v, idn, op)                                                                                                         \
  do {                                                                                                                                         \
    DPP_STEP(T, v, idn, op, 0xB1, 0xF);  /* quad_perm [1,0,3,2] */                                                                             \
    DPP_STEP(T, v, idn, op, 0x4E, 0xF);  /* quad_perm [2,3,0,1] */                                                                             \
    DPP_STEP(T, v, idn, op, 0x124, 0xF); /* row_ror:4 */                                                                                       \
    DPP_STEP(T, v, idn, op, 0x128, 0xF); /* row_ror:8  -> every lane: its row's total */                                                      \
    DPP_STEP(T, v, idn, op, 0x142, 0xA); /* row_bcast15: rows 1, 3 += last lane of rows 0, 2 */                                                \
    DPP_STEP(T, v, idn, op, 0x143, 0xC); /* row_bcast31: rows 2, 3 += lane 31 -> lane 63 holds the wave's total */                             \
  } while (0)
__device__ __forceinline__ float op_max_f(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float op_add_f(float a, float b) { return a + b; }
__device__ __forceinline__ int op_min_i(int a, int b) { return min(a, b); }
__device__ __forceinline__ float wave_max_dpp(float v) {
  const float idn = -INFINITY;
  DPP_WAVE_REDUCE(float, v, idn, op_max_f);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  const float idn = 0.0f;
  DPP_WAVE_REDUCE(float, v, idn, op_add_f);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_min_i(int v) {
  const int idn = 0x7FFFFFFF;
  DPP_WAVE_REDUCE(int, v, idn, op_min_i);
  return __builtin_amdgcn_readlane(v, 63);
}

template <int KCH>
__global__ __launch_bounds__(256) void yolo_decode_px_kernel(mi355det_yolo_geom geom, mi355det_head_view hv, int scale, const float* __restrict__ idf,
                                                             int softmax_cls, float* __restrict__ out, float* __restrict__ score_out,
                                                             int* __restrict__ label_out, int n_total, int bs) {
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const int W = geom.grid[scale], hw = W * W, na = geom.na, C = geom.num_classes, attrs = C + 5, chans = na * attrs;
  const int pitch = (int)hv.sp, off_scale = geom.off[scale];
  const float gridf = (float)W, stride = geom.img_size / gridf;
  // the role of this lane's channels is the same for every pixel
  int an[KCH], at[KCH];
  float mul[KCH], anc[KCH];
  bool live[KCH];
#pragma unroll
  for (int k = 0; k < KCH; ++k) {
    const int c = lane + 64 * k;
    live[k] = c < chans;
    an[k] = live[k] ? c / attrs : 0;
    at[k] = live[k] ? c - an[k] * attrs : 0;
    mul[k] = (live[k] && at[k] >= 5 && idf) ? idf[at[k] - 5] : 1.0f;
    anc[k] = at[k] == 2 ? geom.anchor_w[scale][an[k]] : geom.anchor_h[scale][an[k]];
  }
  const unsigned npix = (unsigned)bs * (unsigned)hw;                     // < 2^31 (checked by the host)
  const unsigned pstep = gridDim.x * 4;
  unsigned p = blockIdx.x * 4 + wid;
  float vn[KCH];                                                         // the NEXT pixel's channels: its loads fly while this pixel is decoded
  auto load_px = [&](unsigned q) {
    const unsigned bq = q / (unsigned)hw, pq = q - bq * (unsigned)hw;
    const float* src = (const float*)hv.ptr + (long long)bq * hv.sb + (long long)pq * pitch;
#pragma unroll
    for (int k = 0; k < KCH; ++k) vn[k] = (live[k] && q < npix) ? src[lane + 64 * k] : 0.0f;
  };
  load_px(p);
  for (; p < npix; p += pstep) {
    const int b = (int)(p / (unsigned)hw), pix = (int)(p - (unsigned)b * (unsigned)hw);
    const int y = pix / W, x = pix - y * W;
    float v[KCH], o[KCH];
#pragma unroll
    for (int k = 0; k < KCH; ++k) v[k] = vn[k];
    load_px(p + pstep);
    const float cx = ((float)x + 0.5f) / gridf, cy = ((float)y + 0.5f) / gridf;
#pragma unroll
    for (int k = 0; k < KCH; ++k) {
      const int t = at[k];
      if (t >= 5) {
        o[k] = mul[k] * v[k];                                    // class logit (tf-idf scaled), finished below
      } else if (t == 2 || t == 3) {
        o[k] = __expf(v[k]) * anc[k] * gridf * stride;
      } else {
        const float sg = 1.0f / (1.0f + __expf(-v[k]));
        o[k] = t == 4 ? sg : (sg + (t == 0 ? cx : cy) * gridf - 0.5f) * stride;
      }
    }
    float my_score = 0.0f;
    int my_label = 0;
    for (int a = 0; a < na; ++a) {
      float m = -INFINITY;
#pragma unroll
      for (int k = 0; k < KCH; ++k)
        if (live[k] && an[k] == a && at[k] >= 5) m = fmaxf(m, o[k]);
      m = wave_max_dpp(m);
      int idx = 0x7FFFFFFF;
#pragma unroll
      for (int k = 0; k < KCH; ++k)
        if (live[k] && an[k] == a && at[k] >= 5 && o[k] == m) idx = min(idx, at[k] - 5);
      idx = wave_min_i(idx);                                       // first maximum, as torch.max
      float best;
      if (softmax_cls) {
        float se = 0.0f;
#pragma unroll
        for (int k = 0; k < KCH; ++k)
          if (live[k] && an[k] == a && at[k] >= 5) {
            o[k] = __expf(o[k] - m);
            se += o[k];
          }
        se = wave_sum_dpp(se);
        const float inv = 1.0f / se;
#pragma unroll
        for (int k = 0; k < KCH; ++k)
          if (live[k] && an[k] == a && at[k] >= 5) o[k] *= inv;
        best = 1.0f * inv;                                         // exp(m - m) * inv
      } else {
#pragma unroll
        for (int k = 0; k < KCH; ++k)
          if (live[k] && an[k] == a && at[k] >= 5) o[k] = 1.0f / (1.0f + __expf(-o[k]));
        best = 1.0f / (1.0f + __expf(-m));
      }
      if (score_out) {
        const int c4 = a * attrs + 4;
        float cand = 0.0f;
#pragma unroll
        for (int k = 0; k < KCH; ++k) cand = (c4 >> 6) == k ? o[k] : cand;
        const float conf = __shfl(cand, c4 & 63, WAVE);
        if (lane == a) {
          my_score = conf * best;
          my_label = idx;
        }
      }
    }
    const long long row0 = (long long)b * n_total + off_scale + (long long)pix * na;
    float* dst = out + row0 * attrs;
#pragma unroll
    for (int k = 0; k < KCH; ++k)
      if (live[k]) dst[lane + 64 * k] = o[k];
    if (score_out && lane < na) {
      score_out[row0 + lane] = my_score;
      label_out[row0 + lane] = my_label;
    }
  }
}


// Channels-last decode, half-wave-per-row form (attrs <= 32 * E): a wave decodes TWO (pixel, anchor) rows, lane i of each half holds attributes
// i, i+32, .. of its row.  Every attribute costs one exponential (the sign / offset of its argument depends on its role), the softmax maximum,
// first arg-max and sum are 32-lane DPP reductions, and the role of a lane never changes, so nothing in the loop branches.  (The
// wave-per-pixel form above needed per-anchor masked loops: ~550 vector instructions per pixel, issue-bound at 1.8 TB/s.)
__device__ __forceinline__ float half_bcast(float v, int half) {      // totals sit in lanes 31 / 63 after the row_bcast15 step
  const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
  const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  return half ? hi : lo;
}
#define DPP_HALF_REDUCE(T, v, idn, op)                                                                                                         \
  do {                                                                                                                                         \
    DPP_STEP(T, v, idn, op, 0xB1, 0xF);                                                                                                        \
    DPP_STEP(T, v, idn, op, 0x4E, 0xF);                                                                                                        \
    DPP_STEP(T, v, idn, op, 0x124, 0xF);                                                                                                       \
    DPP_STEP(T, v, idn, op, 0x128, 0xF);                                                                                                       \
    DPP_STEP(T, v, idn, op, 0x142, 0xA); /* lanes 31 / 63: totals of lanes 0-31 / 32-63 */                                                     \
  } while (0)

struct DecDiv {
  unsigned mul, shift;
};
static DecDiv make_decdiv(unsigned d) {
  DecDiv f;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.shift = l;
  f.mul = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ unsigned ddiv(unsigned n, const DecDiv f) { return (__umulhi(f.mul, n) + n) >> f.shift; }    // n < 2^31

template <int E>
__global__ __launch_bounds__(256) void yolo_decode_row_kernel(mi355det_yolo_geom geom, mi355det_head_view hv, int scale, const float* __restrict__ idf,
                                                              int softmax_cls, float* __restrict__ out, float* __restrict__ score_out,
                                                              int* __restrict__ label_out, int n_total, int rows_per_image, DecDiv dna, DecDiv dW) {
  __shared__ float s_anc[2][MI355DET_MAX_ANCHORS];
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const int half = lane >> 5, li = lane & 31;
  const int W = geom.grid[scale], na = geom.na, C = geom.num_classes, attrs = C + 5;
  if (threadIdx.x < MI355DET_MAX_ANCHORS) {
    s_anc[0][threadIdx.x] = geom.anchor_w[scale][threadIdx.x];
    s_anc[1][threadIdx.x] = geom.anchor_h[scale][threadIdx.x];
  }
  __syncthreads();
  const int pitch = (int)hv.sp, off_scale = geom.off[scale];
  const float gridf = (float)W, stride = geom.img_size / gridf;
  const int b = blockIdx.y;
  const float* src_img = (const float*)hv.ptr + (long long)b * hv.sb;
  float* out_img = out + ((long long)b * n_total + off_scale) * attrs;
  // lane roles never change: element j of lane li is attribute li + 32 j; only element 0 can be a box attribute
  float mul[E];
  bool live[E], cls[E];
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int t = li + 32 * j;
    live[j] = t < attrs;
    cls[j] = live[j] && t >= 5;
    mul[j] = (cls[j] && idf) ? idf[t - 5] : 1.0f;
  }
  const bool is_wh = li == 2 || li == 3, is_x = li == 0, is_conf = li == 4;
  const float sign0 = (cls[0] || !is_wh) ? -1.0f : 1.0f;           // exp argument of element 0 when it is a box attribute: v for w, h; -v otherwise
  const unsigned nrows = (unsigned)rows_per_image;
  const unsigned rstep = gridDim.x * 8;
  unsigned r = (blockIdx.x * 4 + wid) * 2 + half;
  float vn[E];
  auto load_row = [&](unsigned q) {
    const unsigned qq = min(q, nrows - 1);                             // past the end: a valid (unused) row, no branch
    const unsigned pix = ddiv(qq, dna), a = qq - pix * (unsigned)na;
    const float* src = src_img + (long long)pix * pitch + a * attrs;
#pragma unroll
    for (int j = 0; j < E; ++j) vn[j] = src[min(li + 32 * j, attrs - 1)];
  };
  load_row(r);
  for (unsigned r0 = (blockIdx.x * 4 + wid) * 2; r0 < nrows; r0 += rstep, r += rstep) {     // r0: wave-uniform loop bound
    const bool row_ok = r < nrows;
    const unsigned rr = min(r, nrows - 1);
    const unsigned pix = ddiv(rr, dna), a = rr - pix * (unsigned)na;
    const unsigned y = ddiv(pix, dW), x = pix - y * (unsigned)W;
    float v[E];
#pragma unroll
    for (int j = 0; j < E; ++j) v[j] = vn[j];
    load_row(r + rstep);
    // ---- class logits, their maximum and first arg-max over the row (32-lane reductions)
    float z[E];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      z[j] = mul[j] * v[j];
      m = fmaxf(m, cls[j] ? z[j] : -INFINITY);
    }
    {
      const float idn = -INFINITY;
      DPP_HALF_REDUCE(float, m, idn, op_max_f);
    }
    m = half_bcast(m, half);
    int idx = 0x7FFFFFFF;
#pragma unroll
    for (int j = 0; j < E; ++j) idx = min(idx, (cls[j] && z[j] == m) ? li + 32 * j - 5 : 0x7FFFFFFF);
    {
      const int idn = 0x7FFFFFFF;
      DPP_HALF_REDUCE(int, idx, idn, op_min_i);
    }
    const int idx_lo = __builtin_amdgcn_readlane(idx, 31), idx_hi = __builtin_amdgcn_readlane(idx, 63);
    // ---- ONE exponential per attribute: class exp(z - m) (softmax) or exp(-z) (sigmoid); w, h exp(v); x, y, conf exp(-v)
    float e[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const float carg = softmax_cls ? z[j] - m : -z[j];
      const float arg = (j == 0 && !cls[0]) ? sign0 * v[0] : carg;
      e[j] = __expf(arg);
    }
    float inv = 1.0f;
    if (softmax_cls) {                                             // wave-uniform
      float se = 0.0f;
#pragma unroll
      for (int j = 0; j < E; ++j) se += cls[j] ? e[j] : 0.0f;
      {
        const float idn = 0.0f;
        DPP_HALF_REDUCE(float, se, idn, op_add_f);
      }
      inv = __builtin_amdgcn_rcpf(half_bcast(se, half));
    }
    float o[E];
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const float sg = __builtin_amdgcn_rcpf(1.0f + e[j]);
      o[j] = softmax_cls ? e[j] * inv : sg;                        // class attribute
      if (j == 0) {                                                // box attributes live in element 0 only: selects, no branches
        const float cgrid = ((float)(is_x ? x : y) + 0.5f) / gridf;
        const float xy = (sg + cgrid * gridf - 0.5f) * stride;
        const float wh = e[0] * s_anc[li == 3 ? 1 : 0][a] * gridf * stride;
        const float box = is_wh ? wh : (is_conf ? sg : xy);
        o[0] = cls[0] ? o[0] : box;
      }
    }
    float* dst = out_img + (long long)rr * attrs;
    if (row_ok) {
#pragma unroll
      for (int j = 0; j < E; ++j)
        if (live[j]) dst[li + 32 * j] = o[j];
    }
    if (score_out) {                                               // wave-uniform
      // conf sits in element 0 of lane 4 / 36; the largest class probability of a softmax row is exp(m - m) * inv
      const float conf_lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o[0]), 4));
      const float conf_hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o[0]), 36));
      const float best = softmax_cls ? 1.0f * inv : __builtin_amdgcn_rcpf(1.0f + __expf(-m));
      if (li == 0 && row_ok) {
        const long long row = (long long)b * n_total + off_scale + rr;
        score_out[row] = (half ? conf_hi : conf_lo) * best;
        label_out[row] = half ? idx_hi : idx_lo;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// test_one_epoch.py:24-35 — score = conf*max(cls) (first maximum, as torch.max), one wave per row;
// then per image an ORDERED compaction (the reference's boolean-mask order) of rows with score>thr.
__global__ __launch_bounds__(DEC_WAVES* WAVE) void yolo_score_kernel(const float* __restrict__ pred, long long rows, int attrs,
                                                                      float* __restrict__ score, int* __restrict__ label) {
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const int C = attrs - 5;
  for (long long row = (long long)blockIdx.x * DEC_WAVES + wid; row < rows; row += (long long)gridDim.x * DEC_WAVES) {
    const float* p = pred + row * attrs;
    float best = -INFINITY;
    int arg = 0x7FFFFFFF;
    for (int c = lane; c < C; c += WAVE) {
      const float v = p[5 + c];
      if (v > best) {
        best = v;
        arg = c;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, WAVE);
      const int oa = __shfl_xor(arg, o, WAVE);
      if (ob > best || (ob == best && oa < arg)) {
        best = ob;
        arg = oa;
      }
    }
    if (lane == 0) {
      score[row] = p[4] * best;
      label[row] = arg;
    }
  }
}

#define CAND_THREADS 1024
__global__ __launch_bounds__(CAND_THREADS) void yolo_candidates_kernel(const float* __restrict__ pred, const float* __restrict__ score,
                                                                        const int* __restrict__ label, long long n, int attrs,
                                                                        float conf_thr, float* __restrict__ cand,
                                                                        int* __restrict__ count, int max_cand) {
  __shared__ int wsum[CAND_THREADS / WAVE];
  __shared__ int base;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (long long n0 = 0; n0 < n; n0 += CAND_THREADS) {
    const long long i = n0 + threadIdx.x;
    const float sc = i < n ? score[(size_t)b * n + i] : 0.0f;
    const bool pass = i < n && sc > conf_thr;
    const unsigned long long bal = __ballot(pass);
    const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wid] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wid; ++w) off += wsum[w];
    if (pass) {
      const int dst = off + wpre;
      if (dst < max_cand) {
        const float* p = pred + ((size_t)b * n + i) * attrs;
        float* o = cand + ((size_t)b * max_cand + dst) * 6;
        o[0] = p[0] - p[2] / 2.0f;   // helper.get_abs_coord
        o[1] = p[1] - p[3] / 2.0f;
        o[2] = p[0] + p[2] / 2.0f;
        o[3] = p[1] + p[3] / 2.0f;
        o[4] = sc;
        o[5] = (float)label[(size_t)b * n + i];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < CAND_THREADS / WAVE; ++w) t += wsum[w];
      base += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) count[b] = base;
}

int check_geom(const mi355det_yolo_geom* g) {
  if (!g) return fail(MI355DET_EINVAL, "%s: null geometry", "yolo");
  if (g->num_scales < 1 || g->num_scales > MI355DET_MAX_SCALES || g->na < 1 || g->na > MI355DET_MAX_ANCHORS || g->num_classes < 1)
    return fail(MI355DET_EINVAL, "%s: unsupported geometry (scales=%lld, anchors=%lld)", "yolo", g->num_scales, g->na);
  return 0;
}

}  // namespace

extern "C" {

int mi355det_bbox_iou(const float* bb1, const float* bb2, float* out, int64_t m, int64_t n, int iou_type, int xcycwh, int paired,
                      void* stream) {
  if (m < 0 || n < 0 || iou_type < 0 || iou_type > 3) return fail(MI355DET_EINVAL, "%s: bad shape/iou_type", "bbox_iou");
  const long long total = paired ? n : m * n;
  if (total == 0) return 0;
  const int blocks = (int)min((long long)2048, (total + 255) / 256);
  hipLaunchKernelGGL(bbox_iou_kernel, dim3(blocks), dim3(256), 0, S(stream), bb1, bb2, out, (long long)m, (long long)n, iou_type, xcycwh,
                     paired);
  return check_launch("bbox_iou");
}

int mi355det_yolo_assign(const mi355det_yolo_geom* geom, const float* gt_box, const int32_t* gt_off, int32_t bs, int32_t num_gt,
                         int32_t max_gt_per_img, uint64_t* best_key, int64_t* obj_idx, float* tgt, uint8_t* noobj, void* stream) {
  if (int e = check_geom(geom)) return e;
  (void)max_gt_per_img;
  const int N = geom->off[geom->num_scales];
  if (bs <= 0 || num_gt < 0) return fail(MI355DET_EINVAL, "%s: bad batch", "yolo_assign");
  if (num_gt > 0) {
    if (hipMemsetAsync(best_key, 0, sizeof(uint64_t) * (size_t)num_gt, S(stream)) != hipSuccess)
      return fail(MI355DET_ELAUNCH, "%s: memset failed", "yolo_assign");
  }
  hipLaunchKernelGGL(yolo_assign_kernel, dim3((N + ASSIGN_THREADS - 1) / ASSIGN_THREADS, bs), dim3(ASSIGN_THREADS), 0, S(stream), *geom,
                     gt_box, gt_off, (unsigned long long*)best_key, noobj, N);
  if (num_gt > 0)
    hipLaunchKernelGGL(yolo_targets_kernel, dim3((num_gt + 255) / 256), dim3(256), 0, S(stream), *geom, gt_box, gt_off,
                       (const unsigned long long*)best_key, (long long*)obj_idx, tgt, noobj, bs, N);
  return check_launch("yolo_assign");
}

size_t mi355det_yolo_loss_workspace(int32_t bs, int64_t n_anchors) {
  const size_t dense_blocks = (size_t)bs * ((n_anchors + DENSE_THREADS - 1) / DENSE_THREADS);
  return (dense_blocks * 3 + (size_t)bs * POS_NP + 4) * sizeof(float);      // + the no-object element count of reduction='mean'
}

int mi355det_yolo_loss(const mi355det_yolo_geom* geom, const mi355det_yolo_loss_cfg* cfg, const mi355det_head_view* heads,
                       const mi355det_head_view* grads, const int32_t* gt_off, const int64_t* gt_label, const int64_t* obj_idx,
                       const float* tgt, const uint8_t* noobj, const float* idf, int32_t bs, int32_t num_gt, void* workspace,
                       size_t workspace_bytes, float* out12, void* stream) {
  if (int e = check_geom(geom)) return e;
  if (!cfg || !heads || bs <= 0 || num_gt <= 0) return fail(MI355DET_EINVAL, "%s: bad arguments (num_gt must be > 0)", "yolo_loss");
  const int N = geom->off[geom->num_scales];
  if (workspace_bytes < mi355det_yolo_loss_workspace(bs, N)) return fail(MI355DET_EWORKSPACE, "%s: workspace too small", "yolo_loss");
  Views hv, gv;
  for (int k = 0; k < MI355DET_MAX_SCALES; ++k) {
    hv.h[k] = k < geom->num_scales ? heads[k] : mi355det_head_view{nullptr, 0, 0, 0};
    gv.h[k] = (grads && k < geom->num_scales) ? grads[k] : mi355det_head_view{nullptr, 0, 0, 0};
  }
  const int has_grad = grads != nullptr;
  const int dblocks = (N + DENSE_THREADS - 1) / DENSE_THREADS;
  float* dense_p = (float*)workspace;
  float* pos_p = dense_p + (size_t)bs * dblocks * 3;
  if (cfg->class_loss < 0 || cfg->class_loss > 2) return fail(MI355DET_EINVAL, "%s: class_loss must be 0 (bce), 1 (ce) or 2 (eql)", "yolo_loss");
  if (cfg->class_loss == 2 && !cfg->eq_mask) return fail(MI355DET_EINVAL, "%s: class_loss 2 (EQLoss) needs eq_mask", "yolo_loss");
  const bool mean = cfg->reduction_mean != 0;
  int* count_p = (int*)(pos_p + (size_t)bs * POS_NP);
  if (mean) hipLaunchKernelGGL(yolo_noobj_count_kernel, dim3(1), dim3(1024), 0, S(stream), noobj, (long long)bs * N, count_p);
  // 'sum': every gradient / sum(M); 'mean': each term over its own element count (the kernels apply the rest)
  const float inv_ng = 1.0f / (float)num_gt;
  const float gscale = cfg->lambda_no_conf * cfg->grad_scale * (mean ? 1.0f : inv_ng);
  const int* cnt = mean ? count_p : nullptr;
  if (cfg->grad_is_bf16 < 0 || cfg->grad_is_bf16 > 2) return fail(MI355DET_EINVAL, "%s: grad_is_bf16 must be 0 (fp32), 1 (bf16) or 2 (fp16)", "yolo_loss");
#define YOLO_LOSS_LAUNCH(GF)                                                                                                                  \
  do {                                                                                                                                        \
    hipLaunchKernelGGL(yolo_noobj_kernel<GF>, dim3(dblocks, bs), dim3(DENSE_THREADS), 0, S(stream), *geom, hv, gv, has_grad, noobj,           \
                       cfg->alpha, cfg->gamma, gscale, dense_p, N, cnt);                                                                      \
    hipLaunchKernelGGL(yolo_pos_kernel<GF>, dim3(bs), dim3(WAVE), 0, S(stream), *geom, *cfg, hv, gv, has_grad, gt_off,                        \
                       (const long long*)gt_label, (const long long*)obj_idx, tgt, idf, inv_ng, bs, pos_p);                                   \
  } while (0)
  if (cfg->grad_is_bf16 == 1) YOLO_LOSS_LAUNCH(1);
  else if (cfg->grad_is_bf16 == 2) YOLO_LOSS_LAUNCH(2);
  else YOLO_LOSS_LAUNCH(0);
#undef YOLO_LOSS_LAUNCH
  hipLaunchKernelGGL(yolo_reduce_kernel, dim3(1), dim3(256), 0, S(stream), dense_p, bs * dblocks, pos_p, bs, cfg->lambda_no_conf,
                     (float)num_gt, geom->num_classes, cfg->class_loss, cfg->reduction_mean, out12);
  return check_launch("yolo_loss");
}

int mi355det_yolo_decode(const mi355det_yolo_geom* geom, const mi355det_head_view* heads, const float* idf, int32_t bs, int softmax_cls,
                         float* out, float* score_out, int32_t* label_out, void* stream) {
  if (int e = check_geom(geom)) return e;
  if (!heads || bs <= 0) return fail(MI355DET_EINVAL, "%s: bad arguments", "yolo_decode");
  const int N = geom->off[geom->num_scales];
  bool channels_last = true;
  for (int k = 0; k < geom->num_scales; ++k)
    channels_last = channels_last && heads[k].sc == 1 && heads[k].sp >= geom->na * (geom->num_classes + 5) && heads[k].sp <= 4096 / 4 * 4;
  const int chans = geom->na * (geom->num_classes + 5);
  if (channels_last && geom->num_classes + 5 <= 128) {
    // half a wave per (pixel, anchor) row
    const int attrs = geom->num_classes + 5;
    for (int k = 0; k < geom->num_scales; ++k) {
      const long long nrows = (long long)geom->grid[k] * geom->grid[k] * geom->na;      // rows of one image at this scale
      if (nrows >= (1ll << 30)) return fail(MI355DET_EINVAL, "%s: too many rows", "yolo_decode");
      const int blocks = (int)max((long long)1, min((long long)(256 * 8 + bs - 1) / bs, (nrows + 7) / 8));
      const DecDiv dna = make_decdiv((unsigned)geom->na), dW = make_decdiv((unsigned)geom->grid[k]);
#define LAUNCH_ROW(E)                                                                                                                          \
  hipLaunchKernelGGL(yolo_decode_row_kernel<E>, dim3(blocks, bs), dim3(256), 0, S(stream), *geom, heads[k], k, idf, softmax_cls, out, score_out, \
                     (int*)label_out, N, (int)nrows, dna, dW)
      if (attrs <= 32) LAUNCH_ROW(1);
      else if (attrs <= 64) LAUNCH_ROW(2);
      else if (attrs <= 96) LAUNCH_ROW(3);
      else LAUNCH_ROW(4);
#undef LAUNCH_ROW
    }
    return check_launch("yolo_decode");
  }
  if (channels_last && chans <= 512 && geom->na <= 64) {
    // one wave per pixel: lanes over the pixel's channels
    for (int k = 0; k < geom->num_scales; ++k) {
      const long long npix = (long long)bs * geom->grid[k] * geom->grid[k];
      if (npix >= (1ll << 31)) return fail(MI355DET_EINVAL, "%s: too many pixels", "yolo_decode");
      const int blocks = (int)min((long long)256 * 8, (npix + 3) / 4);
#define LAUNCH_PX(KCH)                                                                                                                         \
  hipLaunchKernelGGL(yolo_decode_px_kernel<KCH>, dim3(blocks), dim3(256), 0, S(stream), *geom, heads[k], k, idf, softmax_cls, out, score_out, \
                     (int*)label_out, N, bs)
      if (chans <= 64) LAUNCH_PX(1);
      else if (chans <= 128) LAUNCH_PX(2);
      else if (chans <= 256) LAUNCH_PX(4);
      else LAUNCH_PX(8);
#undef LAUNCH_PX
    }
    return check_launch("yolo_decode");
  }
  if (channels_last) {
    for (int k = 0; k < geom->num_scales; ++k) {
      const int hw = geom->grid[k] * geom->grid[k];
      const int lds = DEC_PIX * (int)heads[k].sp * 4;
      if (lds > 160 * 1024) return fail(MI355DET_EINVAL, "%s: pixel pitch too large for the LDS tile", "yolo_decode");
      (void)hipFuncSetAttribute((const void*)yolo_decode_cl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(yolo_decode_cl_kernel, dim3((hw + DEC_PIX - 1) / DEC_PIX, bs), dim3(256), lds, S(stream), *geom, heads[k], k, idf,
                         softmax_cls, out, score_out, (int*)label_out, N);
    }
    return check_launch("yolo_decode");
  }
  if (score_out) return fail(MI355DET_EINVAL, "%s: fused score output needs channels-last heads", "yolo_decode");
  Views hv;
  for (int k = 0; k < MI355DET_MAX_SCALES; ++k) hv.h[k] = k < geom->num_scales ? heads[k] : mi355det_head_view{nullptr, 0, 0, 0};
  const long long rows = (long long)bs * N;
  const int blocks = (int)min((long long)256 * 16, (rows + DEC_WAVES - 1) / DEC_WAVES);
  hipLaunchKernelGGL(yolo_decode_kernel, dim3(blocks), dim3(DEC_WAVES * WAVE), 0, S(stream), *geom, hv, idf, softmax_cls, out, N, rows);
  return check_launch("yolo_decode");
}

size_t mi355det_yolo_candidates_workspace(int32_t bs, int64_t n) { return (size_t)bs * (size_t)n * 8; }

int mi355det_yolo_candidates(const float* pred, const float* score_in, const int32_t* label_in, int32_t bs, int64_t n, int32_t attrs,
                             float conf_thr, float* cand, int32_t* count, int32_t max_cand, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (bs <= 0 || n <= 0 || attrs < 6 || max_cand <= 0) return fail(MI355DET_EINVAL, "%s: bad arguments", "yolo_candidates");
  const float* score = score_in;
  const int* label = (const int*)label_in;
  if (!score_in || !label_in) {
    if (workspace_bytes < mi355det_yolo_candidates_workspace(bs, n)) return fail(MI355DET_EWORKSPACE, "%s: workspace too small", "yolo_candidates");
    float* sc = (float*)workspace;
    int* lb = (int*)(sc + (size_t)bs * n);
    const long long rows = (long long)bs * n;
    const int blocks = (int)min((long long)256 * 16, (rows + DEC_WAVES - 1) / DEC_WAVES);
    hipLaunchKernelGGL(yolo_score_kernel, dim3(blocks), dim3(DEC_WAVES * WAVE), 0, S(stream), pred, rows, attrs, sc, lb);
    score = sc;
    label = lb;
  }
  hipLaunchKernelGGL(yolo_candidates_kernel, dim3(bs), dim3(CAND_THREADS), 0, S(stream), pred, score, label, (long long)n, attrs, conf_thr,
                     cand, count, max_cand);
  return check_launch("yolo_candidates");
}

}  // extern "C"
