/* CPU oracle, C / OpenMP form - TEST INFRASTRUCTURE ONLY (imported by tests/ and bench.py's cpu_baseline leg, never by the product path).
 *
 * Plain-C restatement of the box operations of the torchvision_models path, float32 arithmetic in the reference's operation order:
 *   box_iou, nms, batched_nms              torchvision.ops (not vendored in /root/reference: published semantics, SURVEY Appendix B;
 *                                          parity unpinned beyond the reference-owned callers, see oracle/README.md)
 *   matcher                                tvision/_utils.py:271-344  Matcher.__call__ / set_low_quality_matches_
 *   encode_boxes / decode_boxes            tvision/_utils.py:79-125,152-223  BoxCoder
 *   sigmoid_focal_loss_sum                 torchvision.ops.sigmoid_focal_loss as called at tvision/retinanet.py:137-141 (reduction 'sum')
 * Pinned by tests/test_oracle_c.py against the numpy oracle (oracle/tv_oracle.py) and, through it, against the reference fixtures
 * tests/golden/g5_7_tvision.npz (Matcher, BoxCoder outputs of the reference's own code).
 * Build: gcc -O2 -fopenmp -shared -fPIC -ffp-contract=off -o oracle/_build/libbox_oracle.so oracle/c/box_oracle.c -lm   (oracle/c/build.sh)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float box_area(const float* b) { return (b[2] - b[0]) * (b[3] - b[1]); }
static inline float iou1(const float* a, const float* b) {
  const float ltx = a[0] > b[0] ? a[0] : b[0], lty = a[1] > b[1] ? a[1] : b[1];
  const float rbx = a[2] < b[2] ? a[2] : b[2], rby = a[3] < b[3] ? a[3] : b[3];
  float w = rbx - ltx, h = rby - lty;
  w = w > 0.f ? w : 0.f;
  h = h > 0.f ? h : 0.f;
  const float inter = w * h;
  return inter / (box_area(a) + box_area(b) - inter);
}

/* out[m][n] = IoU(a[m], b[n]) */
void oracle_box_iou(const float* a, int64_t m, const float* b, int64_t n, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; ++i)
    for (int64_t j = 0; j < n; ++j) out[i * n + j] = iou1(a + 4 * i, b + 4 * j);
}

/* Matcher on the quality matrix q[m][n] -> matches[n] (gt index, -1 below low, -2 between), low-quality rescue as the reference:
 * every prediction attaining a gt's row maximum (ties included) gets its original argmax back */
void oracle_matcher(const float* q, int64_t m, int64_t n, float high, float low, int allow_low_quality, int64_t* matches) {
  int64_t* allm = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n; ++j) {
    float best = q[j];
    int64_t arg = 0;
    for (int64_t i = 1; i < m; ++i)
      if (q[i * n + j] > best) {      /* first maximum, like argmax */
        best = q[i * n + j];
        arg = i;
      }
    allm[j] = arg;
    matches[j] = best < low ? -1 : (best < high ? -2 : arg);
  }
  if (allow_low_quality) {
    for (int64_t i = 0; i < m; ++i) {
      float best = q[i * n];
      for (int64_t j = 1; j < n; ++j)
        if (q[i * n + j] > best) best = q[i * n + j];
      for (int64_t j = 0; j < n; ++j)
        if (q[i * n + j] == best) matches[j] = allm[j];
    }
  }
  free(allm);
}

/* greedy NMS: descending score, ties lower index first (stable); suppress IoU > thr; returns the number of kept indices */
typedef struct { float s; int64_t i; } sc_t;
static int cmp_desc(const void* a, const void* b) {
  const sc_t *x = (const sc_t*)a, *y = (const sc_t*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}
int64_t oracle_nms(const float* boxes, const float* scores, int64_t n, float thr, int64_t* keep) {
  sc_t* order = (sc_t*)malloc(sizeof(sc_t) * (size_t)(n > 0 ? n : 1));
  unsigned char* dead = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
  for (int64_t i = 0; i < n; ++i) {
    order[i].s = scores[i];
    order[i].i = i;
  }
  qsort(order, (size_t)n, sizeof(sc_t), cmp_desc);
  int64_t kept = 0;
  for (int64_t p = 0; p < n; ++p) {
    if (dead[p]) continue;
    const int64_t idx = order[p].i;
    keep[kept++] = idx;
#pragma omp parallel for schedule(static) if (n - p > 4096)
    for (int64_t r = p + 1; r < n; ++r)
      if (!dead[r] && iou1(boxes + 4 * idx, boxes + 4 * order[r].i) > thr) dead[r] = 1;
  }
  free(order);
  free(dead);
  return kept;
}
/* batched_nms: per-category offsets idxs * (max coordinate + 1), then nms */
int64_t oracle_batched_nms(const float* boxes, const float* scores, const int64_t* idxs, int64_t n, float thr, int64_t* keep) {
  if (n == 0) return 0;
  float mx = boxes[0];
  for (int64_t i = 1; i < 4 * n; ++i)
    if (boxes[i] > mx) mx = boxes[i];
  float* sh = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    const float off = (float)idxs[i] * (mx + 1.0f);
    for (int c = 0; c < 4; ++c) sh[4 * i + c] = boxes[4 * i + c] + off;
  }
  const int64_t k = oracle_nms(sh, scores, n, thr, keep);
  free(sh);
  return k;
}

/* BoxCoder.encode_single / decode_single with weights (wx, wy, ww, wh) */
void oracle_encode_boxes(const float* ref, const float* prop, int64_t n, const float* w, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float* r = ref + 4 * i;
    const float* p = prop + 4 * i;
    const float ew = p[2] - p[0], eh = p[3] - p[1], ecx = p[0] + 0.5f * ew, ecy = p[1] + 0.5f * eh;
    const float gw = r[2] - r[0], gh = r[3] - r[1], gcx = r[0] + 0.5f * gw, gcy = r[1] + 0.5f * gh;
    out[4 * i + 0] = w[0] * (gcx - ecx) / ew;
    out[4 * i + 1] = w[1] * (gcy - ecy) / eh;
    out[4 * i + 2] = w[2] * logf(gw / ew);
    out[4 * i + 3] = w[3] * logf(gh / eh);
  }
}
void oracle_decode_boxes(const float* codes, const float* boxes, int64_t n, const float* w, float clip, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float* b = boxes + 4 * i;
    const float* c = codes + 4 * i;
    const float bw = b[2] - b[0], bh = b[3] - b[1], cx = b[0] + 0.5f * bw, cy = b[1] + 0.5f * bh;
    const float dx = c[0] / w[0], dy = c[1] / w[1];
    float dw = c[2] / w[2], dh = c[3] / w[3];
    dw = dw < clip ? dw : clip;
    dh = dh < clip ? dh : clip;
    const float pcx = dx * bw + cx, pcy = dy * bh + cy, pw = expf(dw) * bw, ph = expf(dh) * bh;
    out[4 * i + 0] = pcx - 0.5f * pw;
    out[4 * i + 1] = pcy - 0.5f * ph;
    out[4 * i + 2] = pcx + 0.5f * pw;
    out[4 * i + 3] = pcy + 0.5f * ph;
  }
}

/* sum of the sigmoid focal loss over n elements (double accumulation, fixed order) and optionally d loss / d x */
double oracle_sigmoid_focal_loss_sum(const float* x, const float* t, int64_t n, float alpha, float gamma, float* grad) {
  double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
  for (int64_t i = 0; i < n; ++i) {
    const double xv = x[i], tv = t[i];
    const double p = 1.0 / (1.0 + exp(-xv));
    const double ce = (xv > 0 ? xv : 0) - xv * tv + log1p(exp(-fabs(xv)));
    const double p_t = p * tv + (1 - p) * (1 - tv), q = 1 - p_t;
    double loss = ce * pow(q, gamma);
    const double dpt = (2 * tv - 1) * p * (1 - p);
    double g = (p - tv) * pow(q, gamma) - ce * gamma * pow(q, gamma - 1) * dpt;
    if (alpha >= 0) {
      const double a_t = alpha * tv + (1 - alpha) * (1 - tv);
      loss *= a_t;
      g *= a_t;
    }
    total += (double)(float)loss;
    if (grad) grad[i] = (float)g;
  }
  return total;
}
