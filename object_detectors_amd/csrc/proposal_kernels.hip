// RegionProposalNetwork.filter_proposals for the whole batch in ONE host call (tvision/rpn.py:215-280 and the decode of :336-351):
// per-level top-k of the objectness logits, decode of the SELECTED anchors only, clip to the image, small-box / score filter as a mask,
// per-level NMS of every image side by side, the first post_nms_top_n survivors gathered into dense outputs.  The torch form of the
// same chain is ~60 small launches whose host time (1.7 ms at batch 4) sits between the network forward and the RoI branch with the
// device idle; here the host issues ~14 launches from C.  Latency-bound integer / gather work: nothing to tile.
#include "common.h"

using namespace mi355;

namespace {

constexpr int MAX_LEVELS = 8;

struct ProposalLevels {
  int nlev;
  int start[MAX_LEVELS];        // first anchor of the level in the concatenated [A] axis
  int k[MAX_LEVELS];            // min(pre_nms_top_n, anchors of the level)
  int koff[MAX_LEVELS + 1];     // prefix sums of k: column range of the level in the candidate axis [K]
  long long idx_off[MAX_LEVELS];   // byte offsets of the level's top-k outputs in the workspace: idx [N, k] int64 ...
  long long val_off[MAX_LEVELS];   // ... and val [N, k] float
  long long cnt_off[MAX_LEVELS];   // ... and the number of selected entries per image [N] int32 (< k only for rows with NaN / -inf logits)
};

// One thread per candidate (image, j).  Formulas in the order of ops.box_decode (BoxCoder.decode_single, tvision/_utils.py:196-232, weights 1)
// and clip_boxes_to_image / remove_small_boxes (rpn.py:263-270); -ffp-contract=off keeps them bit-equal to the unfused route.
__global__ __launch_bounds__(256) void rpn_select_kernel(const char* __restrict__ ws, ProposalLevels L, const float* __restrict__ deltas,
                                                         const float* __restrict__ anchors, const float* __restrict__ lim, int n_images,
                                                         long long A, float xform_clip, float min_size, float score_thresh,
                                                         float* __restrict__ boxes, float* __restrict__ masked, float* __restrict__ scores,
                                                         long long* __restrict__ lvl) {
  const int K = L.koff[L.nlev];
  const long long total = (long long)n_images * K;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int img = (int)(t / K), j = (int)(t - (long long)img * K);
    int l = 0;
#pragma unroll
    for (int q = 1; q < MAX_LEVELS; ++q)
      if (q < L.nlev && j >= L.koff[q]) l = q;
    const int jj = j - L.koff[l], kl = L.k[l];
    if (jj >= ((const int*)(ws + L.cnt_off[l]))[img]) {      // not selected (the row holds fewer than k finite logits): a masked zero box
      *(float4*)(boxes + 4 * t) = make_float4(0.f, 0.f, 0.f, 0.f);
      scores[t] = 0.f;
      masked[t] = -INFINITY;
      lvl[t] = l;
      continue;
    }
    const long long a = (long long)L.start[l] + ((const long long*)(ws + L.idx_off[l]))[(long long)img * kl + jj];
    const float logit = ((const float*)(ws + L.val_off[l]))[(long long)img * kl + jj];
    const float4 b = *(const float4*)(anchors + 4 * a), c = *(const float4*)(deltas + 4 * ((long long)img * A + a));
    const float w = b.z - b.x, h = b.w - b.y, cx = b.x + 0.5f * w, cy = b.y + 0.5f * h;
    const float dw = fminf(c.z, xform_clip), dh = fminf(c.w, xform_clip);
    const float pcx = c.x * w + cx, pcy = c.y * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
    float4 o = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
    const float4 m = *(const float4*)(lim + 4 * img);
    o.x = fminf(fmaxf(o.x, 0.f), m.x);
    o.y = fminf(fmaxf(o.y, 0.f), m.y);
    o.z = fminf(fmaxf(o.z, 0.f), m.z);
    o.w = fminf(fmaxf(o.w, 0.f), m.w);
    const float s = 1.0f / (1.0f + expf(-logit));
    const bool valid = (o.z - o.x >= min_size) && (o.w - o.y >= min_size) && (s >= score_thresh);
    *(float4*)(boxes + 4 * t) = o;
    scores[t] = s;
    masked[t] = valid ? s : -INFINITY;
    lvl[t] = l;
  }
}

// One workgroup per image: the kept list is in descending (masked) score order, so the masked candidates that survived come last and the
// valid survivors are a prefix; count them, cut at post_nms_top_n, gather.
__global__ __launch_bounds__(256) void rpn_gather_kernel(const float* __restrict__ boxes, const float* __restrict__ masked,
                                                         const float* __restrict__ scores, const long long* __restrict__ keep,
                                                         const int* __restrict__ keep_cnt, int K, int post, float* __restrict__ out_boxes,
                                                         float* __restrict__ out_scores, int* __restrict__ out_counts) {
  __shared__ int s_cnt;
  const int img = blockIdx.x;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const int kc = min(keep_cnt[img], K);
  const long long* kp = keep + (long long)img * K;
  int mine = 0;
  for (int j = threadIdx.x; j < kc; j += blockDim.x) mine += masked[(long long)img * K + kp[j]] > -INFINITY ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, WAVE);
  if ((threadIdx.x & (WAVE - 1)) == 0 && mine) atomicAdd(&s_cnt, mine);
  __syncthreads();
  const int cnt = min(s_cnt, post);
  if (threadIdx.x == 0) out_counts[img] = cnt;
  for (int j = threadIdx.x; j < post; j += blockDim.x) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
    if (j < cnt) {
      const long long src = (long long)img * K + kp[j];
      b = *(const float4*)(boxes + 4 * src);
      s = scores[src];
    }
    *(float4*)(out_boxes + 4 * ((long long)img * post + j)) = b;
    out_scores[(long long)img * post + j] = s;
  }
}

// ---- RoIHeads.select_training_samples (tvision/roi_heads.py:627-713) for the whole batch --------------------------------------------
// Two launches around the one host read the sampler needs (its `torch.randperm` calls take the positive / negative counts as sizes and
// stay in torch so that the draws are the reference's):
//   roi_match_kernel   candidates of image i = its proposals followed by its ground-truth boxes (add_gt_proposals); box_iou + Matcher
//                      (high = low threshold style, no low-quality rescue) + label assignment (assign_targets_to_proposals); counts of
//                      positives (label >= 1) and negatives (label == 0) per image
//   roi_sample_kernel  positive[perm_pos[:num_pos]] and negative[perm_neg[:num_neg]] (index lists in ascending order, as torch.where
//                      gives them), the union in ascending order, gathers, BoxCoder.encode of the matched ground truth
constexpr int ROI_MAX_IMAGES = 64;
constexpr int ROI_MAX_GT = 1024;
constexpr int ROI_MAX_CAND = 8192;
constexpr int ROI_MAX_SAMPLES = 1024;

struct RoiImages {
  int n;
  int gt_off[ROI_MAX_IMAGES + 1];
};

struct RoiSampleArgs {
  int gt_off[ROI_MAX_IMAGES + 1], out_off[ROI_MAX_IMAGES + 1];
  int num_pos[ROI_MAX_IMAGES], num_neg[ROI_MAX_IMAGES];
  const long long* perm_pos[ROI_MAX_IMAGES];
  const long long* perm_neg[ROI_MAX_IMAGES];
};

__device__ __forceinline__ float roi_iou(const float4 a, const float4 b) {      // torchvision box_iou(a = ground truth, b = candidate)
  const float area_a = (a.z - a.x) * (a.w - a.y), area_b = (b.z - b.x) * (b.w - b.y);
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.0f), h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.0f);
  const float inter = w * h;
  return inter / (area_a + area_b - inter);
}

__device__ __forceinline__ float4 roi_candidate(const float* __restrict__ props, const float* __restrict__ gt, int img, int P, int pc, int gt0, int j) {
  return j < pc ? *(const float4*)(props + 4 * ((long long)img * P + j)) : *(const float4*)(gt + 4 * (long long)(gt0 + j - pc));
}

__global__ __launch_bounds__(256) void roi_match_kernel(const float* __restrict__ props, const int* __restrict__ pcount, int P,
                                                        const float* __restrict__ gt, const long long* __restrict__ gt_labels, RoiImages I,
                                                        float hi, float lo, int C, int* __restrict__ matched, int* __restrict__ label,
                                                        int* __restrict__ counts) {
  __shared__ float4 sg[ROI_MAX_GT];
  __shared__ int s_pos, s_neg;
  const int img = blockIdx.y, gt0 = I.gt_off[img], g = I.gt_off[img + 1] - gt0;
  const int pc = min(pcount[img], P), c = pc + g;
  if ((int)(blockIdx.x * blockDim.x) >= c) return;
  for (int q = threadIdx.x; q < g; q += blockDim.x) sg[q] = *(const float4*)(gt + 4 * (long long)(gt0 + q));
  if (threadIdx.x == 0) s_pos = s_neg = 0;
  __syncthreads();
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  int lab = -2;
  if (j < c) {
    const float4 b = roi_candidate(props, gt, img, P, pc, gt0, j);
    float best = -INFINITY;
    int arg = 0;
    for (int q = 0; q < g; ++q) {
      const float v = roi_iou(sg[q], b);
      if (q == 0 || v > best) {           // first maximum, as torch.max(dim=0)
        best = v;
        arg = q;
      }
    }
    int m = arg;
    if (best < lo) m = -1;                // Matcher.BELOW_LOW_THRESHOLD
    else if (best < hi) m = -2;           // Matcher.BETWEEN_THRESHOLDS
    const int cl = max(m, 0);             // roi_heads.py:640 clamp(min=0)
    lab = m == -1 ? 0 : (m == -2 ? -1 : (int)gt_labels[gt0 + cl]);
    matched[(long long)img * C + j] = cl;
    label[(long long)img * C + j] = lab;
  }
  const unsigned long long bp = __ballot(lab >= 1), bn = __ballot(lab == 0);
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    if (bp) atomicAdd(&s_pos, __popcll(bp));
    if (bn) atomicAdd(&s_neg, __popcll(bn));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_pos) atomicAdd(counts + 2 * img, s_pos);
    if (s_neg) atomicAdd(counts + 2 * img + 1, s_neg);
  }
}

__global__ __launch_bounds__(1024) void roi_sample_kernel(const float* __restrict__ props, const int* __restrict__ pcount, int P,
                                                          const float* __restrict__ gt, RoiSampleArgs A, int C, const int* __restrict__ matched,
                                                          const int* __restrict__ label, float wx, float wy, float ww, float wh,
                                                          float* __restrict__ rois, long long* __restrict__ out_labels,
                                                          long long* __restrict__ out_matched, float* __restrict__ out_reg) {
  __shared__ unsigned short posl[ROI_MAX_CAND], negl[ROI_MAX_CAND];      // candidate indices < 8192
  __shared__ int chosen[ROI_MAX_SAMPLES];
  __shared__ int wsp[16], wsn[16], s_runp, s_runn;
  const int img = blockIdx.x, gt0 = A.gt_off[img], g = A.gt_off[img + 1] - gt0;
  const int pc = min(pcount[img], P), c = pc + g;
  const int* lab = label + (long long)img * C;
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  if (threadIdx.x == 0) s_runp = s_runn = 0;
  __syncthreads();
  for (int base = 0; base < c; base += 1024) {          // index lists of the positives / negatives in ascending order (torch.where)
    const int j = base + threadIdx.x;
    const int l = j < c ? lab[j] : -2;
    const unsigned long long bp = __ballot(l >= 1), bn = __ballot(l == 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) {
      wsp[wid] = __popcll(bp);
      wsn[wid] = __popcll(bn);
    }
    __syncthreads();
    int op = s_runp, on = s_runn;
    for (int w = 0; w < wid; ++w) op += wsp[w], on += wsn[w];
    if (l >= 1) posl[op + __popcll(bp & below)] = (unsigned short)j;
    if (l == 0) negl[on + __popcll(bn & below)] = (unsigned short)j;
    __syncthreads();
    if (threadIdx.x == 0) {
      int tp = 0, tn = 0;
      for (int w = 0; w < 16; ++w) tp += wsp[w], tn += wsn[w];
      s_runp += tp;
      s_runn += tn;
    }
    __syncthreads();
  }
  const int np = A.num_pos[img], nn = A.num_neg[img], ns = np + nn;
  int npad = 64;
  while (npad < ns) npad <<= 1;
  for (int q = threadIdx.x; q < npad; q += 1024)
    chosen[q] = q < np ? (int)posl[A.perm_pos[img][q]] : (q < ns ? (int)negl[A.perm_neg[img][q - np]] : 0x7fffffff);
  __syncthreads();
  for (int kk = 2; kk <= npad; kk <<= 1)                 // ascending: the union mask of the reference read back with torch.where
    for (int jj = kk >> 1; jj > 0; jj >>= 1) {
      const int q = threadIdx.x, x = q ^ jj;
      if (q < npad && x > q) {
        const int a = chosen[q], b = chosen[x];
        if (((q & kk) == 0) ? a > b : a < b) {
          chosen[q] = b;
          chosen[x] = a;
        }
      }
      __syncthreads();
    }
  for (int q = threadIdx.x; q < ns; q += 1024) {
    const int j = chosen[q], m = matched[(long long)img * C + j];
    const float4 p = roi_candidate(props, gt, img, P, pc, gt0, j), r = *(const float4*)(gt + 4 * (long long)(gt0 + m));
    const long long o = A.out_off[img] + q;
    rois[5 * o] = (float)img;
    rois[5 * o + 1] = p.x;
    rois[5 * o + 2] = p.y;
    rois[5 * o + 3] = p.z;
    rois[5 * o + 4] = p.w;
    out_labels[o] = lab[j];
    out_matched[o] = m;
    // BoxCoder.encode_single (tvision/_utils.py:79-125), the operation order of box_encode_kernel
    const float ew = p.z - p.x, eh = p.w - p.y, ecx = p.x + 0.5f * ew, ecy = p.y + 0.5f * eh;
    const float gw = r.z - r.x, gh = r.w - r.y, gcx = r.x + 0.5f * gw, gcy = r.y + 0.5f * gh;
    *(float4*)(out_reg + 4 * o) = make_float4(wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh, ww * logf(gw / ew), wh * logf(gh / eh));
  }
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct ProposalWs {
  ProposalLevels L;
  int K;
  size_t topk_ws, boxes, masked, scores, lvl, keep, keep_cnt, nms_ws, total;
};

int proposal_layout(int n_images, const int64_t* level_counts, int nlev, int pre, ProposalWs& W) {
  if (n_images <= 0 || nlev <= 0 || nlev > MAX_LEVELS || pre <= 0 || !level_counts) return 1;
  size_t off = 0;
  long long start = 0;
  W.L.nlev = nlev;
  W.L.koff[0] = 0;
  for (int l = 0; l < nlev; ++l) {
    if (level_counts[l] <= 0 || start + level_counts[l] >= (1ll << 31)) return 1;
    const int k = (int)(level_counts[l] < pre ? level_counts[l] : pre);
    W.L.start[l] = (int)start;
    W.L.k[l] = k;
    W.L.koff[l + 1] = W.L.koff[l] + k;
    W.L.idx_off[l] = (long long)off;
    off = align256(off + (size_t)n_images * k * sizeof(int64_t));
    W.L.val_off[l] = (long long)off;
    off = align256(off + (size_t)n_images * k * sizeof(float));
    W.L.cnt_off[l] = (long long)off;
    off = align256(off + (size_t)n_images * sizeof(int32_t));
    start += level_counts[l];
  }
  for (int l = nlev; l < MAX_LEVELS; ++l) W.L.start[l] = W.L.k[l] = 0, W.L.koff[l + 1] = W.L.koff[nlev], W.L.idx_off[l] = W.L.val_off[l] = W.L.cnt_off[l] = 0;
  W.K = W.L.koff[nlev];
  const size_t NK = (size_t)n_images * W.K;
  W.topk_ws = off, off = align256(off + mi355det_topk_workspace(n_images));
  W.boxes = off, off = align256(off + NK * 4 * sizeof(float));
  W.masked = off, off = align256(off + NK * sizeof(float));
  W.scores = off, off = align256(off + NK * sizeof(float));
  W.lvl = off, off = align256(off + NK * sizeof(int64_t));
  W.keep = off, off = align256(off + NK * sizeof(int64_t));
  W.keep_cnt = off, off = align256(off + sizeof(int32_t) * (size_t)n_images);
  W.nms_ws = off, off = align256(off + mi355det_nms_workspace(n_images, W.K));
  W.total = off;
  return 0;
}

// ---- RetinaNet.postprocess_detections for the whole batch (tvision/retinanet.py:414-472) ------------------------------------------------
// per level: thresholded top-k over the flattened [HWA x K] scores of every image (mi355det_topk_ws), then ONE kernel for all levels:
// anchor / class from the flat index, decode + clip of the selected anchors, sigmoid; per-class NMS of all images side by side; the first
// detections_per_img survivors gathered.  Candidates a level could not fill (fewer than k scores above the threshold) are masked entries.
struct RetinaLevels {
  int nlev, num_classes;
  int k[MAX_LEVELS], koff[MAX_LEVELS + 1];
  long long hwa[MAX_LEVELS];
  long long idx_off[MAX_LEVELS], val_off[MAX_LEVELS], cnt_off[MAX_LEVELS];
  const float* reg[MAX_LEVELS];        // [N, HWA_l, 4]
  const float* anchors[MAX_LEVELS];    // [HWA_l, 4]
};

__global__ __launch_bounds__(256) void retina_select_kernel(const char* __restrict__ ws, RetinaLevels L, const float* __restrict__ lim, int n_images,
                                                            float xform_clip, float* __restrict__ boxes, float* __restrict__ masked,
                                                            float* __restrict__ scores, long long* __restrict__ labels) {
  const int K = L.koff[L.nlev];
  const long long total = (long long)n_images * K;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int img = (int)(t / K), j = (int)(t - (long long)img * K);
    int l = 0;
#pragma unroll
    for (int q = 1; q < MAX_LEVELS; ++q)
      if (q < L.nlev && j >= L.koff[q]) l = q;
    const int jj = j - L.koff[l], kl = L.k[l];
    if (jj >= ((const int*)(ws + L.cnt_off[l]))[img]) {
      *(float4*)(boxes + 4 * t) = make_float4(0.f, 0.f, 0.f, 0.f);
      scores[t] = 0.f;
      masked[t] = -INFINITY;
      labels[t] = 0;
      continue;
    }
    const long long flat = ((const long long*)(ws + L.idx_off[l]))[(long long)img * kl + jj];
    const float logit = ((const float*)(ws + L.val_off[l]))[(long long)img * kl + jj];
    const long long a = flat / L.num_classes;
    const int cls = (int)(flat - a * L.num_classes);
    const float4 b = *(const float4*)(L.anchors[l] + 4 * a), c = *(const float4*)(L.reg[l] + 4 * ((long long)img * L.hwa[l] + a));
    const float w = b.z - b.x, h = b.w - b.y, cx = b.x + 0.5f * w, cy = b.y + 0.5f * h;
    const float dw = fminf(c.z, xform_clip), dh = fminf(c.w, xform_clip);
    const float pcx = c.x * w + cx, pcy = c.y * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
    float4 o = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
    const float4 m = *(const float4*)(lim + 4 * img);
    o.x = fminf(fmaxf(o.x, 0.f), m.x);
    o.y = fminf(fmaxf(o.y, 0.f), m.y);
    o.z = fminf(fmaxf(o.z, 0.f), m.z);
    o.w = fminf(fmaxf(o.w, 0.f), m.w);
    const float sc = 1.0f / (1.0f + expf(-logit));
    *(float4*)(boxes + 4 * t) = o;
    scores[t] = sc;
    masked[t] = sc;
    labels[t] = cls;
  }
}

__global__ __launch_bounds__(256) void retina_gather_kernel(const float* __restrict__ boxes, const float* __restrict__ masked,
                                                            const float* __restrict__ scores, const long long* __restrict__ labels,
                                                            const long long* __restrict__ keep, const int* __restrict__ keep_cnt, int K, int post,
                                                            float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                            long long* __restrict__ out_labels, int* __restrict__ out_counts) {
  __shared__ int s_cnt;
  const int img = blockIdx.x;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const int kc = min(keep_cnt[img], K);
  const long long* kp = keep + (long long)img * K;
  int mine = 0;
  for (int j = threadIdx.x; j < kc; j += blockDim.x) mine += masked[(long long)img * K + kp[j]] > -INFINITY ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, WAVE);
  if ((threadIdx.x & (WAVE - 1)) == 0 && mine) atomicAdd(&s_cnt, mine);
  __syncthreads();
  const int cnt = min(s_cnt, post);
  if (threadIdx.x == 0) out_counts[img] = cnt;
  for (int j = threadIdx.x; j < post; j += blockDim.x) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    float sc = 0.f;
    long long lb = 0;
    if (j < cnt) {
      const long long src = (long long)img * K + kp[j];
      b = *(const float4*)(boxes + 4 * src);
      sc = scores[src];
      lb = labels[src];
    }
    *(float4*)(out_boxes + 4 * ((long long)img * post + j)) = b;
    out_scores[(long long)img * post + j] = sc;
    out_labels[(long long)img * post + j] = lb;
  }
}

struct RetinaWs {
  RetinaLevels L;
  int K;
  size_t topk_ws, boxes, masked, scores, labels, keep, keep_cnt, nms_ws, total;
};

int retina_layout(int n_images, const int64_t* level_anchors, int nlev, int num_classes, int topk, RetinaWs& W) {
  if (n_images <= 0 || nlev <= 0 || nlev > MAX_LEVELS || num_classes <= 0 || topk <= 0 || topk > 16384 || !level_anchors) return 1;
  size_t off = 0;
  W.L.nlev = nlev;
  W.L.num_classes = num_classes;
  W.L.koff[0] = 0;
  for (int l = 0; l < nlev; ++l) {
    const long long n = level_anchors[l] * (long long)num_classes;
    if (level_anchors[l] <= 0 || n >= (1ll << 32)) return 1;
    const int k = (int)(n < topk ? n : topk);
    W.L.k[l] = k;
    W.L.hwa[l] = level_anchors[l];
    W.L.koff[l + 1] = W.L.koff[l] + k;
    W.L.idx_off[l] = (long long)off;
    off = align256(off + (size_t)n_images * k * sizeof(int64_t));
    W.L.val_off[l] = (long long)off;
    off = align256(off + (size_t)n_images * k * sizeof(float));
    W.L.cnt_off[l] = (long long)off;
    off = align256(off + (size_t)n_images * sizeof(int32_t));
  }
  for (int l = nlev; l < MAX_LEVELS; ++l) {
    W.L.k[l] = 0, W.L.hwa[l] = 0, W.L.koff[l + 1] = W.L.koff[nlev], W.L.idx_off[l] = W.L.val_off[l] = W.L.cnt_off[l] = 0;
    W.L.reg[l] = W.L.anchors[l] = nullptr;
  }
  W.K = W.L.koff[nlev];
  const size_t NK = (size_t)n_images * W.K;
  W.topk_ws = off, off = align256(off + mi355det_topk_workspace(n_images));
  W.boxes = off, off = align256(off + NK * 4 * sizeof(float));
  W.masked = off, off = align256(off + NK * sizeof(float));
  W.scores = off, off = align256(off + NK * sizeof(float));
  W.labels = off, off = align256(off + NK * sizeof(int64_t));
  W.keep = off, off = align256(off + NK * sizeof(int64_t));
  W.keep_cnt = off, off = align256(off + sizeof(int32_t) * (size_t)n_images);
  W.nms_ws = off, off = align256(off + mi355det_nms_workspace(n_images, W.K));
  W.total = off;
  return 0;
}

// ---- RegionProposalNetwork.compute_loss (tvision/rpn.py:282-318), forward and gradient in one launch ------------------------------------
// objectness_loss = mean over the sampled anchors of BCE-with-logits, box_loss = sum over the positive anchors of smooth-L1 (beta 1/9) /
// number of sampled anchors.  The gradients go straight into the dense [T] / [T,4] buffers the network backward reads (zero elsewhere): the
// autograd form needs ~25 launches, among them two sort-based `index_put(accumulate)` for the gathers' backward.  One workgroup, fixed
// summation order (at most a few thousand sampled anchors).
__global__ __launch_bounds__(1024) void rpn_loss_kernel(const float* __restrict__ obj, const float* __restrict__ deltas,
                                                        const float* __restrict__ labels, const float* __restrict__ targets,
                                                        const long long* __restrict__ pos, int P, const long long* __restrict__ sampled, int S,
                                                        float* __restrict__ losses, float* __restrict__ grad_obj, float* __restrict__ grad_deltas) {
  __shared__ float red[2][1024 / WAVE];
  const float inv = 1.0f / (float)S, beta = 1.0f / 9;
  float lo = 0.f, lb = 0.f;
  for (int i = threadIdx.x; i < S; i += 1024) {
    const long long a = sampled[i];
    const float x = obj[a], y = labels[a];
    lo += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
    grad_obj[a] = (1.0f / (1.0f + expf(-x)) - y) * inv;
  }
  for (int i = threadIdx.x; i < 4 * P; i += 1024) {
    const long long a = pos[i >> 2] * 4 + (i & 3);
    const float d = deltas[a] - targets[a], n = fabsf(d);
    lb += n < beta ? 0.5f * n * n / beta : n - 0.5f * beta;
    grad_deltas[a] = (n < beta ? d / beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f))) * inv;
  }
  lo = wave_sum(lo);
  lb = wave_sum(lb);
  if ((threadIdx.x & (WAVE - 1)) == 0) {
    red[0][threadIdx.x / WAVE] = lo;
    red[1][threadIdx.x / WAVE] = lb;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    float t = 0.f;
    for (int w = 0; w < 1024 / WAVE; ++w) t += red[threadIdx.x][w];
    losses[threadIdx.x] = t * inv;
  }
}

// ---- RoIHeads.postprocess_detections for the whole batch (tvision/roi_heads.py:715-781) ------------------------------------------------
// scores [N, P, C] (softmax / sigmoid / gombit already applied, class 0 and padded proposals pushed below the threshold by the caller),
// box_regression [N, P, C, 4], proposals [N, P, 4]: thresholded top-k over the flattened P x C scores of every image, decode of the
// selected (proposal, class) pairs with the head's BoxCoder weights, clip, small boxes masked, per-class NMS, the first detections_per_img.
__global__ __launch_bounds__(256) void roi_det_select_kernel(const long long* __restrict__ idx, const float* __restrict__ val,
                                                             const int* __restrict__ cnt, int k, int P, int C, const float* __restrict__ reg,
                                                             const float* __restrict__ props, const float* __restrict__ lim, int n_images, float wx,
                                                             float wy, float ww, float wh, float xform_clip, float min_size,
                                                             float* __restrict__ boxes, float* __restrict__ masked, float* __restrict__ scores,
                                                             long long* __restrict__ labels) {
  const long long total = (long long)n_images * k;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int img = (int)(t / k), j = (int)(t - (long long)img * k);
    if (j >= cnt[img]) {
      *(float4*)(boxes + 4 * t) = make_float4(0.f, 0.f, 0.f, 0.f);
      scores[t] = 0.f;
      masked[t] = -INFINITY;
      labels[t] = 0;
      continue;
    }
    const long long flat = idx[t];
    const int p = (int)(flat / C), cls = (int)(flat - (long long)p * C);
    const float4 b = *(const float4*)(props + 4 * ((long long)img * P + p));
    const float4 c = *(const float4*)(reg + 4 * (((long long)img * P + p) * C + cls));
    const float w = b.z - b.x, h = b.w - b.y, cx = b.x + 0.5f * w, cy = b.y + 0.5f * h;
    const float dx = c.x / wx, dy = c.y / wy, dw = fminf(c.z / ww, xform_clip), dh = fminf(c.w / wh, xform_clip);      // box_decode_kernel's order
    const float pcx = dx * w + cx, pcy = dy * h + cy, pw = expf(dw) * w, ph = expf(dh) * h;
    float4 o = make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
    const float4 m = *(const float4*)(lim + 4 * img);
    o.x = fminf(fmaxf(o.x, 0.f), m.x);
    o.y = fminf(fmaxf(o.y, 0.f), m.y);
    o.z = fminf(fmaxf(o.z, 0.f), m.z);
    o.w = fminf(fmaxf(o.w, 0.f), m.w);
    const float sc = val[t];
    const bool valid = (o.z - o.x >= min_size) && (o.w - o.y >= min_size);
    *(float4*)(boxes + 4 * t) = o;
    scores[t] = sc;
    masked[t] = valid ? sc : -INFINITY;
    labels[t] = cls;
  }
}

struct RoiDetWs {
  size_t idx, val, cnt, topk_ws, boxes, masked, scores, labels, keep, keep_cnt, nms_ws, total;
};

int roi_det_layout(int n_images, long long row, int k, RoiDetWs& W) {
  if (n_images <= 0 || row <= 0 || row >= (1ll << 32) || k <= 0 || k > 16384 || k > row) return 1;
  const size_t NK = (size_t)n_images * k;
  size_t off = 0;
  W.idx = off, off = align256(off + NK * sizeof(int64_t));
  W.val = off, off = align256(off + NK * sizeof(float));
  W.cnt = off, off = align256(off + sizeof(int32_t) * (size_t)n_images);
  W.topk_ws = off, off = align256(off + mi355det_topk_workspace(n_images));
  W.boxes = off, off = align256(off + NK * 4 * sizeof(float));
  W.masked = off, off = align256(off + NK * sizeof(float));
  W.scores = off, off = align256(off + NK * sizeof(float));
  W.labels = off, off = align256(off + NK * sizeof(int64_t));
  W.keep = off, off = align256(off + NK * sizeof(int64_t));
  W.keep_cnt = off, off = align256(off + sizeof(int32_t) * (size_t)n_images);
  W.nms_ws = off, off = align256(off + mi355det_nms_workspace(n_images, k));
  W.total = off;
  return 0;
}

}  // namespace

extern "C" {

size_t mi355det_rpn_proposals_workspace(int32_t n_images, const int64_t* level_counts, int32_t nlev, int32_t pre_nms_top_n) {
  ProposalWs W;
  if (proposal_layout(n_images, level_counts, nlev, pre_nms_top_n, W)) return 0;
  return W.total;
}

int mi355det_rpn_proposals(const float* objectness, const float* deltas, const float* anchors, const float* clip_limits, int32_t n_images,
                           const int64_t* level_counts, int32_t nlev, int32_t pre_nms_top_n, int32_t post_nms_top_n, float nms_thresh,
                           float score_thresh, float min_size, float xform_clip, float* out_boxes, float* out_scores, int32_t* out_counts,
                           void* workspace, size_t workspace_bytes, void* stream) {
  ProposalWs W;
  if (proposal_layout(n_images, level_counts, nlev, pre_nms_top_n, W))
    return fail(MI355DET_EINVAL, "%s: need 1..8 non-empty levels, fewer than 2^31 anchors, positive batch and pre_nms_top_n", "rpn_proposals");
  if (post_nms_top_n <= 0) return fail(MI355DET_EINVAL, "%s: post_nms_top_n must be positive", "rpn_proposals");
  if (!objectness || !deltas || !anchors || !clip_limits || !out_boxes || !out_scores || !out_counts || !workspace)
    return fail(MI355DET_EINVAL, "%s: null argument", "rpn_proposals");
  if (workspace_bytes < W.total) return fail(MI355DET_EWORKSPACE, "%s: workspace too small", "rpn_proposals");
  char* ws = (char*)workspace;
  long long A = 0;
  for (int l = 0; l < nlev; ++l) A += level_counts[l];
  {                                      // rpn.py:215-228: per-level top-k of the logits; every level and image in one launch sequence
    int64_t seg_start[MAX_LEVELS];
    int32_t seg_k[MAX_LEVELS];
    int64_t* idx_out[MAX_LEVELS];
    float* val_out[MAX_LEVELS];
    int32_t* cnt_out[MAX_LEVELS];
    for (int l = 0; l < nlev; ++l) {
      seg_start[l] = W.L.start[l];
      seg_k[l] = W.L.k[l];
      idx_out[l] = (int64_t*)(ws + W.L.idx_off[l]);
      val_out[l] = (float*)(ws + W.L.val_off[l]);
      cnt_out[l] = (int32_t*)(ws + W.L.cnt_off[l]);
    }
    if (int e = mi355det_topk_segments(objectness, n_images, A, nlev, seg_start, level_counts, seg_k, -INFINITY, idx_out, val_out, cnt_out,
                                       ws + W.topk_ws, mi355det_topk_workspace(n_images), stream))
      return e;
  }
  const long long total = (long long)n_images * W.K;
  hipLaunchKernelGGL(rpn_select_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, S(stream), ws, W.L, deltas, anchors, clip_limits, n_images,
                     A, xform_clip, min_size, score_thresh, (float*)(ws + W.boxes), (float*)(ws + W.masked), (float*)(ws + W.scores),
                     (long long*)(ws + W.lvl));
  if (int e = mi355det_nms_batch((const float*)(ws + W.boxes), (const float*)(ws + W.masked), (const int64_t*)(ws + W.lvl), n_images, W.K,
                                 nms_thresh, (int64_t*)(ws + W.keep), (int32_t*)(ws + W.keep_cnt), ws + W.nms_ws,
                                 mi355det_nms_workspace(n_images, W.K), stream))
    return e;
  hipLaunchKernelGGL(rpn_gather_kernel, dim3(n_images), dim3(256), 0, S(stream), (const float*)(ws + W.boxes), (const float*)(ws + W.masked),
                     (const float*)(ws + W.scores), (const long long*)(ws + W.keep), (const int*)(ws + W.keep_cnt), W.K, post_nms_top_n, out_boxes,
                     out_scores, out_counts);
  return check_launch("rpn_proposals");
}

int mi355det_roi_match(const float* proposals, const int32_t* proposal_counts, int32_t n_images, int32_t max_proposals, const float* gt_boxes,
                       const int64_t* gt_labels, const int32_t* gt_offsets, float fg_iou_thresh, float bg_iou_thresh, int32_t row_stride,
                       int32_t* matched, int32_t* labels, int32_t* counts, void* stream) {
  if (n_images <= 0 || n_images > ROI_MAX_IMAGES || max_proposals < 0 || !gt_offsets)
    return fail(MI355DET_EINVAL, "%s: 1..64 images", "roi_match");
  if (!proposals || !proposal_counts || !gt_boxes || !gt_labels || !matched || !labels || !counts) return fail(MI355DET_EINVAL, "%s: null argument", "roi_match");
  RoiImages I{};
  I.n = n_images;
  int gmax = 0;
  for (int i = 0; i <= n_images; ++i) I.gt_off[i] = gt_offsets[i];
  for (int i = 0; i < n_images; ++i) {
    const int g = gt_offsets[i + 1] - gt_offsets[i];
    // the reference's Matcher raises on an image without ground truth (tvision/_utils.py:282-291): the Python mirror does that
    if (g <= 0 || g > ROI_MAX_GT) return fail(MI355DET_EINVAL, "%s: 1..1024 ground-truth boxes per image", "roi_match");
    gmax = g > gmax ? g : gmax;
  }
  if (row_stride < max_proposals + gmax) return fail(MI355DET_EINVAL, "%s: row_stride < max_proposals + ground-truth boxes", "roi_match");
  if (hipMemsetAsync(counts, 0, sizeof(int32_t) * 2 * (size_t)n_images, S(stream)) != hipSuccess) return fail(MI355DET_ELAUNCH, "%s: memset failed", "roi_match");
  hipLaunchKernelGGL(roi_match_kernel, dim3((max_proposals + gmax + 255) / 256, n_images), dim3(256), 0, S(stream), proposals, proposal_counts,
                     max_proposals, gt_boxes, (const long long*)gt_labels, I, fg_iou_thresh, bg_iou_thresh, row_stride, matched, labels, counts);
  return check_launch("roi_match");
}

int mi355det_roi_sample(const float* proposals, const int32_t* proposal_counts, int32_t n_images, int32_t max_proposals, const float* gt_boxes,
                        const int32_t* gt_offsets, int32_t row_stride, const int32_t* matched, const int32_t* labels,
                        const int64_t* const* perm_pos, const int64_t* const* perm_neg, const int32_t* num_pos, const int32_t* num_neg, float wx,
                        float wy, float ww, float wh, float* rois, int64_t* out_labels, int64_t* out_matched, float* out_regression_targets,
                        void* stream) {
  if (n_images <= 0 || n_images > ROI_MAX_IMAGES || !gt_offsets || !perm_pos || !perm_neg || !num_pos || !num_neg)
    return fail(MI355DET_EINVAL, "%s: 1..64 images", "roi_sample");
  if (!proposals || !proposal_counts || !gt_boxes || !matched || !labels || !rois || !out_labels || !out_matched || !out_regression_targets)
    return fail(MI355DET_EINVAL, "%s: null argument", "roi_sample");
  RoiSampleArgs A{};
  int gmax = 0, total = 0;
  for (int i = 0; i <= n_images; ++i) A.gt_off[i] = gt_offsets[i];
  for (int i = 0; i < n_images; ++i) {
    const int g = gt_offsets[i + 1] - gt_offsets[i];
    gmax = g > gmax ? g : gmax;
    if (num_pos[i] < 0 || num_neg[i] < 0 || num_pos[i] + num_neg[i] > ROI_MAX_SAMPLES)
      return fail(MI355DET_EINVAL, "%s: at most 1024 samples per image", "roi_sample");
    if ((num_pos[i] && !perm_pos[i]) || (num_neg[i] && !perm_neg[i])) return fail(MI355DET_EINVAL, "%s: missing permutation", "roi_sample");
    A.out_off[i] = total;
    A.num_pos[i] = num_pos[i];
    A.num_neg[i] = num_neg[i];
    A.perm_pos[i] = (const long long*)perm_pos[i];
    A.perm_neg[i] = (const long long*)perm_neg[i];
    total += num_pos[i] + num_neg[i];
  }
  A.out_off[n_images] = total;
  if (max_proposals + gmax > ROI_MAX_CAND || row_stride < max_proposals + gmax)
    return fail(MI355DET_EINVAL, "%s: at most 8192 candidates (proposals + ground truth) per image", "roi_sample");
  if (total == 0) return 0;
  hipLaunchKernelGGL(roi_sample_kernel, dim3(n_images), dim3(1024), 0, S(stream), proposals, proposal_counts, max_proposals, gt_boxes, A, row_stride,
                     matched, labels, wx, wy, ww, wh, rois, (long long*)out_labels, (long long*)out_matched, out_regression_targets);
  return check_launch("roi_sample");
}

size_t mi355det_retina_detections_workspace(int32_t n_images, const int64_t* level_anchors, int32_t nlev, int32_t num_classes, int32_t topk_candidates) {
  RetinaWs W;
  if (retina_layout(n_images, level_anchors, nlev, num_classes, topk_candidates, W)) return 0;
  return W.total;
}

int mi355det_retina_detections(const float* const* cls_logits, const float* const* bbox_regression, const float* const* anchors,
                               const int64_t* level_anchors, int32_t nlev, int32_t n_images, int32_t num_classes, const float* clip_limits,
                               float logit_thresh, int32_t topk_candidates, float nms_thresh, int32_t detections_per_img, float xform_clip,
                               float* out_boxes, float* out_scores, int64_t* out_labels, int32_t* out_counts, void* workspace,
                               size_t workspace_bytes, void* stream) {
  RetinaWs W;
  if (retina_layout(n_images, level_anchors, nlev, num_classes, topk_candidates, W))
    return fail(MI355DET_EINVAL, "%s: need 1..8 levels with fewer than 2^32 scores per image, 1 <= topk_candidates <= 16384", "retina_detections");
  if (detections_per_img <= 0 || !cls_logits || !bbox_regression || !anchors || !clip_limits || !out_boxes || !out_scores || !out_labels || !out_counts ||
      !workspace)
    return fail(MI355DET_EINVAL, "%s: null argument or detections_per_img <= 0", "retina_detections");
  if (workspace_bytes < W.total) return fail(MI355DET_EWORKSPACE, "%s: workspace too small", "retina_detections");
  char* ws = (char*)workspace;
  for (int l = 0; l < nlev; ++l) {
    if (!cls_logits[l] || !bbox_regression[l] || !anchors[l]) return fail(MI355DET_EINVAL, "%s: null level pointer", "retina_detections");
    W.L.reg[l] = bbox_regression[l];
    W.L.anchors[l] = anchors[l];
    const long long n = level_anchors[l] * (long long)num_classes;          // retinanet.py:437-445: threshold, then top-k of the flattened scores
    if (int e = mi355det_topk_ws(cls_logits[l], n_images, n, n, W.L.k[l], logit_thresh, (int64_t*)(ws + W.L.idx_off[l]), (float*)(ws + W.L.val_off[l]),
                                 (int32_t*)(ws + W.L.cnt_off[l]), ws + W.topk_ws, mi355det_topk_workspace(n_images), stream))
      return e;
  }
  const long long total = (long long)n_images * W.K;
  hipLaunchKernelGGL(retina_select_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, S(stream), ws, W.L, clip_limits, n_images, xform_clip,
                     (float*)(ws + W.boxes), (float*)(ws + W.masked), (float*)(ws + W.scores), (long long*)(ws + W.labels));
  if (int e = mi355det_nms_batch((const float*)(ws + W.boxes), (const float*)(ws + W.masked), (const int64_t*)(ws + W.labels), n_images, W.K, nms_thresh,
                                 (int64_t*)(ws + W.keep), (int32_t*)(ws + W.keep_cnt), ws + W.nms_ws, mi355det_nms_workspace(n_images, W.K), stream))
    return e;
  hipLaunchKernelGGL(retina_gather_kernel, dim3(n_images), dim3(256), 0, S(stream), (const float*)(ws + W.boxes), (const float*)(ws + W.masked),
                     (const float*)(ws + W.scores), (const long long*)(ws + W.labels), (const long long*)(ws + W.keep), (const int*)(ws + W.keep_cnt), W.K,
                     detections_per_img, out_boxes, out_scores, (long long*)out_labels, out_counts);
  return check_launch("retina_detections");
}

int mi355det_rpn_loss(const float* objectness, const float* pred_bbox_deltas, const float* labels, const float* regression_targets, int64_t total,
                      const int64_t* pos_idx, int32_t num_pos, const int64_t* sampled_idx, int32_t num_sampled, float* losses, float* grad_objectness,
                      float* grad_deltas, void* stream) {
  if (total <= 0 || num_pos < 0 || num_sampled <= 0 || num_pos > num_sampled)
    return fail(MI355DET_EINVAL, "%s: need total > 0 and 0 <= num_pos <= num_sampled, num_sampled > 0", "rpn_loss");
  if (!objectness || !pred_bbox_deltas || !labels || !regression_targets || !sampled_idx || (num_pos && !pos_idx) || !losses || !grad_objectness || !grad_deltas)
    return fail(MI355DET_EINVAL, "%s: null argument", "rpn_loss");
  if (hipMemsetAsync(grad_objectness, 0, sizeof(float) * (size_t)total, S(stream)) != hipSuccess ||
      hipMemsetAsync(grad_deltas, 0, sizeof(float) * 4 * (size_t)total, S(stream)) != hipSuccess)
    return fail(MI355DET_ELAUNCH, "%s: memset failed", "rpn_loss");
  hipLaunchKernelGGL(rpn_loss_kernel, dim3(1), dim3(1024), 0, S(stream), objectness, pred_bbox_deltas, labels, regression_targets,
                     (const long long*)pos_idx, num_pos, (const long long*)sampled_idx, num_sampled, losses, grad_objectness, grad_deltas);
  return check_launch("rpn_loss");
}

size_t mi355det_roi_detections_workspace(int32_t n_images, int32_t max_proposals, int32_t num_classes, int32_t max_candidates) {
  RoiDetWs W;
  if (max_proposals <= 0 || num_classes <= 0 || roi_det_layout(n_images, (long long)max_proposals * num_classes, max_candidates, W)) return 0;
  return W.total;
}

int mi355det_roi_detections(const float* scores, const float* box_regression, const float* proposals, const float* clip_limits, int32_t n_images,
                            int32_t max_proposals, int32_t num_classes, float score_thresh, int32_t max_candidates, float wx, float wy, float ww,
                            float wh, float xform_clip, float min_size, float nms_thresh, int32_t detections_per_img, float* out_boxes,
                            float* out_scores, int64_t* out_labels, int32_t* out_counts, int32_t* candidate_counts, void* workspace,
                            size_t workspace_bytes, void* stream) {
  RoiDetWs W;
  if (max_proposals <= 0 || num_classes <= 0 || roi_det_layout(n_images, (long long)max_proposals * num_classes, max_candidates, W))
    return fail(MI355DET_EINVAL, "%s: need a positive batch, 1 <= max_candidates <= min(16384, proposals x classes)", "roi_detections");
  if (detections_per_img <= 0 || !scores || !box_regression || !proposals || !clip_limits || !out_boxes || !out_scores || !out_labels || !out_counts ||
      !candidate_counts || !workspace)
    return fail(MI355DET_EINVAL, "%s: null argument or detections_per_img <= 0", "roi_detections");
  if (workspace_bytes < W.total) return fail(MI355DET_EWORKSPACE, "%s: workspace too small", "roi_detections");
  char* ws = (char*)workspace;
  const long long row = (long long)max_proposals * num_classes;
  if (int e = mi355det_topk_ws(scores, n_images, row, row, max_candidates, score_thresh, (int64_t*)(ws + W.idx), (float*)(ws + W.val), candidate_counts,
                               ws + W.topk_ws, mi355det_topk_workspace(n_images), stream))
    return e;
  const long long total = (long long)n_images * max_candidates;
  hipLaunchKernelGGL(roi_det_select_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, S(stream), (const long long*)(ws + W.idx),
                     (const float*)(ws + W.val), (const int*)candidate_counts, max_candidates, max_proposals, num_classes, box_regression, proposals,
                     clip_limits, n_images, wx, wy, ww, wh, xform_clip, min_size, (float*)(ws + W.boxes), (float*)(ws + W.masked),
                     (float*)(ws + W.scores), (long long*)(ws + W.labels));
  if (int e = mi355det_nms_batch((const float*)(ws + W.boxes), (const float*)(ws + W.masked), (const int64_t*)(ws + W.labels), n_images, max_candidates,
                                 nms_thresh, (int64_t*)(ws + W.keep), (int32_t*)(ws + W.keep_cnt), ws + W.nms_ws,
                                 mi355det_nms_workspace(n_images, max_candidates), stream))
    return e;
  hipLaunchKernelGGL(retina_gather_kernel, dim3(n_images), dim3(256), 0, S(stream), (const float*)(ws + W.boxes), (const float*)(ws + W.masked),
                     (const float*)(ws + W.scores), (const long long*)(ws + W.labels), (const long long*)(ws + W.keep), (const int*)(ws + W.keep_cnt),
                     max_candidates, detections_per_img, out_boxes, out_scores, (long long*)out_labels, out_counts);
  return check_launch("roi_detections");
}

}  // extern "C"
