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

}  // extern "C"
