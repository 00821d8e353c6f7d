// Fast R-CNN box-head loss, forward + gradients in one pass (reference: torchvision_models/tvision/roi_heads.py:24-96, called at
// :826-827 as fastrcnn_loss(tfidf * class_logits, box_regression, labels, regression_targets, weights=classification_weights,
// loss_type=...)).  HBM-bound: reads the [N,K] logits and the [N,4K] box deltas once, writes their gradients once.
//
//   loss_type 0 'ce'          F.cross_entropy(x, labels, weight=w)  = sum_i w[y_i] (lse_i - x_i,y_i) / sum_i w[y_i]
//             1 'bce'         BCE-with-logits against the one-hot target with the background column zeroed, sum / N
//             2 'focal_loss'  torchvision sigmoid_focal_loss (alpha 0.25, gamma 2), sum / N
//             3 'gombit'      c = clamp(x - 1.96, -3, 5), p = exp(-exp(-c)), BCE(p, y) sum / N, and "/= 4 when the loss exceeds 5" (:71-72)
//             4 'gombit_fl'   the same with the focal factor alpha_t (1 - p_t)^2
//   box loss: smooth-L1 (beta 1) over the 4 deltas of the labelled class of every positive row, sum / N (:84-93)
// x = class_scale[k] * logits (the tf-idf row, :826); gradients are returned w.r.t. the UNSCALED logits.
//
// One wave per row (K = 91 .. 1204): lanes stride over the classes, row reductions by DPP/shuffle, per-row losses land in a workspace
// and are summed by one workgroup in a fixed order (deterministic); the gombit "/4" rescale, which depends on the total, is applied to
// the already written gradients by a third launch only for that loss type.
#include "common.h"

using namespace mi355;

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float bce_logits(float x, float t) { return fmaxf(x, 0.0f) - x * t + log1pf(expf(-fabsf(x))); }

template <int TYPE>
__global__ __launch_bounds__(256) void frcnn_loss_kernel(const float* __restrict__ logits, const float* __restrict__ breg, const long long* __restrict__ labels,
                                                          const float* __restrict__ tgt, const float* __restrict__ cscale, const float* __restrict__ cw,
                                                          int n, int k, const float* __restrict__ wsum, float* __restrict__ row_cls,
                                                          float* __restrict__ row_box, float* __restrict__ glogits, float* __restrict__ gbox) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int y = (int)labels[row];
  const float* xr = logits + (size_t)row * k;
  float* gr = glogits ? glogits + (size_t)row * k : nullptr;
  const float inv_n = 1.0f / (float)n;
  float loss = 0.0f;
  if (TYPE == 0) {
    float mx = -INFINITY;
    for (int c = lane; c < k; c += 64) mx = fmaxf(mx, xr[c] * (cscale ? cscale[c] : 1.0f));
    mx = wave_max(mx);
    float se = 0.0f;
    for (int c = lane; c < k; c += 64) se += expf(xr[c] * (cscale ? cscale[c] : 1.0f) - mx);
    se = wave_sum(se);
    const float wy = cw ? cw[y] : 1.0f;
    const float norm = wy / wsum[0];
    const float xy = xr[y] * (cscale ? cscale[y] : 1.0f);
    loss = norm * ((mx + logf(se)) - xy);
    if (gr) {
      const float inv_se = 1.0f / se;
      for (int c = lane; c < k; c += 64) {
        const float s = cscale ? cscale[c] : 1.0f;
        const float p = expf(xr[c] * s - mx) * inv_se;
        gr[c] = norm * (p - (c == y ? 1.0f : 0.0f)) * s;
      }
    }
  } else {
    for (int c = lane; c < k; c += 64) {
      const float s = cscale ? cscale[c] : 1.0f;
      const float x = xr[c] * s;
      const float t = (c == y && c != 0) ? 1.0f : 0.0f;     // y_onehot[:, 0] = 0 (:52)
      float l, g;
      if (TYPE == 1) {
        l = bce_logits(x, t);
        g = sigmoidf_(x) - t;
      } else if (TYPE == 2) {
        const float p = sigmoidf_(x);
        const float ce = bce_logits(x, t);
        const float pt = p * t + (1.0f - p) * (1.0f - t);
        const float q = 1.0f - pt;
        const float at = 0.25f * t + 0.75f * (1.0f - t);
        l = at * (ce * (q * q));
        const float dpt = (2.0f * t - 1.0f) * p * (1.0f - p);
        g = at * (q * q * (p - t) - 2.0f * q * dpt * ce);
      } else {
        const float u = x - 1.96f;
        const float c5 = fminf(fmaxf(u, -3.0f), 5.0f);
        const bool pass = u >= -3.0f && u <= 5.0f;            // clamp passes the gradient on the closed interval
        const float e = expf(-c5);
        const float pe = expf(-e);
        // F.binary_cross_entropy clamps its logs at -100: log(pe) = -e >= -e^3, log(1 - pe) >= log(1 - exp(-e^-5)) ~ -5: never reached
        const float l1p = logf(1.0f - pe);
        const float b = t > 0.5f ? e : -l1p;
        const float dpe = pe * e;                             // d pe / d c
        const float db = t > 0.5f ? -1.0f / pe : 1.0f / (1.0f - pe);   // d bce / d pe
        if (TYPE == 3) {
          l = b;
          g = db * dpe;
        } else {
          const float pt = pe * t + (1.0f - pe) * (1.0f - t);
          const float q = 1.0f - pt;
          const float at = 0.25f * t + 0.75f * (1.0f - t);
          l = at * (b * (q * q));
          const float dq = -(2.0f * t - 1.0f);                // d q / d pe
          g = at * (2.0f * q * dq * b + q * q * db) * dpe;
        }
        if (!pass) g = 0.0f;
      }
      loss += l;
      if (gr) gr[c] = g * inv_n * s;
    }
    loss = wave_sum(loss) * inv_n;
  }
  // ---- box regression: smooth-L1 on the labelled class of positive rows, zeros elsewhere
  float lb = 0.0f;
  const float* br = breg + (size_t)row * 4 * k;
  float* gb = gbox ? gbox + (size_t)row * 4 * k : nullptr;
  for (int c = lane; c < 4 * k; c += 64) {
    float g = 0.0f;
    if (y > 0 && (c >> 2) == y) {
      const float d = br[c] - tgt[(size_t)row * 4 + (c & 3)];
      const float a = fabsf(d);
      lb += a < 1.0f ? 0.5f * d * d : a - 0.5f;
      g = (a < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f)) * inv_n;
    }
    if (gb) gb[c] = g;
  }
  lb = wave_sum(lb) * inv_n;
  if (lane == 0) {
    row_cls[row] = loss;
    row_box[row] = lb;
  }
}

// sum of the class weights of the labels (the 'mean' normaliser of a weighted cross entropy); one workgroup, fixed order
__global__ __launch_bounds__(256) void frcnn_wsum_kernel(const long long* __restrict__ labels, const float* __restrict__ cw, int n, float* __restrict__ wsum) {
  __shared__ float sm[256];
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) s += cw ? cw[labels[i]] : 1.0f;
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) wsum[0] = sm[0];
}

__global__ __launch_bounds__(256) void frcnn_finalize_kernel(const float* __restrict__ row_cls, const float* __restrict__ row_box, int n, int type,
                                                              float* __restrict__ losses, float* __restrict__ rescale) {
  __shared__ float s1[256], s2[256];
  float a = 0.0f, b = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) {
    a += row_cls[i];
    b += row_box[i];
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      s1[threadIdx.x] += s1[threadIdx.x + o];
      s2[threadIdx.x] += s2[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float c = s1[0];
    float r = 1.0f;
    if (type == 3 && c > 5.0f) {     // roi_heads.py:71-72
      c *= 0.25f;
      r = 0.25f;
    }
    losses[0] = c;
    losses[1] = s2[0];
    rescale[0] = r;
  }
}

__global__ __launch_bounds__(256) void frcnn_rescale_kernel(float* __restrict__ g, long long total, const float* __restrict__ rescale) {
  const float r = rescale[0];
  if (r == 1.0f) return;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) g[i] *= r;
}

}  // namespace

extern "C" {

size_t mi355det_fastrcnn_loss_workspace(int32_t n) { return (size_t)(2 * (n > 0 ? n : 0) + 8) * sizeof(float); }

int mi355det_fastrcnn_loss(const float* class_logits, const float* box_regression, const int64_t* labels, const float* regression_targets,
                           const float* class_scale, const float* class_weights, int32_t n, int32_t k, int32_t loss_type, float* losses,
                           float* grad_logits, float* grad_box, void* workspace, size_t workspace_bytes, void* stream) {
  if (n <= 0 || k <= 1) return fail(MI355DET_EINVAL, "%s: need n >= 1 rows and k >= 2 classes", "fastrcnn_loss");
  if (loss_type < 0 || loss_type > 4) return fail(MI355DET_EINVAL, "%s: loss_type must be 0..4 (ce, bce, focal_loss, gombit, gombit_fl)", "fastrcnn_loss");
  if (workspace_bytes < mi355det_fastrcnn_loss_workspace(n)) return fail(MI355DET_EWORKSPACE, "%s: workspace too small", "fastrcnn_loss");
  if (class_weights && loss_type != 0) return fail(MI355DET_EINVAL, "%s: class weights apply to 'ce' only (roi_heads.py:45-46)", "fastrcnn_loss");
  float* row_cls = (float*)workspace;
  float* row_box = row_cls + n;
  float* wsum = row_box + n;
  float* rescale = wsum + 1;
  const long long* lab = (const long long*)labels;
  const int blocks = (n + 3) / 4;
  hipStream_t st = S(stream);
  if (loss_type == 0) hipLaunchKernelGGL(frcnn_wsum_kernel, dim3(1), dim3(256), 0, st, lab, class_weights, n, wsum);
#define LAUNCH_T(T)                                                                                                                         \
  hipLaunchKernelGGL(frcnn_loss_kernel<T>, dim3(blocks), dim3(256), 0, st, class_logits, box_regression, lab, regression_targets, class_scale, \
                     class_weights, n, k, wsum, row_cls, row_box, grad_logits, grad_box)
  switch (loss_type) {
    case 0: LAUNCH_T(0); break;
    case 1: LAUNCH_T(1); break;
    case 2: LAUNCH_T(2); break;
    case 3: LAUNCH_T(3); break;
    default: LAUNCH_T(4); break;
  }
#undef LAUNCH_T
  hipLaunchKernelGGL(frcnn_finalize_kernel, dim3(1), dim3(256), 0, st, row_cls, row_box, n, loss_type, losses, rescale);
  if (loss_type == 3 && grad_logits) {
    const long long total = (long long)n * k;
    hipLaunchKernelGGL(frcnn_rescale_kernel, dim3((int)min((long long)2048, (total + 255) / 256)), dim3(256), 0, st, grad_logits, total, rescale);
  }
  return check_launch("fastrcnn_loss");
}

}  // extern "C"
