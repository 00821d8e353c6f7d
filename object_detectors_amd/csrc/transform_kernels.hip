// Input-side transform of the detection step: GeneralizedRCNNTransform (torchvision_models/tvision/transform.py:88-226:
// normalize -> bilinear resize (align_corners=False) -> zero-pad into the batch tensor), resize_boxes (:279-293) and the YOLO multi-scale
// F.interpolate (yolo/procedures/train_one_epoch.py:64-69).  HBM-bound: every output element is written once, every input element read ~once
// (the four taps of neighbouring outputs hit the same lines).
#include "common.h"

using namespace mi355;

namespace {

// out[p][y][x] for planes p of one image (or of a same-size batch): bilinear sample of (in - mean[p % c]) / std[p % c], zero outside [oh, ow].
// Index and weight arithmetic follow ATen's upsample_bilinear2d (area_pixel_compute_source_index, align_corners = false):
//   src = max(scale * (dst + 0.5) - 0.5, 0), i0 = (int)src, i1 = i0 + (i0 < in - 1), l1 = src - i0, l0 = 1 - l1
//   val = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, int planes, int c, int h, int w, const float* __restrict__ mean,
                                                              const float* __restrict__ stdv, float* __restrict__ out, int oh, int ow, int ph, int pw,
                                                              float rh, float rw) {
  const long long total = (long long)planes * ph * pw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % pw);
    const long long t = i / pw;
    const int y = (int)(t % ph), p = (int)(t / ph);
    float v = 0.0f;
    if (y < oh && x < ow) {
      const float sy = fmaxf(rh * ((float)y + 0.5f) - 0.5f, 0.0f), sx = fmaxf(rw * ((float)x + 0.5f) - 0.5f, 0.0f);
      const int y0 = (int)sy, x0 = (int)sx;
      const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
      const float h1 = sy - (float)y0, h0 = 1.0f - h1, w1 = sx - (float)x0, w0 = 1.0f - w1;
      const float* src = in + (size_t)p * h * w;
      float v00 = src[(size_t)y0 * w + x0], v01 = src[(size_t)y0 * w + x1], v10 = src[(size_t)y1 * w + x0], v11 = src[(size_t)y1 * w + x1];
      if (mean) {
        const float m = mean[p % c], s = stdv[p % c];
        v00 = (v00 - m) / s;
        v01 = (v01 - m) / s;
        v10 = (v10 - m) / s;
        v11 = (v11 - m) / s;
      }
      v = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
    }
    out[i] = v;
  }
}

__global__ void resize_boxes_kernel(const float* __restrict__ boxes, float* __restrict__ out, long long n, float ratio_h, float ratio_w) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 4) return;
  out[i] = boxes[i] * ((i & 1) ? ratio_h : ratio_w);       // (xmin, ymin, xmax, ymax): x by the width ratio, y by the height ratio
}

}  // namespace

extern "C" {

int mi355det_resize_bilinear(const float* in, int32_t planes, int32_t c, int32_t h, int32_t w, const float* mean, const float* stdv, float* out,
                             int32_t out_h, int32_t out_w, int32_t pad_h, int32_t pad_w, void* stream) {
  if (planes <= 0 || c <= 0 || h <= 0 || w <= 0 || out_h <= 0 || out_w <= 0 || pad_h < out_h || pad_w < out_w)
    return fail(MI355DET_EINVAL, "%s: bad geometry (need planes, sizes > 0 and pad >= out)", "resize_bilinear");
  if ((mean == nullptr) != (stdv == nullptr)) return fail(MI355DET_EINVAL, "%s: mean and std come together", "resize_bilinear");
  const float rh = (float)h / (float)out_h, rw = (float)w / (float)out_w;      // recompute_scale_factor / size= : scale from the two sizes
  const long long total = (long long)planes * pad_h * pad_w;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3((int)min((long long)8192, (total + 255) / 256)), dim3(256), 0, S(stream), in, planes, c, h, w, mean, stdv,
                     out, out_h, out_w, pad_h, pad_w, rh, rw);
  return check_launch("resize_bilinear");
}

int mi355det_resize_boxes(const float* boxes, float* out, int64_t n, int32_t orig_h, int32_t orig_w, int32_t new_h, int32_t new_w, void* stream) {
  if (n < 0 || orig_h <= 0 || orig_w <= 0) return fail(MI355DET_EINVAL, "%s: bad sizes", "resize_boxes");
  if (n == 0) return 0;
  const float rh = (float)new_h / (float)orig_h, rw = (float)new_w / (float)orig_w;
  hipLaunchKernelGGL(resize_boxes_kernel, dim3((int)((n * 4 + 255) / 256)), dim3(256), 0, S(stream), boxes, out, (long long)n, rh, rw);
  return check_launch("resize_boxes");
}

}  // extern "C"
