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

// ---- output side: kept detections -> the numbers of the COCO result dicts -------------------------------------
// yolo/procedures/test_one_epoch.py:41-66: rows [k,6] = (x1,y1,x2,y2,score,label) in network pixels -> bbox (xmin, ymin, w, h) in the
// ORIGINAL image's pixels (x / inp_dim * W, y / inp_dim * H), area = w*h, category id (80 -> 91 COCO map, or label + 1).
// torchvision_models/detection/coco_eval.py:83-105,169-171: convert_to_xywh = the same with inp_dim = W = H = 1 and labels untouched.
namespace {

__constant__ int c_coco80to91[80] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 27, 28, 31, 32, 33, 34,
                                     35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63,
                                     64, 65, 67, 70, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 84, 85, 86, 87, 88, 89, 90};

__global__ void coco_rows_kernel(const float* __restrict__ boxes, int box_ld, const float* __restrict__ labels_f, const long long* __restrict__ labels_i,
                                 int lab_ld, long long k, float inp_dim, float img_h, float img_w, int scale, int label_mode, float* __restrict__ bbox,
                                 float* __restrict__ area, long long* __restrict__ cat) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const float* r = boxes + i * box_ld;
  float xmin = r[0], ymin = r[1], xmax = r[2], ymax = r[3];
  if (scale) {
    xmin = xmin / inp_dim * img_w;
    ymin = ymin / inp_dim * img_h;
    xmax = xmax / inp_dim * img_w;
    ymax = ymax / inp_dim * img_h;
  }
  const float w = xmax - xmin, h = ymax - ymin;
  bbox[i * 4 + 0] = xmin;
  bbox[i * 4 + 1] = ymin;
  bbox[i * 4 + 2] = w;
  bbox[i * 4 + 3] = h;
  if (area) area[i] = w * h;
  if (cat) {
    const long long l = labels_i ? labels_i[i * lab_ld] : (long long)labels_f[i * lab_ld];     // (atrbs[:,5]).long()
    cat[i] = label_mode == 1 ? (long long)c_coco80to91[l < 0 ? 0 : (l > 79 ? 79 : l)] : label_mode == 0 ? l + 1 : l;
  }
}

}  // namespace

extern "C" int mi355det_coco_rows(const float* boxes, int32_t box_ld, const float* labels_f32, const int64_t* labels_i64, int32_t label_ld, int64_t k,
                                  float inp_dim, float img_h, float img_w, int32_t scale, int32_t label_mode, float* bbox_xywh, float* area,
                                  int64_t* category_id, void* stream) {
  if (k < 0 || box_ld < 4 || (scale && !(inp_dim > 0.f))) return fail(MI355DET_EINVAL, "%s: bad arguments", "coco_rows");
  if (label_mode < 0 || label_mode > 2) return fail(MI355DET_EINVAL, "%s: label_mode 0 (+1), 1 (COCO 80->91) or 2 (as is)", "coco_rows");
  if (category_id && !labels_f32 && !labels_i64) return fail(MI355DET_EINVAL, "%s: category ids need labels", "coco_rows");
  if (k == 0) return 0;
  hipLaunchKernelGGL(coco_rows_kernel, dim3((int)((k + 255) / 256)), dim3(256), 0, S(stream), boxes, box_ld, labels_f32, (const long long*)labels_i64,
                     label_ld, (long long)k, inp_dim, img_h, img_w, scale, label_mode, bbox_xywh, area, (long long*)category_id);
  return check_launch("coco_rows");
}
