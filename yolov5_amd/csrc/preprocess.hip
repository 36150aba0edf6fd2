// Input pre-processing on the device -- for a whole batch in one launch:
//     letterbox(im, new_shape, auto=False)            utils/augmentations.py:85-115: cv2.resize(INTER_LINEAR) + 114 border
//     im.transpose((2, 0, 1))[::-1]                   HWC -> CHW, BGR -> RGB (utils/dataloaders.py LoadImages, detect.py:205)
//     im.half() / 255                                 detect.py:208-209, val.py:261-262, models/common.py:926
// The geometry (ratio, new_unpad, top/left) is computed by the host mirror of `letterbox` and arrives per image in a job
// table; this kernel does the pixel work.  cv2 is a third-party dependency of the reference (opencv-python, absent here):
// the resize restates OpenCV's published 8-bit INTER_LINEAR algorithm (modules/imgproc/src/resize.cpp):
//   * scale = 1 / (dst / src) in double; source position f = (float)((d + 0.5) * scale - 0.5), s = floor(f), f -= s;
//     horizontally s < 0 -> (s, f) = (0, 0) and s >= w - 1 -> (w - 1, 0); vertically rows are clamped, f kept;
//   * weights are 11-bit fixed point: short(rint(w * 2048)) (round half to even);
//   * horizontal pass in int32: h = S[s] * a0 + S[s + 1] * a1; vertical pass
//     out = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
//   * an exact 2x down-scale in both directions is re-routed to INTER_AREA: out = (p00 + p01 + p10 + p11 + 2) >> 2.
// HBM-bound byte work: one lane produces 8 consecutive output pixels of one row (3 channels) and stores 16 bytes per
// channel plane (fp16 CHW) -- reads are gathers from at most two source rows that stay in L2.
#include <hip/hip_runtime.h>

#include "../../include/yolov5_hip.h"
#include "y5_common.h"
#include "y5_host.h"
#include "resize_u8.h"

namespace {
constexpr int PX = 8;  // output pixels per lane

struct LbParams {
  const y5_letterbox_job* jobs;
  void* dst;
  int B, H, W, pad, swap_rb, dst_dtype, chw, div255;
};

}  // namespace

__global__ __launch_bounds__(256)
void y5_letterbox_kernel(const LbParams p) {
  const int b = blockIdx.y;
  const int nxg = (p.W + PX - 1) / PX;  // lanes per output row
  const int id = blockIdx.x * 256 + threadIdx.x;
  const int y = id / nxg;
  if (y >= p.H) return;
  const int x0 = (id - y * nxg) * PX;
  const y5_letterbox_job j = p.jobs[b];
  const unsigned char* src = static_cast<const unsigned char*>(j.src);
  const int yy = y - j.top;
  const bool row_in = yy >= 0 && yy < j.nh;
  const bool resize = j.nw != j.w0 || j.nh != j.h0;
  const double sx = 1.0 / ((double)j.nw / (double)j.w0), sy = 1.0 / ((double)j.nh / (double)j.h0);
  const bool area2 = resize && is_int_scale(sx, 2) && is_int_scale(sy, 2);
  Axis ay{};
  if (row_in && resize && !area2) ay = axis_y(yy, sy, j.h0);
  unsigned char v[3][PX];
#pragma unroll
  for (int i = 0; i < PX; ++i) {
    const int xx = x0 + i - j.left;
    int c0 = p.pad, c1 = p.pad, c2 = p.pad;
    if (row_in && xx >= 0 && xx < j.nw) {
      if (!resize) {
        const unsigned char* s = src + (size_t)yy * j.stride + xx * 3;
        c0 = s[0]; c1 = s[1]; c2 = s[2];
      } else if (area2) {
        const unsigned char* s = src + (size_t)(2 * yy) * j.stride + (2 * xx) * 3;
        const unsigned char* t = s + j.stride;
        c0 = (s[0] + s[3] + t[0] + t[3] + 2) >> 2;
        c1 = (s[1] + s[4] + t[1] + t[4] + 2) >> 2;
        c2 = (s[2] + s[5] + t[2] + t[5] + 2) >> 2;
      } else {
        const Axis ax = axis_x(xx, sx, j.w0);
        const unsigned char* r0 = src + (size_t)ay.s0 * j.stride;
        const unsigned char* r1 = src + (size_t)ay.s1 * j.stride;
        int o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int h0 = r0[ax.s0 * 3 + c] * ax.w0 + r0[ax.s1 * 3 + c] * ax.w1;
          const int h1 = r1[ax.s0 * 3 + c] * ax.w0 + r1[ax.s1 * 3 + c] * ax.w1;
          o[c] = (((ay.w0 * (h0 >> 4)) >> 16) + ((ay.w1 * (h1 >> 4)) >> 16) + 2) >> 2;
        }
        c0 = o[0]; c1 = o[1]; c2 = o[2];
      }
    }
    v[0][i] = (unsigned char)(p.swap_rb ? c2 : c0);
    v[1][i] = (unsigned char)c1;
    v[2][i] = (unsigned char)(p.swap_rb ? c0 : c2);
  }
  const int nvalid = p.W - x0 < PX ? p.W - x0 : PX;
  const size_t plane = (size_t)p.H * p.W;
  if (p.dst_dtype == Y5_U8) {
    unsigned char* d = static_cast<unsigned char*>(p.dst) + (size_t)b * 3 * plane;
    for (int i = 0; i < nvalid; ++i)
      for (int c = 0; c < 3; ++c) {
        const size_t o = p.chw ? (size_t)c * plane + (size_t)y * p.W + x0 + i : ((size_t)y * p.W + x0 + i) * 3 + c;
        d[o] = v[c][i];
      }
  } else if (p.dst_dtype == Y5_F16) {
    _Float16* d = static_cast<_Float16*>(p.dst) + (size_t)b * 3 * plane;
    for (int c = 0; c < 3; ++c) {
      _Float16 h[PX];
#pragma unroll
      for (int i = 0; i < PX; ++i) {
        // `.half() / 255`: the division is carried out in fp32 on the half value (exact for 0..255) and rounded once
        const float f = (float)v[c][i];
        h[i] = (_Float16)(p.div255 ? f / 255.0f : f);
      }
      if (p.chw && nvalid == PX && (p.W & 7) == 0) {
        half8_t hv;
        for (int i = 0; i < PX; ++i) hv[i] = h[i];
        *reinterpret_cast<half8_t*>(d + (size_t)c * plane + (size_t)y * p.W + x0) = hv;
      } else {
        for (int i = 0; i < nvalid; ++i) {
          const size_t o = p.chw ? (size_t)c * plane + (size_t)y * p.W + x0 + i : ((size_t)y * p.W + x0 + i) * 3 + c;
          d[o] = h[i];
        }
      }
    }
  } else {
    float* d = static_cast<float*>(p.dst) + (size_t)b * 3 * plane;
    for (int i = 0; i < nvalid; ++i)
      for (int c = 0; c < 3; ++c) {
        const size_t o = p.chw ? (size_t)c * plane + (size_t)y * p.W + x0 + i : ((size_t)y * p.W + x0 + i) * 3 + c;
        const float f = (float)v[c][i];
        d[o] = p.div255 ? f / 255.0f : f;
      }
  }
}

extern "C" int y5_letterbox_batch(const y5_letterbox_job* jobs_dev, int B, int H, int W, int pad_value, int swap_rb, void* dst, int dst_dtype,
                                  int dst_chw, int div255, void* stream_) {
  if (!jobs_dev || !dst) return y5_fail(Y5_ERR_BAD_ARG, "letterbox_batch: null pointer");
  if (B < 1 || B > 65535 || H < 1 || W < 1 || pad_value < 0 || pad_value > 255)
    return y5_fail(Y5_ERR_BAD_ARG, "letterbox_batch: need 1 <= B <= 65535, H, W >= 1, 0 <= pad_value <= 255");
  if (dst_dtype != Y5_U8 && dst_dtype != Y5_F16 && dst_dtype != Y5_F32) return y5_fail(Y5_ERR_BAD_ARG, "letterbox_batch: dst dtype must be u8, f16 or f32");
  LbParams p{};
  p.jobs = jobs_dev; p.dst = dst; p.B = B; p.H = H; p.W = W; p.pad = pad_value; p.swap_rb = swap_rb; p.dst_dtype = dst_dtype;
  p.chw = dst_chw; p.div255 = div255;
  const long long lanes = (long long)H * ((W + PX - 1) / PX);
  if (lanes > 0x7fffffffLL) return y5_fail(Y5_ERR_UNSUPPORTED, "letterbox_batch: output too large");
  hipLaunchKernelGGL(y5_letterbox_kernel, dim3((unsigned)((lanes + 255) / 256), B), dim3(256), 0, static_cast<hipStream_t>(stream_), p);
  return y5_check_launch("y5_letterbox_batch");
}

// ---- test-time augmentation glue (models/yolo.py:269-312 `_forward_augment`) ---------------------------------------------------------------
// y5_scale_img: utils/torch_utils.py `scale_img(img.flip(f), ratio, gs)` for a whole NCHW batch in one launch: optional flip of the SOURCE (2 = up-down,
// 3 = left-right, as torch's dim index), F.interpolate(size = (OH, OW), mode = "bilinear", align_corners = False) -- source index
// (dst + 0.5) * (in / out) - 0.5 clamped at 0, the +1 neighbour clamped at the border, weights in fp32 -- and F.pad to (PH, PW) with `pad_value`
// (0.447, the ImageNet mean) on the right / bottom.  src u8 is read as x / 255 (the `im.half() / 255` of val.py:262 folded in).
namespace {
struct ScaleParams {
  const void* src; void* dst;
  int src_dtype, dst_dtype, B, C, H, W, OH, OW, PH, PW, flip;
  float pad, sh, sw;
};
template <typename T> __device__ inline float load_px(const void* p, size_t i);
template <> __device__ inline float load_px<unsigned char>(const void* p, size_t i) { return (float)static_cast<const unsigned char*>(p)[i] / 255.0f; }
template <> __device__ inline float load_px<_Float16>(const void* p, size_t i) { return (float)static_cast<const _Float16*>(p)[i]; }
template <> __device__ inline float load_px<float>(const void* p, size_t i) { return static_cast<const float*>(p)[i]; }

template <typename TS, typename TD>
__global__ __launch_bounds__(256)
void y5_scale_img_kernel(const ScaleParams p) {
  const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)p.PH * p.PW;
  if (id >= per) return;
  const int plane = blockIdx.y;                     // b * C + c
  const int oy = (int)(id / p.PW), ox = (int)(id - (long long)oy * p.PW);
  float v = p.pad;
  if (oy < p.OH && ox < p.OW) {
    float fy = p.sh * ((float)oy + 0.5f) - 0.5f, fx = p.sw * ((float)ox + 0.5f) - 0.5f;   // area_pixel_compute_source_index, align_corners = False
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < p.H - 1 ? 1 : 0), x1 = x0 + (x0 < p.W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    auto at = [&](int y, int x) {
      if (p.flip == 2) y = p.H - 1 - y;
      if (p.flip == 3) x = p.W - 1 - x;
      return load_px<TS>(p.src, ((size_t)plane * p.H + y) * p.W + x);
    };
    v = (1.f - ly) * ((1.f - lx) * at(y0, x0) + lx * at(y0, x1)) + ly * ((1.f - lx) * at(y1, x0) + lx * at(y1, x1));
  }
  static_cast<TD*>(p.dst)[(size_t)plane * per + id] = (TD)v;
}

struct DescaleParams { void* z; long long rows; int no; float inv_scale, img_h, img_w; int flip; };
// models/yolo.py:283-299 `_descale_pred` (inplace): p[..., :4] /= scale; flip 2: y = img_h - y; flip 3: x = img_w - x
template <typename T>
__global__ __launch_bounds__(256)
void y5_tta_descale_kernel(const DescaleParams p) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= p.rows) return;
  T* q = static_cast<T*>(p.z) + r * p.no;
  float x = (float)q[0] / p.inv_scale, y = (float)q[1] / p.inv_scale;   // (inv_scale holds the scale: a division, like the reference)
  const float w = (float)q[2] / p.inv_scale, h = (float)q[3] / p.inv_scale;
  if (p.flip == 2) y = p.img_h - y;
  if (p.flip == 3) x = p.img_w - x;
  q[0] = (T)x; q[1] = (T)y; q[2] = (T)w; q[3] = (T)h;
}
}  // namespace

extern "C" int y5_scale_img(const void* src, int src_dtype, int B, int Cn, int H, int W, int flip, int OH, int OW, int PH, int PW, float pad_value,
                            void* dst, int dst_dtype, void* stream_) {
  if (!src || !dst) return y5_fail(Y5_ERR_BAD_ARG, "scale_img: null pointer");
  if (B < 1 || Cn < 1 || H < 1 || W < 1 || OH < 1 || OW < 1 || PH < OH || PW < OW || (long long)B * Cn > 65535)
    return y5_fail(Y5_ERR_BAD_ARG, "scale_img: bad sizes (need PH >= OH, PW >= OW, B * C <= 65535)");
  if (flip != 0 && flip != 2 && flip != 3) return y5_fail(Y5_ERR_BAD_ARG, "scale_img: flip must be 0, 2 (up-down) or 3 (left-right)");
  if ((src_dtype != Y5_U8 && src_dtype != Y5_F16 && src_dtype != Y5_F32) || (dst_dtype != Y5_F16 && dst_dtype != Y5_F32))
    return y5_fail(Y5_ERR_BAD_ARG, "scale_img: src u8 / f16 / f32, dst f16 / f32");
  ScaleParams p{};
  p.src = src; p.dst = dst; p.src_dtype = src_dtype; p.dst_dtype = dst_dtype; p.B = B; p.C = Cn; p.H = H; p.W = W; p.OH = OH; p.OW = OW;
  p.PH = PH; p.PW = PW; p.flip = flip; p.pad = pad_value; p.sh = (float)H / (float)OH; p.sw = (float)W / (float)OW;
  const dim3 g((unsigned)(((long long)PH * PW + 255) / 256), (unsigned)(B * Cn));
  hipStream_t st = static_cast<hipStream_t>(stream_);
#define Y5_SI(TS, TD) hipLaunchKernelGGL((y5_scale_img_kernel<TS, TD>), g, dim3(256), 0, st, p)
  if (dst_dtype == Y5_F16) {
    if (src_dtype == Y5_U8) Y5_SI(unsigned char, _Float16); else if (src_dtype == Y5_F16) Y5_SI(_Float16, _Float16); else Y5_SI(float, _Float16);
  } else {
    if (src_dtype == Y5_U8) Y5_SI(unsigned char, float); else if (src_dtype == Y5_F16) Y5_SI(_Float16, float); else Y5_SI(float, float);
  }
#undef Y5_SI
  return y5_check_launch("y5_scale_img");
}

extern "C" int y5_tta_descale(void* z, int dtype, long long rows, int no, float scale, int flip, float img_h, float img_w, void* stream_) {
  if (!z || rows < 1 || no < 4 || !(scale > 0.f)) return y5_fail(Y5_ERR_BAD_ARG, "tta_descale: bad args");
  if (flip != 0 && flip != 2 && flip != 3) return y5_fail(Y5_ERR_BAD_ARG, "tta_descale: flip must be 0, 2 or 3");
  if (dtype != Y5_F16 && dtype != Y5_F32) return y5_fail(Y5_ERR_BAD_ARG, "tta_descale: dtype must be f16 or f32");
  DescaleParams p{z, rows, no, scale, img_h, img_w, flip};
  const dim3 g((unsigned)((rows + 255) / 256));
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (dtype == Y5_F16) hipLaunchKernelGGL(y5_tta_descale_kernel<_Float16>, g, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(y5_tta_descale_kernel<float>, g, dim3(256), 0, st, p);
  return y5_check_launch("y5_tta_descale");
}
