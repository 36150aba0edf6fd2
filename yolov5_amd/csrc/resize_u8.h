// OpenCV's 8-bit INTER_LINEAR resize arithmetic as device helpers (shared by the letterbox kernel, preprocess.hip, and the mosaic /
// augmentation kernel, augment.hip).  cv2 is a third-party dependency of the reference (opencv-python, absent here): restated from the
// published algorithm (modules/imgproc/src/resize.cpp), see the header of preprocess.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace {
struct Axis { int s0, s1, w0, w1; };  // two taps and their 11-bit weights

__device__ inline int rint_short(float v) {  // saturate_cast<short>(float): round half to even
  const int i = (int)rintf(v);
  return i < -32768 ? -32768 : (i > 32767 ? 32767 : i);
}

__device__ inline Axis axis_x(int d, double scale, int n) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { s = 0; f = 0.f; }
  if (s >= n - 1) { s = n - 1; f = 0.f; }
  Axis a;
  a.s0 = s; a.s1 = s + 1 < n ? s + 1 : n - 1;
  a.w0 = rint_short((1.f - f) * 2048.f); a.w1 = rint_short(f * 2048.f);
  return a;
}

__device__ inline Axis axis_y(int d, double scale, int n) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  const int s = (int)floorf(f);
  f -= (float)s;
  Axis a;
  a.s0 = s < 0 ? 0 : (s < n ? s : n - 1);
  a.s1 = s + 1 < 0 ? 0 : (s + 1 < n ? s + 1 : n - 1);
  a.w0 = rint_short((1.f - f) * 2048.f); a.w1 = rint_short(f * 2048.f);
  return a;
}

__device__ inline bool is_int_scale(double scale, int k) { return fabs(scale - (double)k) < 2.220446049250313e-16 && (int)rint(scale) == k; }

// one channel of pixel (yy, xx) of `src` (h0 x w0, HWC uint8, 3 channels, row stride in bytes) resized to (nh x nw) the way
// cv2.resize(..., INTER_LINEAR) computes it (identity when the sizes agree, 2x2 area mean for an exact 2x down-scale)
struct ResizeGeom { double sx, sy; bool resize, area2; };
__device__ inline ResizeGeom resize_geom(int h0, int w0, int nh, int nw) {
  ResizeGeom g;
  g.resize = nw != w0 || nh != h0;
  g.sx = 1.0 / ((double)nw / (double)w0);
  g.sy = 1.0 / ((double)nh / (double)h0);
  g.area2 = g.resize && is_int_scale(g.sx, 2) && is_int_scale(g.sy, 2);
  return g;
}
__device__ inline void resized_pixel(const unsigned char* src, int h0, int w0, int stride, const ResizeGeom& g, int yy, int xx, int out[3]) {
  if (!g.resize) {
    const unsigned char* s = src + (size_t)yy * stride + xx * 3;
    out[0] = s[0]; out[1] = s[1]; out[2] = s[2];
  } else if (g.area2) {
    const unsigned char* s = src + (size_t)(2 * yy) * stride + (2 * xx) * 3;
    const unsigned char* t = s + stride;
    out[0] = (s[0] + s[3] + t[0] + t[3] + 2) >> 2;
    out[1] = (s[1] + s[4] + t[1] + t[4] + 2) >> 2;
    out[2] = (s[2] + s[5] + t[2] + t[5] + 2) >> 2;
  } else {
    const Axis ax = axis_x(xx, g.sx, w0);
    const Axis ay = axis_y(yy, g.sy, h0);
    const unsigned char* r0 = src + (size_t)ay.s0 * stride;
    const unsigned char* r1 = src + (size_t)ay.s1 * stride;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0v = r0[ax.s0 * 3 + c] * ax.w0 + r0[ax.s1 * 3 + c] * ax.w1;
      const int h1v = r1[ax.s0 * 3 + c] * ax.w0 + r1[ax.s1 * 3 + c] * ax.w1;
      out[c] = (((ay.w0 * (h0v >> 4)) >> 16) + ((ay.w1 * (h1v >> 4)) >> 16) + 2) >> 2;
    }
  }
}
}  // namespace
