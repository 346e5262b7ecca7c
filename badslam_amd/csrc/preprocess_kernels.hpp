// preprocess_kernels.hpp -- keyframe preprocessing producers for gfx950: brightness,
// bilateral depth filter + cutoff, pixel normals, point radii + isolated-pixel removal, min / max depth.
// Replaces BS/cuda_image_processing.cu:165-194 and BS/cuda_depth_processing.cu:42-465.  These define the
// u16 / half / uchar4 image formats the bundle-adjustment kernels read.  All outputs are
// integers (u8 / u16 / half bits): they are compared bit for bit with the oracle.
#pragma once

#include <hip/hip_fp16.h>

#include "device_math.hpp"

namespace bslam {

struct Img {   // pitched 2-D image
  uint8_t* base; uint32_t pitch; int width, height;
  template <class T> __device__ __forceinline__ T& at(int y, int x) const { return *((T*)(base + (size_t)y * pitch) + x); }
};

// ComputeBrightnessKernel BS/cuda_image_processing.cu:165-176: rgb (3 B / pixel) -> rgb + luma
__global__ __launch_bounds__(256) void brightness_kernel(Img rgb, Img color) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= color.width || y >= color.height) return;
  const uint8_t* p = rgb.base + (size_t)y * rgb.pitch + 3 * (size_t)x;
  const uint32_t r = p[0], g = p[1], b = p[2];
  const uint32_t intensity = (uint32_t)f2i((0.299f * (float)r + 0.587f * (float)g + 0.114f * (float)b) + 0.5f) & 0xffu;
  color.at<uint32_t>(y, x) = r | (g << 8) | (b << 16) | (intensity << 24);
}

// BilateralFilteringAndDepthCutoffCUDAKernel BS/cuda_depth_processing.cu:42-100
__global__ __launch_bounds__(256) void bilateral_kernel(float denom_xy, float denom_value, int radius, int radius_squared, uint32_t max_depth,
                                                       float raw_to_float_depth, Img in, Img out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= out.width || y >= out.height) return;
  const uint32_t center_value = in.at<uint16_t>(y, x);
  if (center_value == 0 || center_value > max_depth) { out.at<uint16_t>(y, x) = BSLAM_UNKNOWN_DEPTH; return; }
  const float inv_center_value = 1.0f / (raw_to_float_depth * (float)center_value);
  float sum = 0, weight = 0;
  const int min_y = max(0, y - radius), max_y = min(out.height - 1, y + radius);
  const int min_x = max(0, x - radius), max_x = min(out.width - 1, x + radius);
  for (int sy = min_y; sy <= max_y; ++sy) {
    const int dy = sy - y;
    for (int sx = min_x; sx <= max_x; ++sx) {
      const int dx = sx - x;
      const int grid_distance_squared = dx * dx + dy * dy;
      if (grid_distance_squared > radius_squared) continue;
      const uint32_t sample = in.at<uint16_t>(sy, sx);
      if (sample == 0) continue;
      const float inv_sample = 1.0f / (raw_to_float_depth * (float)sample);
      float value_distance_squared = inv_center_value - inv_sample;
      value_distance_squared *= value_distance_squared;
      const float w = det_expf((float)(-grid_distance_squared) / denom_xy + -value_distance_squared / denom_value);
      sum += w * inv_sample;
      weight += w;
    }
  }
  if (weight == 0) { out.at<uint16_t>(y, x) = BSLAM_UNKNOWN_DEPTH; return; }
  const float v = 1.0f / (raw_to_float_depth * sum / weight);
  out.at<uint16_t>(y, x) = (uint16_t)min(65535, max(0, f2i(v)));   // cvt.rzi.u16.f32 saturates
}

__device__ __forceinline__ uint32_t small_float_to_s8(float value) {   // BS/util.cuh:105-107
  return (uint32_t)f2i(value * 127 + ((value > 0) ? 0.5f : -0.5f)) & 0xffu;
}
__device__ __forceinline__ uint32_t image_space_normal_to_u16(float x, float y) { return small_float_to_s8(x) | (small_float_to_s8(y) << 8); }

// ComputeNormalsCUDAKernel BS/cuda_depth_processing.cu:134-255
__global__ __launch_bounds__(256) void normals_kernel(CamConsts c, Img in, Img out_depth, Img out_normals) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= in.width || y >= in.height) return;
  const uint32_t zero_normal = image_space_normal_to_u16(0, 0);
  constexpr int kBorder = 1;
  bool ok = !(x < kBorder || y < kBorder || x >= in.width - kBorder || y >= in.height - kBorder);
  uint32_t cc = 0, rr = 0, ll = 0, bb = 0, tt = 0;
  if (ok) {
    cc = in.at<uint16_t>(y, x);
    ok = !(cc & BSLAM_INVALID_DEPTH_BIT);
  }
  if (ok) {
    rr = in.at<uint16_t>(y, x + 1); ll = in.at<uint16_t>(y, x - 1); bb = in.at<uint16_t>(y + 1, x); tt = in.at<uint16_t>(y - 1, x);
    ok = !((rr | ll | bb | tt) & BSLAM_INVALID_DEPTH_BIT);
  }
  if (!ok) { out_depth.at<uint16_t>(y, x) = BSLAM_UNKNOWN_DEPTH; out_normals.at<uint16_t>(y, x) = (uint16_t)zero_normal; return; }
  auto cf = [&](int yy, int xx) { return *(const float*)((const uint8_t*)c.cfactor + (size_t)(yy / c.cell) * c.cfactor_pitch + 4 * (size_t)(xx / c.cell)); };
  const float center_depth = raw_to_calibrated_depth(c.a, cf(y, x), c.raw_to_float_depth, cc);
  const float left_depth = raw_to_calibrated_depth(c.a, cf(y, x - 1), c.raw_to_float_depth, ll);
  const float top_depth = raw_to_calibrated_depth(c.a, cf(y - 1, x), c.raw_to_float_depth, tt);
  const float right_depth = raw_to_calibrated_depth(c.a, cf(y, x + 1), c.raw_to_float_depth, rr);
  const float bottom_depth = raw_to_calibrated_depth(c.a, cf(y + 1, x), c.raw_to_float_depth, bb);
  const f3 left_point = unproject(c, x - 1, y, left_depth);
  const f3 top_point = unproject(c, x, y - 1, top_depth);
  const f3 right_point = unproject(c, x + 1, y, right_depth);
  const f3 bottom_point = unproject(c, x, y + 1, bottom_depth);
  const f3 center_point = unproject(c, x, y, center_depth);
  constexpr float kRatioThresholdSquared = 2.f * 2.f;
  const float left_dist_squared = sqlen(sub3(left_point, center_point));
  const float right_dist_squared = sqlen(sub3(right_point, center_point));
  const float left_right_ratio = left_dist_squared / right_dist_squared;
  f3 left_to_right;
  if (left_right_ratio < kRatioThresholdSquared && left_right_ratio > 1.f / kRatioThresholdSquared) left_to_right = sub3(right_point, left_point);
  else if (left_dist_squared < right_dist_squared) left_to_right = sub3(center_point, left_point);
  else left_to_right = sub3(right_point, center_point);
  const float bottom_dist_squared = sqlen(sub3(bottom_point, center_point));
  const float top_dist_squared = sqlen(sub3(top_point, center_point));
  const float bottom_top_ratio = bottom_dist_squared / top_dist_squared;
  f3 bottom_to_top;
  if (bottom_top_ratio < kRatioThresholdSquared && bottom_top_ratio > 1.f / kRatioThresholdSquared) bottom_to_top = sub3(top_point, bottom_point);
  else if (bottom_dist_squared < top_dist_squared) bottom_to_top = sub3(center_point, bottom_point);
  else bottom_to_top = sub3(top_point, center_point);
  f3 normal = cross(left_to_right, bottom_to_top);
  const float length = norm3(normal);
  if (!(length > 1e-6f)) {
    normal = mk3(0, 0, -1);
  } else {
    const float inv_length = ((c.fy_inv < 0) ? -1.0f : 1.0f) / length;
    normal.x *= inv_length;
    normal.y *= inv_length;
  }
  out_normals.at<uint16_t>(y, x) = (uint16_t)image_space_normal_to_u16(normal.x, normal.y);
  out_depth.at<uint16_t>(y, x) = (uint16_t)cc;
}

// ComputePointRadiiAndRemoveIsolatedPixelsCUDAKernel<4> BS/cuda_depth_processing.cu:286-357
__global__ __launch_bounds__(256) void radii_kernel(CamConsts c, float raw_to_float_depth, Img in, Img out_radius, Img out_depth) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= in.width || y >= in.height) return;
  const uint32_t d16 = in.at<uint16_t>(y, x);
  if (d16 & BSLAM_INVALID_DEPTH_BIT) {
    out_depth.at<uint16_t>(y, x) = BSLAM_UNKNOWN_DEPTH;
    out_radius.at<uint16_t>(y, x) = 0;   // the reference leaves it unwritten
    return;
  }
  const float depth = raw_to_float_depth * (float)d16;
  const f3 local = mk3(depth * (c.fx_inv * x + c.cx_inv), depth * (c.fy_inv * y + c.cy_inv), depth);
  int neighbor_count = 0;
  float min_sq = __uint_as_float(0x7f800000u);
  for (int dy = y - 1; dy < y + 2; ++dy) {
    for (int dx = x - 1; dx < x + 2; ++dx) {
      if ((dx != x && dy != y) || (dx == x && dy == y)) continue;
      if (dx < 0 || dy < 0 || dx >= in.width || dy >= in.height) continue;
      const uint32_t dd = in.at<uint16_t>(dy, dx);
      if (dd & BSLAM_INVALID_DEPTH_BIT) continue;
      ++neighbor_count;
      const float ddepth = raw_to_float_depth * (float)dd;
      const f3 other = mk3(ddepth * (c.fx_inv * dx + c.cx_inv), ddepth * (c.fy_inv * dy + c.cy_inv), ddepth);
      const float dsq = sqlen(sub3(other, local));
      if (dsq < min_sq) min_sq = dsq;
    }
  }
  const bool valid = neighbor_count >= 4;
  out_radius.at<uint16_t>(y, x) = __half_as_ushort(__float2half_rn(valid ? min_sq : 0.f));
  out_depth.at<uint16_t>(y, x) = valid ? (uint16_t)d16 : (uint16_t)BSLAM_UNKNOWN_DEPTH;
}

// ComputeMinMaxDepthCUDAKernel BS/cuda_depth_processing.cu:391-428: positive floats order like their bit patterns
__global__ __launch_bounds__(256) void min_max_depth_kernel(float raw_to_float_depth, Img depth, uint32_t* __restrict__ result /* [2] */) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  float mn = __uint_as_float(0x7f800000u), mx = 0.f;
  if (x < depth.width && y < depth.height) {
    const uint32_t d16 = depth.at<uint16_t>(y, x);
    if (!(d16 & BSLAM_INVALID_DEPTH_BIT)) mn = mx = raw_to_float_depth * (float)d16;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, off, 64));
    mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&result[0], __float_as_uint(mn));
    atomicMax(&result[1], __float_as_uint(mx));
  }
}

}  // namespace bslam
