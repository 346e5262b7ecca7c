/*
 * bso_scene.c -- ORACLE (test infrastructure only; see bslam_oracle.h).
 *
 * Restates the producers on either side of the hot path so that the reference's
 * known-answer scenes (the .cc files under BS/test/) can be rebuilt on the CPU: keyframe
 * preprocessing (brightness, normals, radii) and surfel creation.
 * BS/ = /root/reference/applications/badslam/src/badslam/
 */
#include <math.h>
#include <stdlib.h>

#include "bslam_oracle.h"
#include "bso_math.h"

/* ComputeBrightnessKernel BS/cuda_image_processing.cu:165-176 */
void bso_compute_brightness(int width, int height, const uint8_t* rgb, const bslam_buffer2d* out_color) {
  for (int y = 0; y < height; ++y) {
    for (int x = 0; x < width; ++x) {
      const uint8_t* c = rgb + 3 * ((size_t)y * width + x);
      uint8_t intensity = (uint8_t)((0.299f * c[0] + 0.587f * c[1] + 0.114f * c[2]) + 0.5f);
      uint8_t* o = (uint8_t*)out_color->address + (size_t)y * out_color->pitch + 4 * (size_t)x;
      o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = intensity;
    }
  }
}

/* IEEE binary16 conversions (CUDA __float2half_rn / __half2float, used at
 * BS/cuda_depth_processing.cu:355 and BS/kernel_create_surfels.cu:118). */
static uint16_t float_to_half_rn(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t mant = x & 0x007fffffu;
  int32_t exp = (int32_t)((x >> 23) & 0xff);
  if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x0200u : 0));
  int32_t e = exp - 127 + 15;
  if (e >= 0x1f) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    mant |= 0x00800000u;
    uint32_t shift = (uint32_t)(14 - e);
    uint32_t half_mant = mant >> shift;
    uint32_t rem = mant & ((1u << shift) - 1);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_mant & 1))) half_mant++;
    return (uint16_t)(sign | half_mant);
  }
  uint32_t half = (uint32_t)(e << 10) | (mant >> 13);
  uint32_t rem = mant & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) half++;
  return (uint16_t)(sign | half);
}

float bso_half_to_float(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f;
  uint32_t mant = h & 0x3ffu;
  uint32_t x;
  if (exp == 0) {
    if (mant == 0) {
      x = sign;
    } else {
      int e = -1;
      do { ++e; mant <<= 1; } while (!(mant & 0x400u));
      mant &= 0x3ffu;
      x = sign | ((uint32_t)(127 - 15 - e) << 23) | (mant << 13);
    }
  } else if (exp == 0x1f) {
    x = sign | 0x7f800000u | (mant << 13);
  } else {
    x = sign | ((exp + 127 - 15) << 23) | (mant << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

/* BilateralFilteringAndDepthCutoffCUDAKernel BS/cuda_depth_processing.cu:42-100 (+ host wrapper :102-132) */
void bso_bilateral_filter_and_depth_cutoff(float sigma_xy, float sigma_value, float radius_factor, uint16_t max_depth,
                                           float raw_to_float_depth, const bslam_buffer2d* in_depth, const bslam_buffer2d* out_depth) {
  const int w = out_depth->width, h = out_depth->height;
  const float denom_xy = 2.0f * sigma_xy * sigma_xy, denom_value = 2.0f * sigma_value * sigma_value;
  const int radius = (int)(radius_factor * sigma_xy + 0.5f);
  const int radius_squared = radius * radius;
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      const uint16_t center_value = BSO_AT(uint16_t, in_depth, y, x);
      if (center_value == 0 || center_value > max_depth) { BSO_AT(uint16_t, out_depth, y, x) = BSLAM_UNKNOWN_DEPTH; continue; }
      const float inv_center_value = 1.0f / (raw_to_float_depth * center_value);
      float sum = 0, weight = 0;
      const int min_y = y - radius > 0 ? y - radius : 0, max_y = y + radius < h - 1 ? y + radius : h - 1;
      const int min_x = x - radius > 0 ? x - radius : 0, max_x = x + radius < w - 1 ? x + radius : w - 1;
      for (int sy = min_y; sy <= max_y; ++sy) {
        const int dy = sy - y;
        for (int sx = min_x; sx <= max_x; ++sx) {
          const int dx = sx - x;
          const int grid_distance_squared = dx * dx + dy * dy;
          if (grid_distance_squared > radius_squared) continue;
          const uint16_t sample = BSO_AT(uint16_t, in_depth, sy, sx);
          if (sample == 0) continue;
          const float inv_sample = 1.0f / (raw_to_float_depth * sample);
          float value_distance_squared = inv_center_value - inv_sample;
          value_distance_squared *= value_distance_squared;
          const float wgt = bso_expf(-grid_distance_squared / denom_xy + -value_distance_squared / denom_value);
          sum += wgt * inv_sample;
          weight += wgt;
        }
      }
      if (weight == 0) { BSO_AT(uint16_t, out_depth, y, x) = BSLAM_UNKNOWN_DEPTH; continue; }
      const float v = 1.0f / (raw_to_float_depth * sum / weight);
      BSO_AT(uint16_t, out_depth, y, x) = (uint16_t)(v >= 65535.f ? 65535 : (v <= 0.f || v != v ? 0 : (int)v));   /* cvt.rzi.u16.f32 saturates */
    }
  }
}

/* ComputeNormalsCUDAKernel BS/cuda_depth_processing.cu:134-255 */
void bso_compute_normals(const bslam_camera4f* depth_camera, const bslam_depth_params* dp, const bslam_buffer2d* in_depth,
                         const bslam_buffer2d* out_depth, const bslam_buffer2d* out_normals) {
  const int w = in_depth->width, h = in_depth->height;
  bso_unprojector u = bso_make_unprojector(depth_camera);
  const int cell = dp->sparse_surfel_cell_size;
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      uint16_t* od = &BSO_AT(uint16_t, out_depth, y, x);
      uint16_t* on = &BSO_AT(uint16_t, out_normals, y, x);
      const int kBorder = 1;
      if (x < kBorder || y < kBorder || x >= w - kBorder || y >= h - kBorder) {
        *od = BSLAM_UNKNOWN_DEPTH; *on = bso_image_space_normal_to_u16(0, 0); continue;
      }
      uint16_t c = BSO_AT(uint16_t, in_depth, y, x);
      if (c & BSLAM_INVALID_DEPTH_BIT) { *od = BSLAM_UNKNOWN_DEPTH; *on = bso_image_space_normal_to_u16(0, 0); continue; }
      uint16_t rr = BSO_AT(uint16_t, in_depth, y, x + 1);
      uint16_t ll = BSO_AT(uint16_t, in_depth, y, x - 1);
      uint16_t bb = BSO_AT(uint16_t, in_depth, y + 1, x);
      uint16_t tt = BSO_AT(uint16_t, in_depth, y - 1, x);
      if ((rr & BSLAM_INVALID_DEPTH_BIT) || (ll & BSLAM_INVALID_DEPTH_BIT) || (bb & BSLAM_INVALID_DEPTH_BIT) || (tt & BSLAM_INVALID_DEPTH_BIT)) {
        *od = BSLAM_UNKNOWN_DEPTH; *on = bso_image_space_normal_to_u16(0, 0); continue;
      }
#define CF(yy, xx) BSO_AT(float, &dp->cfactor_buffer, (yy) / cell, (xx) / cell)
      float center_depth = bso_raw_to_calibrated_depth(dp->a, CF(y, x), dp->raw_to_float_depth, c);
      float left_depth = bso_raw_to_calibrated_depth(dp->a, CF(y, x - 1), dp->raw_to_float_depth, ll);
      float top_depth = bso_raw_to_calibrated_depth(dp->a, CF(y - 1, x), dp->raw_to_float_depth, tt);
      float right_depth = bso_raw_to_calibrated_depth(dp->a, CF(y, x + 1), dp->raw_to_float_depth, rr);
      float bottom_depth = bso_raw_to_calibrated_depth(dp->a, CF(y + 1, x), dp->raw_to_float_depth, bb);
#undef CF
      bso_f3 left_point = bso_unproject(&u, x - 1, y, left_depth);
      bso_f3 top_point = bso_unproject(&u, x, y - 1, top_depth);
      bso_f3 right_point = bso_unproject(&u, x + 1, y, right_depth);
      bso_f3 bottom_point = bso_unproject(&u, x, y + 1, bottom_depth);
      bso_f3 center_point = bso_unproject(&u, x, y, center_depth);
      const float kRatioThresholdSquared = 2.f * 2.f;
      float left_dist_squared = bso_sqlen(bso_sub(left_point, center_point));
      float right_dist_squared = bso_sqlen(bso_sub(right_point, center_point));
      float left_right_ratio = left_dist_squared / right_dist_squared;
      bso_f3 left_to_right;
      if (left_right_ratio < kRatioThresholdSquared && left_right_ratio > 1.f / kRatioThresholdSquared) left_to_right = bso_sub(right_point, left_point);
      else if (left_dist_squared < right_dist_squared) left_to_right = bso_sub(center_point, left_point);
      else left_to_right = bso_sub(right_point, center_point);
      float bottom_dist_squared = bso_sqlen(bso_sub(bottom_point, center_point));
      float top_dist_squared = bso_sqlen(bso_sub(top_point, center_point));
      float bottom_top_ratio = bottom_dist_squared / top_dist_squared;
      bso_f3 bottom_to_top;
      if (bottom_top_ratio < kRatioThresholdSquared && bottom_top_ratio > 1.f / kRatioThresholdSquared) bottom_to_top = bso_sub(top_point, bottom_point);
      else if (bottom_dist_squared < top_dist_squared) bottom_to_top = bso_sub(center_point, bottom_point);
      else bottom_to_top = bso_sub(top_point, center_point);
      bso_f3 normal = bso_cross(left_to_right, bottom_to_top);
      float length = bso_norm(normal);
      if (!(length > 1e-6f)) {
        normal = bso_make3(0, 0, -1);
      } else {
        float inv_length = ((u.fy_inv < 0) ? -1.0f : 1.0f) / length;
        normal.x *= inv_length;
        normal.y *= inv_length;
      }
      *on = bso_image_space_normal_to_u16(normal.x, normal.y);
      *od = c;
    }
  }
}

/* ComputePointRadiiAndRemoveIsolatedPixelsCUDAKernel<4> BS/cuda_depth_processing.cu:286-357.  in_depth is the
 * output of bso_compute_normals (its 1-pixel border is invalid, so the 4-neighbourhood never leaves the image). */
void bso_compute_point_radii_and_remove_isolated_pixels(const bslam_camera4f* depth_camera, float raw_to_float_depth,
                                                        const bslam_buffer2d* in_depth, const bslam_buffer2d* out_radius,
                                                        const bslam_buffer2d* out_depth) {
  const int w = in_depth->width, h = in_depth->height;
  bso_unprojector u = bso_make_unprojector(depth_camera);
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      const uint16_t d16 = BSO_AT(uint16_t, in_depth, y, x);
      if (d16 & BSLAM_INVALID_DEPTH_BIT) {
        BSO_AT(uint16_t, out_depth, y, x) = BSLAM_UNKNOWN_DEPTH;
        BSO_AT(uint16_t, out_radius, y, x) = 0;   /* the reference leaves it unwritten */
        continue;
      }
      float depth = raw_to_float_depth * d16;
      bso_f3 local = bso_make3(depth * (u.fx_inv * x + u.cx_inv), depth * (u.fy_inv * y + u.cy_inv), depth);
      int neighbor_count = 0;
      float min_sq = INFINITY;
      for (int dy = y - 1; dy < y + 2; ++dy) {
        for (int dx = x - 1; dx < x + 2; ++dx) {
          if ((dx != x && dy != y) || (dx == x && dy == y)) continue;
          if (dx < 0 || dy < 0 || dx >= w || dy >= h) continue;     /* only reachable if the input has a valid border */
          uint16_t dd = BSO_AT(uint16_t, in_depth, dy, dx);
          if (dd & BSLAM_INVALID_DEPTH_BIT) continue;
          ++neighbor_count;
          float ddepth = raw_to_float_depth * dd;
          bso_f3 other = bso_make3(ddepth * (u.fx_inv * dx + u.cx_inv), ddepth * (u.fy_inv * dy + u.cy_inv), ddepth);
          float dsq = bso_sqlen(bso_sub(other, local));
          if (dsq < min_sq) min_sq = dsq;
        }
      }
      int valid = neighbor_count >= 4;
      BSO_AT(uint16_t, out_radius, y, x) = float_to_half_rn(valid ? min_sq : 0);
      BSO_AT(uint16_t, out_depth, y, x) = valid ? d16 : BSLAM_UNKNOWN_DEPTH;
    }
  }
}

/* ComputeMinMaxDepthCUDA BS/cuda_depth_processing.cu:391-465 (init values inf / 0, BS/cuda_depth_processing.cu:374-388) */
void bso_compute_min_max_depth(const bslam_buffer2d* depth, float raw_to_float_depth, float* min_depth_out, float* max_depth_out) {
  float min_depth = INFINITY, max_depth = 0.f;
  for (int y = 0; y < depth->height; ++y) {
    for (int x = 0; x < depth->width; ++x) {
      const uint16_t d16 = BSO_AT(uint16_t, depth, y, x);
      if (d16 & BSLAM_INVALID_DEPTH_BIT) continue;
      const float dm = raw_to_float_depth * d16;
      if (dm < min_depth) min_depth = dm;
      if (dm > max_depth) max_depth = dm;
    }
  }
  if (min_depth_out) *min_depth_out = min_depth;
  if (max_depth_out) *max_depth_out = max_depth;
}

/* The depth half of the Keyframe constructor from raw images (BS/keyframe.cc:117-147): normals, radii + isolated
 * pixel removal, min / max of the normals stage's depth. */
void bso_preprocess_depth(const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
                          const bslam_buffer2d* in_depth, const bslam_buffer2d* out_depth,
                          const bslam_buffer2d* out_normals, const bslam_buffer2d* out_radius,
                          float* min_depth_out, float* max_depth_out) {
  const int w = in_depth->width, h = in_depth->height;
  uint16_t* tmp = (uint16_t*)malloc((size_t)w * h * sizeof(uint16_t));
  bslam_buffer2d tmp_buf = {tmp, h, w, (size_t)w * sizeof(uint16_t)};
  bso_compute_normals(depth_camera, dp, in_depth, &tmp_buf, out_normals);
  bso_compute_point_radii_and_remove_isolated_pixels(depth_camera, dp->raw_to_float_depth, &tmp_buf, out_radius, out_depth);
  bso_compute_min_max_depth(&tmp_buf, dp->raw_to_float_depth, min_depth_out, max_depth_out);
  free(tmp);
}
