/*
 * bso_odometry.c -- ORACLE (test infrastructure only; see bslam_oracle.h).
 *
 * Pairwise frame tracking (odometry) on the CPU: pyramid construction
 * (BS/kernel_downsample.cu), the image-pair Gauss-Newton coefficients and cost (BS/kernel_opt_pose.cu:422-661,
 * 939-1171, GradientXY variant -- the one BadSlam::RunOdometry uses, BS/bad_slam.cc:831) and the coarse-to-fine
 * loop of TrackFramePairwise (BS/pairwise_frame_tracking.cc:256-678) with use_pyramid_level_0 = true.
 * BS/ = /root/reference/applications/badslam/src/badslam/
 *
 * Naming follows the reference: the "surfel" images are the BASE frame's (the frame whose pixels are projected),
 * the "frame" images are the TRACKED frame's; estimate_frame_T_surfel_frame = (base_T_frame)^-1.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bslam_oracle.h"
#include "bso_math.h"

/* single-channel u8 texture, clamp addressing, linear filter, normalised float (same model as bso_tex_w) */
static float texel_u8(const bslam_buffer2d* img, int ix, int iy) {
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (ix > img->width - 1) ix = img->width - 1;
  if (iy > img->height - 1) iy = img->height - 1;
  return BSO_AT(uint8_t, img, iy, ix) * (1.0f / 255.0f);
}
static float tex_u8(const bslam_buffer2d* img, float x, float y, int mode) {
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fx = floorf(xb), fy = floorf(yb);
  float a = xb - fx, b = yb - fy;
  if (mode == BSLAM_TEX_FIXED_POINT_1_8) {
    a = floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f);
    b = floorf(b * 256.0f + 0.5f) * (1.0f / 256.0f);
  }
  const int i = (int)fminf(fmaxf(fx, -2.0f), (float)img->width);
  const int j = (int)fminf(fmaxf(fy, -2.0f), (float)img->height);
  const float t00 = texel_u8(img, i, j), t10 = texel_u8(img, i + 1, j), t01 = texel_u8(img, i, j + 1), t11 = texel_u8(img, i + 1, j + 1);
  return (((1.0f - a) * (1.0f - b) * t00 + a * (1.0f - b) * t10) + (1.0f - a) * b * t01) + a * b * t11;
}
/* twin shape of the residual's sampling (bso_math.h): byte units, nested fma interpolation */
static float texel_u8_b(const bslam_buffer2d* img, int ix, int iy) {
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (ix > img->width - 1) ix = img->width - 1;
  if (iy > img->height - 1) iy = img->height - 1;
  return (float)BSO_AT(uint8_t, img, iy, ix);
}
static float tex_u8_b(const bslam_buffer2d* img, float x, float y, int mode) {
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fx = floorf(xb), fy = floorf(yb);
  const float a = bso_quantise_weight(xb - fx, mode), b = bso_quantise_weight(yb - fy, mode);
  const int i = (int)fminf(fmaxf(fx, -2.0f), (float)img->width);
  const int j = (int)fminf(fmaxf(fy, -2.0f), (float)img->height);
  return bso_bilinear_bytes(texel_u8_b(img, i, j), texel_u8_b(img, i + 1, j), texel_u8_b(img, i, j + 1), texel_u8_b(img, i + 1, j + 1), a, b);
}
static uint8_t to_u8(float v) { return (uint8_t)(v >= 255.f ? 255 : (v <= 0.f || v != v ? 0 : (int)v)); }   /* cvt.rzi.u8.f32 saturates */

/* ComputeBrightnessKernel(texture) BS/cuda_image_processing.cu:196-205: luma of the uchar4 keyframe colour image */
void bso_brightness_from_color(const bslam_buffer2d* color_uchar4, const bslam_buffer2d* out_u8) {
  for (int y = 0; y < out_u8->height; ++y)
    for (int x = 0; x < out_u8->width; ++x) {
      const uint8_t luma = ((const uint8_t*)color_uchar4->address + (size_t)y * color_uchar4->pitch + 4 * (size_t)x)[3];
      BSO_AT(uint8_t, out_u8, y, x) = to_u8(255.f * (luma * (1.0f / 255.0f)));   /* tex at a texel centre = the texel */
    }
}

/* CalibrateDepthAndTransformColorToDepthCUDAKernel BS/kernel_downsample.cu:236-266 */
void bso_calibrate_depth_and_transform_color_to_depth(const bslam_camera4f* depth_camera, const bslam_camera4f* color_camera,
                                                      const bslam_depth_params* dp, const bslam_buffer2d* depth_u16, const bslam_buffer2d* color_u8,
                                                      int tex_mode, const bslam_buffer2d* out_depth, const bslam_buffer2d* out_color) {
  bso_depth_to_color d2c = bso_make_depth_to_color(depth_camera, color_camera);
  const int cell = dp->sparse_surfel_cell_size;
  for (int y = 0; y < out_depth->height; ++y)
    for (int x = 0; x < out_depth->width; ++x) {
      const uint16_t raw = BSO_AT(uint16_t, depth_u16, y, x);
      float depth = 0;
      if (!(raw & BSLAM_INVALID_DEPTH_BIT))
        depth = bso_raw_to_calibrated_depth(dp->a, BSO_AT(float, &dp->cfactor_buffer, y / cell, x / cell), dp->raw_to_float_depth, raw);
      bso_f2 pc = {x + 0.5f, y + 0.5f}, color_pxy;
      const int in_bounds = bso_depth_to_color_pxy(pc, &d2c, &color_pxy);
      BSO_AT(float, out_depth, y, x) = in_bounds ? depth : 0;
      BSO_AT(uint8_t, out_color, y, x) = to_u8(255.f * tex_u8(color_u8, color_pxy.x, color_pxy.y, tex_mode) + 0.5f);
    }
}

/* CalibrateDepthCUDAKernel BS/kernel_downsample.cu:292-312 */
void bso_calibrate_depth(const bslam_depth_params* dp, const bslam_buffer2d* depth_u16, const bslam_buffer2d* out_depth) {
  const int cell = dp->sparse_surfel_cell_size;
  for (int y = 0; y < out_depth->height; ++y)
    for (int x = 0; x < out_depth->width; ++x) {
      const uint16_t raw = BSO_AT(uint16_t, depth_u16, y, x);
      BSO_AT(float, out_depth, y, x) = (raw & BSLAM_INVALID_DEPTH_BIT) ? 0.f
          : bso_raw_to_calibrated_depth(dp->a, BSO_AT(float, &dp->cfactor_buffer, y / cell, x / cell), dp->raw_to_float_depth, raw);
    }
}

/* CUDABuffer_<u8>::SetToReadModeNormalized LV/cuda/cuda_buffer.cu:82-102 */
void bso_set_to_read_mode_normalized(const bslam_buffer2d* in_u8, const bslam_buffer2d* out_u8) {
  for (int y = 0; y < out_u8->height; ++y)
    for (int x = 0; x < out_u8->width; ++x) BSO_AT(uint8_t, out_u8, y, x) = to_u8(255.f * (BSO_AT(uint8_t, in_u8, y, x) * (1.0f / 255.0f)));
}

/* DownsampleImagesCUDAKernel BS/kernel_downsample.cu:105-152 */
void bso_downsample_images(const bslam_buffer2d* depth, const bslam_buffer2d* normals, const bslam_buffer2d* color_u8, int tex_mode,
                           const bslam_buffer2d* out_depth, const bslam_buffer2d* out_normals, const bslam_buffer2d* out_color) {
  static const int kOffsets[4][2] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}};
  for (int y = 0; y < out_depth->height; ++y)
    for (int x = 0; x < out_depth->width; ++x) {
      float depths[4], depth_sum = 0;
      int depth_count = 0;
      for (int i = 0; i < 4; ++i) {
        depths[i] = BSO_AT(float, depth, 2 * y + kOffsets[i][0], 2 * x + kOffsets[i][1]);
        if (depths[i] > 0) { depth_sum += depths[i]; depth_count += 1; }
        else depths[i] = INFINITY;
      }
      if (depth_count == 0) {
        BSO_AT(float, out_depth, y, x) = 0;
      } else {
        const float average_depth = depth_sum / depth_count;
        int closest_index = 0;
        float closest_distance = INFINITY;
        for (int i = 0; i < 4; ++i) {
          const float distance = fabsf(depths[i] - average_depth);
          if (distance < closest_distance) { closest_index = i; closest_distance = distance; }
        }
        BSO_AT(float, out_depth, y, x) = depths[closest_index];
        BSO_AT(uint16_t, out_normals, y, x) = BSO_AT(uint16_t, normals, 2 * y + kOffsets[closest_index][0], 2 * x + kOffsets[closest_index][1]);
      }
      BSO_AT(uint8_t, out_color, y, x) = to_u8(255.f * tex_u8(color_u8, 2 * x + 1.0f, 2 * y + 1.0f, tex_mode) + 0.5f);
    }
}

/* ComputeSobelGradientMagnitudeCUDA(texture, gradmag) BS/cuda_image_processing.cu:104-167: Sobel magnitude of the luma channel
 * (.w of the uchar4 colour image read through the clamp texture at texel centres), normalised to 0..255 and truncated. */
void bso_compute_sobel_gradient_magnitude(const bslam_buffer2d* color_uchar4, const bslam_buffer2d* out_u8) {
  const int w = out_u8->width, h = out_u8->height;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      float I[3][3];
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          int ix = x + dx, iy = y + dy;
          if (ix < 0) ix = 0;
          if (iy < 0) iy = 0;
          if (ix > color_uchar4->width - 1) ix = color_uchar4->width - 1;
          if (iy > color_uchar4->height - 1) iy = color_uchar4->height - 1;
          const uint8_t luma = ((const uint8_t*)color_uchar4->address + (size_t)iy * color_uchar4->pitch + 4 * (size_t)ix)[3];
          I[dy + 1][dx + 1] = 255.f * (luma * (1.0f / 255.0f));   /* 255 * tex2D<float4>(...).w at a texel centre */
        }
      const float gx = 1 * I[0][2] - 1 * I[0][0] + 2 * I[1][2] - 2 * I[1][0] + 1 * I[2][2] - 1 * I[2][0];
      const float gy = 1 * I[2][0] - 1 * I[0][0] + 2 * I[2][1] - 2 * I[0][1] + 1 * I[2][2] - 1 * I[0][2];
      const float kNormalizer = 255.99f / (1.41421356237309504880f * 4 * 255.f);
      BSO_AT(uint8_t, out_u8, y, x) = to_u8(kNormalizer * sqrtf(gx * gx + gy * gy));
    }
}

/* CalibrateAndDownsampleImagesCUDAKernel<downsample_color> BS/kernel_downsample.cu:40-105: the first pyramid step when the tracked
 * frame's level 0 is not used: raw u16 depth -> calibrated float depth of the 2x2 block's member closest to the block mean.
 * (The cfactor cell is looked up with the DOWNSAMPLED pixel coordinates, as the reference does, :65-66.) */
void bso_calibrate_and_downsample_images(int downsample_color, const bslam_depth_params* dp, const bslam_buffer2d* depth_u16, const bslam_buffer2d* normals,
                                         const bslam_buffer2d* color_u8, int tex_mode, const bslam_buffer2d* out_depth, const bslam_buffer2d* out_normals,
                                         const bslam_buffer2d* out_color) {
  static const int kOffsets[4][2] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}};
  const int cell = dp->sparse_surfel_cell_size;
  for (int y = 0; y < out_depth->height; ++y)
    for (int x = 0; x < out_depth->width; ++x) {
      float depths[4], depth_sum = 0;
      int depth_count = 0;
      for (int i = 0; i < 4; ++i) {
        const uint16_t raw = BSO_AT(uint16_t, depth_u16, 2 * y + kOffsets[i][0], 2 * x + kOffsets[i][1]);
        if (!(raw & BSLAM_INVALID_DEPTH_BIT)) {
          depths[i] = bso_raw_to_calibrated_depth(dp->a, BSO_AT(float, &dp->cfactor_buffer, y / cell, x / cell), dp->raw_to_float_depth, raw);
          depth_sum += depths[i];
          depth_count += 1;
        } else {
          depths[i] = INFINITY;
        }
      }
      if (depth_count == 0) {
        BSO_AT(float, out_depth, y, x) = 0;
      } else {
        const float average_depth = depth_sum / depth_count;
        int closest_index = 0;
        float closest_distance = INFINITY;
        for (int i = 0; i < 4; ++i) {
          const float distance = fabsf(depths[i] - average_depth);
          if (distance < closest_distance) { closest_index = i; closest_distance = distance; }
        }
        BSO_AT(float, out_depth, y, x) = depths[closest_index];
        BSO_AT(uint16_t, out_normals, y, x) = BSO_AT(uint16_t, normals, 2 * y + kOffsets[closest_index][0], 2 * x + kOffsets[closest_index][1]);
      }
      const float color = downsample_color ? tex_u8(color_u8, 2 * x + 1.0f, 2 * y + 1.0f, tex_mode) : tex_u8(color_u8, x + 0.5f, y + 0.5f, tex_mode);
      BSO_AT(uint8_t, out_color, y, x) = to_u8(255.f * color + 0.5f);
    }
}

/* Variant switches of the tracker (the reference passes them as arguments, BS/pairwise_frame_tracking.cc:164-166):
 * use_gradmag: ONE colour residual on gradient-magnitude images (BS/kernel_opt_pose.cu:192-222, 713-937, 1173-1338) instead of
 * the two descriptor residuals; use_pyramid_level_0 = 0: the tracked frame enters the pyramid at level 1 through
 * CalibrateAndDownsampleImagesCUDA and level 0 is not tracked. */
static int g_use_gradmag = 0, g_use_pyramid_level_0 = 1;
void bso_set_tracking_variant(int use_gradmag, int use_pyramid_level_0) { g_use_gradmag = use_gradmag != 0; g_use_pyramid_level_0 = use_pyramid_level_0 != 0; }

/* ---- per-pixel evaluation shared by the coefficient and the cost kernel ---- */
typedef struct {
  int visible;
  float raw_depth_residual, depth_jacobian[6];
  float r1, r2, J1[6], J2[6];
} pair_terms;

static void point_gradient_u8(const bslam_buffer2d* img, bso_f2 p, float* dx, float* dy) {   /* one block of BS/cost_function.cuh:265-301 */
  int ix = bso_f2i(fmaxf(0.f, p.x - 0.5f));
  int iy = bso_f2i(fmaxf(0.f, p.y - 0.5f));
  const float tx = fmaxf(0.f, fminf(1.f, p.x - 0.5f - ix));
  const float ty = fmaxf(0.f, fminf(1.f, p.y - 0.5f - iy));
  if (ix > img->width - 1) ix = img->width - 1;
  if (iy > img->height - 1) iy = img->height - 1;
  if (!BSO_LITERAL) {   /* byte units; the caller applies 180 / 255 */
    bso_bilinear_gradient_bytes(texel_u8_b(img, ix, iy), texel_u8_b(img, ix + 1, iy), texel_u8_b(img, ix, iy + 1), texel_u8_b(img, ix + 1, iy + 1), tx, ty, dx, dy);
    return;
  }
  const float tl = texel_u8(img, ix, iy), tr = texel_u8(img, ix + 1, iy), bl = texel_u8(img, ix, iy + 1), br = texel_u8(img, ix + 1, iy + 1);
  *dx = (br - bl) * ty + (tr - tl) * (1 - ty);
  *dy = (br - tr) * tx + (bl - tl) * (1 - tx);
}

static void evaluate_pixel(int x, int y, int use_depth, int use_desc, int need_jacobians,
                           const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, float baseline_fx, float threshold_factor,
                           const bslam_mat3x4* T, const bslam_buffer2d* surfel_depth, const bslam_buffer2d* surfel_normals, const bslam_buffer2d* surfel_color,
                           const bslam_buffer2d* frame_depth, const bslam_buffer2d* frame_normals, const bslam_buffer2d* frame_color, int tex_mode,
                           pair_terms* t) {
  t->visible = 0;
  bso_unprojector u = bso_make_unprojector(depth_camera);
  bso_depth_to_color d2c = bso_make_depth_to_color(depth_camera, color_camera);
  const float surfel_calibrated_depth = BSO_AT(float, surfel_depth, y, x);
  if (!(surfel_calibrated_depth > 0)) return;
  bso_f3 local;
  if (!bso_mul34_if_z_positive(T, bso_unproject(&u, x, y, surfel_calibrated_depth), &local)) return;
  int px, py;
  bso_f2 pxy;
  if (!bso_project_surfel_to_image(frame_depth->width, frame_depth->height, depth_camera, local, &px, &py, &pxy)) return;
  const float pixel_depth = BSO_AT(float, frame_depth, py, px);
  if (!(pixel_depth > 0)) return;
  /* IsAssociatedWithPixel<false>(… image normals …) BS/surfel_projection_nvcc_only.cuh:168-215 */
  const bso_f3 n_local = bso_rotate34(T, bso_u16_to_image_space_normal(BSO_AT(uint16_t, surfel_normals, y, x)));
  const float stddev = bso_depth_stddev(bso_unproj_nx(&u, px), bso_unproj_ny(&u, py), pixel_depth, n_local, baseline_fx);
  if (fabsf(local.z - pixel_depth) > (threshold_factor * BSO_DEPTH_TUKEY) * stddev) return;
  if ((1.0f / bso_norm(local)) * bso_dot(local, n_local) > 0) return;
  if (bso_dot(n_local, bso_u16_to_image_space_normal(BSO_AT(uint16_t, frame_normals, py, px))) < BSO_COS_NORMAL_COMPAT) return;
  int visible = 1;
  if (use_depth) {
    const float inv_stddev = bso_depth_inv_stddev(bso_unproj_nx(&u, px), bso_unproj_ny(&u, py), pixel_depth, n_local, baseline_fx);
    const bso_f3 lu = bso_unproject(&u, px, py, pixel_depth);
    t->raw_depth_residual = bso_depth_residual(inv_stddev, n_local, lu, local);
    bso_jac_depth_pose(inv_stddev, n_local, lu, t->depth_jacobian);
  }
  if (use_desc && g_use_gradmag) {
    /* ComputeRawColorResidualAndJacobian BS/kernel_opt_pose.cu:192-222, BS/cost_function.cuh:324-352 (the reference's expressions as
       written in both evaluation shapes: not on the hot path) */
    bso_f2 cp;
    if (bso_depth_to_color_pxy(pxy, &d2c, &cp)) {
      t->r1 = 255.f * tex_u8(frame_color, cp.x, cp.y, tex_mode) - (float)BSO_AT(uint8_t, surfel_color, y, x);
      t->r2 = 0;
      if (need_jacobians) {
        int ix = bso_f2i(fmaxf(0.f, cp.x - 0.5f)), iy = bso_f2i(fmaxf(0.f, cp.y - 0.5f));
        const float tx = fmaxf(0.f, fminf(1.f, cp.x - 0.5f - ix)), ty = fmaxf(0.f, fminf(1.f, cp.y - 0.5f - iy));
        if (ix > frame_color->width - 1) ix = frame_color->width - 1;
        if (iy > frame_color->height - 1) iy = frame_color->height - 1;
        const float tl = 255.f * texel_u8(frame_color, ix, iy), tr = 255.f * texel_u8(frame_color, ix + 1, iy);
        const float bl = 255.f * texel_u8(frame_color, ix, iy + 1), br = 255.f * texel_u8(frame_color, ix + 1, iy + 1);
        float gx = (br - bl) * ty + (tr - tl) * (1 - ty);
        float gy = (br - tr) * tx + (bl - tl) * (1 - tx);
        gx *= color_camera->fx;
        gy *= color_camera->fy;
        bso_jac_desc_pose(gx, gy, local, t->J1);
        for (int i = 0; i < 6; ++i) t->J2[i] = 0;
      }
    } else {
      visible = 0;
    }
  } else if (use_desc) {
    if (x < surfel_depth->width - 1 && y < surfel_depth->height - 1) {
      const float intensity = 1 / 255.f * BSO_AT(uint8_t, surfel_color, y, x);
      const float t1_intensity = 1 / 255.f * BSO_AT(uint8_t, surfel_color, y, x + 1);
      const float t2_intensity = 1 / 255.f * BSO_AT(uint8_t, surfel_color, y + 1, x);
      const float d1 = (180.f * (t1_intensity - intensity)), d2 = (180.f * (t2_intensity - intensity));
      const bso_f3 sn = bso_u16_to_image_space_normal(BSO_AT(uint16_t, surfel_normals, y, x));
      const float plane_d = (bso_unproj_nx(&u, x) * surfel_calibrated_depth) * sn.x + (bso_unproj_ny(&u, y) * surfel_calibrated_depth) * sn.y +
                            surfel_calibrated_depth * sn.z;
      const float x1_depth = plane_d / (bso_unproj_nx(&u, x + 1) * sn.x + bso_unproj_ny(&u, y) * sn.y + sn.z);
      const bso_f3 x1_local = bso_mul34(T, bso_unproject(&u, x + 1, y, x1_depth));
      const bso_f2 pxy_t1 = bso_project(depth_camera->fx, depth_camera->fy, depth_camera->cx, depth_camera->cy, x1_local);
      if (pxy_t1.x < 0 || pxy_t1.y < 0 || bso_f2i(pxy_t1.x) >= frame_depth->width || bso_f2i(pxy_t1.y) >= frame_depth->height) visible = 0;
      const float y1_depth = plane_d / (bso_unproj_nx(&u, x) * sn.x + bso_unproj_ny(&u, y + 1) * sn.y + sn.z);
      const bso_f3 y1_local = bso_mul34(T, bso_unproject(&u, x, y + 1, y1_depth));
      const bso_f2 pxy_t2 = bso_project(depth_camera->fx, depth_camera->fy, depth_camera->cx, depth_camera->cy, y1_local);
      if (pxy_t2.x < 0 || pxy_t2.y < 0 || bso_f2i(pxy_t2.x) >= frame_depth->width || bso_f2i(pxy_t2.y) >= frame_depth->height) visible = 0;
      bso_f2 c0, c1, c2;
      if (visible && x1_local.z > 0 && y1_local.z > 0 && bso_depth_to_color_pxy(pxy, &d2c, &c0) && bso_depth_to_color_pxy(pxy_t1, &d2c, &c1) &&
          bso_depth_to_color_pxy(pxy_t2, &d2c, &c2)) {
        if (!BSO_LITERAL) {
          const float b0 = tex_u8_b(frame_color, c0.x, c0.y, tex_mode), b1 = tex_u8_b(frame_color, c1.x, c1.y, tex_mode), b2 = tex_u8_b(frame_color, c2.x, c2.y, tex_mode);
          t->r1 = BSO_FMA(BSO_DESC_SCALE, b1 - b0, -d1);
          t->r2 = BSO_FMA(BSO_DESC_SCALE, b2 - b0, -d2);
        } else {
          const float i0 = tex_u8(frame_color, c0.x, c0.y, tex_mode), i1 = tex_u8(frame_color, c1.x, c1.y, tex_mode), i2 = tex_u8(frame_color, c2.x, c2.y, tex_mode);
          t->r1 = (180.f * (i1 - i0)) - d1;
          t->r2 = (180.f * (i2 - i0)) - d2;
        }
        if (need_jacobians) {
          float cdx, cdy, t1dx, t1dy, t2dx, t2dy;
          point_gradient_u8(frame_color, c0, &cdx, &cdy);
          point_gradient_u8(frame_color, c1, &t1dx, &t1dy);
          point_gradient_u8(frame_color, c2, &t2dx, &t2dy);
          const float scale = BSO_LITERAL ? 180.f : BSO_DESC_SCALE;   /* the twin's gradients are in byte units */
          float gx1 = scale * (t1dx - cdx), gy1 = scale * (t1dy - cdy), gx2 = scale * (t2dx - cdx), gy2 = scale * (t2dy - cdy);
          gx1 *= color_camera->fx; gx2 *= color_camera->fx;
          gy1 *= color_camera->fy; gy2 *= color_camera->fy;
          bso_jac_desc_pose(gx1, gy1, local, t->J1);
          bso_jac_desc_pose(gx2, gy2, local, t->J2);
        }
      } else {
        visible = 0;
      }
    } else {
      visible = 0;
    }
  }
  t->visible = visible;
}

static void add_h_b(float raw, float w, const float* J, double* H, double* b) {
  int idx = 0;
  for (int r = 0; r < 6; ++r)
    for (int c = r; c < 6; ++c) H[idx++] += (double)(w * J[r] * J[c]);
  const float wr = w * raw;
  for (int i = 0; i < 6; ++i) b[i] += (double)(wr * J[i]);
}

/* AccumulatePoseEstimationCoeffsFromImagesCUDA (GradientXY) BS/kernel_opt_pose.cu:422-661, BS/kernel_opt_pose.cc:99-192.
 * H, b: float64 sums of the fp32 terms (the reference adds them in an unspecified order with atomics). */
void bso_accumulate_pose_coeffs_from_images(
    int use_depth, int use_desc, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, float baseline_fx, float threshold_factor,
    const bslam_buffer2d* frame_depth, const bslam_buffer2d* frame_normals, const bslam_buffer2d* frame_color,
    const bslam_mat3x4* estimate_frame_T_surfel_frame,
    const bslam_buffer2d* surfel_depth, const bslam_buffer2d* surfel_normals, const bslam_buffer2d* surfel_color, int tex_mode,
    double* H /* [21] */, double* b /* [6] */, uint32_t* visible_count) {
  for (int i = 0; i < 21; ++i) H[i] = 0;
  for (int i = 0; i < 6; ++i) b[i] = 0;
  uint32_t count = 0;
  for (int y = 0; y < surfel_depth->height; ++y)
    for (int x = 0; x < surfel_depth->width; ++x) {
      pair_terms t;
      evaluate_pixel(x, y, use_depth, use_desc, 1, color_camera, depth_camera, baseline_fx, threshold_factor, estimate_frame_T_surfel_frame,
                     surfel_depth, surfel_normals, surfel_color, frame_depth, frame_normals, frame_color, tex_mode, &t);
      if (!t.visible) continue;
      ++count;
      if (use_depth) add_h_b(t.raw_depth_residual, BSO_DEPTH_RESIDUAL_WEIGHT * bso_tukey_weight(t.raw_depth_residual, threshold_factor * BSO_DEPTH_TUKEY),
                             t.depth_jacobian, H, b);
      if (use_desc) {
        add_h_b(t.r1, threshold_factor * BSO_DESC_RESIDUAL_WEIGHT * bso_huber_weight(t.r1, BSO_DESC_HUBER), t.J1, H, b);
        if (!g_use_gradmag) add_h_b(t.r2, threshold_factor * BSO_DESC_RESIDUAL_WEIGHT * bso_huber_weight(t.r2, BSO_DESC_HUBER), t.J2, H, b);
      }
    }
  if (visible_count) *visible_count = count;
}

/* ComputeCostAndResidualCountFromImagesCUDA (GradientXY) BS/kernel_opt_pose.cu:939-1171 */
void bso_compute_cost_and_residual_count_from_images(
    int use_depth, int use_desc, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, float baseline_fx, float threshold_factor,
    const bslam_buffer2d* frame_depth, const bslam_buffer2d* frame_normals, const bslam_buffer2d* frame_color,
    const bslam_mat3x4* estimate_frame_T_surfel_frame,
    const bslam_buffer2d* surfel_depth, const bslam_buffer2d* surfel_normals, const bslam_buffer2d* surfel_color, int tex_mode,
    uint32_t* residual_count, double* cost) {
  uint32_t count = 0;
  double sum = 0;
  for (int y = 0; y < surfel_depth->height; ++y)
    for (int x = 0; x < surfel_depth->width; ++x) {
      pair_terms t;
      evaluate_pixel(x, y, use_depth, use_desc, 0, color_camera, depth_camera, baseline_fx, threshold_factor, estimate_frame_T_surfel_frame,
                     surfel_depth, surfel_normals, surfel_color, frame_depth, frame_normals, frame_color, tex_mode, &t);
      if (!t.visible) continue;
      if (use_depth) { count += 1; sum += (double)(BSO_DEPTH_RESIDUAL_WEIGHT * bso_tukey_residual(t.raw_depth_residual, threshold_factor * BSO_DEPTH_TUKEY)); }
      if (use_desc && g_use_gradmag) {            /* BS/kernel_opt_pose.cu:1289-1297: one residual */
        count += 1;
        sum += (double)(threshold_factor * BSO_DESC_RESIDUAL_WEIGHT * bso_huber_residual(t.r1, BSO_DESC_HUBER));
      } else if (use_desc) {
        count += 2;
        sum += (double)(threshold_factor * BSO_DESC_RESIDUAL_WEIGHT * bso_huber_residual(t.r1, BSO_DESC_HUBER));
        sum += (double)(threshold_factor * BSO_DESC_RESIDUAL_WEIGHT * bso_huber_residual(t.r2, BSO_DESC_HUBER));
      }
    }
  *residual_count = count;
  *cost = sum;
}

/* ---- pyramids + TrackFramePairwise ---- */
typedef struct { bslam_buffer2d depth, normals, color; } level_images;

static bslam_buffer2d alloc_img(int w, int h, size_t elem) {
  bslam_buffer2d b = {calloc((size_t)w * h, elem), h, w, (size_t)w * elem};
  return b;
}
static bslam_camera4f scaled_camera(const bslam_camera4f* c, double factor) {   /* CameraImpl::Scaled LV/camera.h:1696-1705, PixelMapping4::ScaleParameters :1086-1096 */
  bslam_camera4f s = *c;
  s.fx = (float)(c->fx * (float)factor); s.fy = (float)(c->fy * (float)factor);
  s.cx = (float)(c->cx * (float)factor); s.cy = (float)(c->cy * (float)factor);
  s.width = (int)(factor * c->width + 0.5f);
  s.height = (int)(factor * c->height + 0.5f);
  return s;
}

void bso_build_tracking_pyramids(
    int num_scales, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    const bslam_buffer2d* tracked_depth_u16, const bslam_buffer2d* tracked_normals, const bslam_buffer2d* tracked_color_uchar4,
    const bslam_buffer2d* base_depth_u16, const bslam_buffer2d* base_normals, const bslam_buffer2d* base_color_uchar4, int tex_mode,
    bslam_buffer2d* out /* [num_scales][6]: base depth, normals, color, tracked depth, normals, color */) {
  const int w = depth_camera->width, h = depth_camera->height;
  /* BadSlam::RunOdometry BS/bad_slam.cc:859-897 */
  bslam_buffer2d base_gradmag = alloc_img(base_color_uchar4->width, base_color_uchar4->height, 1);
  if (g_use_gradmag) bso_compute_sobel_gradient_magnitude(base_color_uchar4, &base_gradmag);
  else bso_brightness_from_color(base_color_uchar4, &base_gradmag);
  out[0] = alloc_img(w, h, 4);
  out[1] = *base_normals;
  out[2] = alloc_img(w, h, 1);
  bso_calibrate_depth_and_transform_color_to_depth(depth_camera, color_camera, dp, base_depth_u16, &base_gradmag, tex_mode, &out[0], &out[2]);
  bslam_buffer2d tracked_gradmag = alloc_img(tracked_color_uchar4->width, tracked_color_uchar4->height, 1);
  if (g_use_gradmag) bso_compute_sobel_gradient_magnitude(tracked_color_uchar4, &tracked_gradmag);
  else bso_brightness_from_color(tracked_color_uchar4, &tracked_gradmag);
  /* TrackFramePairwise with use_pyramid_level_0 (BS/pairwise_frame_tracking.cc:293-301); without it the level-0 tracked images
     stay empty placeholders (never read) and level 1 comes from CalibrateAndDownsampleImagesCUDA (:303-323) */
  out[3] = alloc_img(w, h, 4);
  out[4] = *tracked_normals;
  out[5] = alloc_img(w, h, 1);
  if (g_use_pyramid_level_0) {
    bso_calibrate_depth(dp, tracked_depth_u16, &out[3]);
    bso_set_to_read_mode_normalized(&tracked_gradmag, &out[5]);
  }
  for (int s = 1; s < num_scales; ++s) {   /* :312-341 */
    const int sw = (int)(w / pow(2, s)), sh = (int)(h / pow(2, s));
    bslam_buffer2d* cur = out + 6 * s;
    const bslam_buffer2d* prev = out + 6 * (s - 1);
    for (int side = 0; side < 2; ++side) {
      cur[3 * side + 0] = alloc_img(sw, sh, 4);
      cur[3 * side + 1] = alloc_img(sw, sh, 2);
      cur[3 * side + 2] = alloc_img(sw, sh, 1);
      if (side == 1 && s == 1 && !g_use_pyramid_level_0)
        bso_calibrate_and_downsample_images(depth_camera->width == color_camera->width, dp, tracked_depth_u16, tracked_normals, &tracked_gradmag, tex_mode,
                                            &cur[3], &cur[4], &cur[5]);
      else
        bso_downsample_images(&prev[3 * side + 0], &prev[3 * side + 1], &prev[3 * side + 2], tex_mode, &cur[3 * side + 0], &cur[3 * side + 1], &cur[3 * side + 2]);
    }
  }
  free(base_gradmag.address);
  free(tracked_gradmag.address);
}

void bso_free_tracking_pyramids(int num_scales, bslam_buffer2d* p) {
  for (int s = 0; s < num_scales; ++s)
    for (int i = 0; i < 6; ++i)
      if (!(s == 0 && (i == 1 || i == 4))) free(p[6 * s + i].address);   /* level-0 normals are the caller's */
}

int bso_is_scale_n_pose_estimation_converged(const float x[6], float scaling_factor) {   /* BS/convergence_analysis.h:56-63 */
  const float translation_threshold = 1e-08f, rotation_threshold = 1e-08f;
  float n = 0;
  for (int i = 0; i < 6; ++i) { const float v = (i < 3) ? x[i] : x[i] * (translation_threshold / rotation_threshold); n += v * v; }
  return n < scaling_factor * scaling_factor * translation_threshold;
}

/* TrackFramePairwise BS/pairwise_frame_tracking.cc:256-678 (use_pyramid_level_0 / use_gradmag: bso_set_tracking_variant) */
void bso_track_frame_pairwise(
    int num_scales, int use_depth, int use_desc, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    const bslam_buffer2d* tracked_depth_u16, const bslam_buffer2d* tracked_normals, const bslam_buffer2d* tracked_color_uchar4,
    const bslam_buffer2d* base_depth_u16, const bslam_buffer2d* base_normals, const bslam_buffer2d* base_color_uchar4, int tex_mode,
    int test_different_initial_estimates, const bslam_se3f* init1, const bslam_se3f* init2, bslam_se3f* out_base_T_frame, int* iterations_per_scale) {
  bslam_buffer2d* pyr = (bslam_buffer2d*)calloc((size_t)num_scales * 6, sizeof(bslam_buffer2d));
  bso_build_tracking_pyramids(num_scales, color_camera, depth_camera, dp, tracked_depth_u16, tracked_normals, tracked_color_uchar4, base_depth_u16,
                              base_normals, base_color_uchar4, tex_mode, pyr);
  const int kMaxIterationsPerScale = 30;
  bslam_se3f estimate = *init1, chosen_initial = *init1;
  for (int scale = num_scales - 1; scale >= (g_use_pyramid_level_0 ? 0 : 1); --scale) {   /* :367 */
    const float scaling_factor = (float)pow(2, scale);
    const bslam_camera4f tcc = scaled_camera(color_camera, (depth_camera->width == color_camera->width) ? (1.f / scaling_factor) : (2.f / scaling_factor));
    const bslam_camera4f tdc = scaled_camera(depth_camera, 1.f / scaling_factor);
    const float threshold_factor = scaling_factor;
    const bslam_buffer2d* L = pyr + 6 * scale;
    if (scale != num_scales - 1 || test_different_initial_estimates) {   /* :428-489 */
      const bslam_se3f last = (scale != num_scales - 1) ? estimate : *init1;
      const bslam_se3f other = (scale != num_scales - 1) ? chosen_initial : *init2;
      bslam_se3f inv;
      bslam_mat3x4 M;
      uint32_t count_last, count_other;
      double cost_last, cost_other;
      bso_se3_inverse(&last, &inv); bso_se3_matrix3x4(&inv, &M);
      bso_compute_cost_and_residual_count_from_images(use_depth, use_desc, &tcc, &tdc, dp->baseline_fx, threshold_factor, &L[3], &L[4], &L[5], &M, &L[0], &L[1], &L[2],
                                                      tex_mode, &count_last, &cost_last);
      bso_se3_inverse(&other, &inv); bso_se3_matrix3x4(&inv, &M);
      bso_compute_cost_and_residual_count_from_images(use_depth, use_desc, &tcc, &tdc, dp->baseline_fx, threshold_factor, &L[3], &L[4], &L[5], &M, &L[0], &L[1], &L[2],
                                                      tex_mode, &count_other, &cost_other);
      if (count_last > 2 * count_other) estimate = last;
      else if (count_other > 2 * count_last) estimate = other;
      else if ((float)cost_last < (float)cost_other) estimate = last;
      else estimate = other;
      if (scale == num_scales - 1) chosen_initial = estimate;
    }
    int iteration;
    for (iteration = 0; iteration < kMaxIterationsPerScale; ++iteration) {
      bslam_se3f inv;
      bslam_mat3x4 M;
      bso_se3_inverse(&estimate, &inv); bso_se3_matrix3x4(&inv, &M);
      double H64[21], b64[6];
      bso_accumulate_pose_coeffs_from_images(use_depth, use_desc, &tcc, &tdc, dp->baseline_fx, threshold_factor, &L[3], &L[4], &L[5], &M, &L[0], &L[1], &L[2],
                                             tex_mode, H64, b64, NULL);
      float H[21], b[6], x[6];
      for (int i = 0; i < 21; ++i) H[i] = (float)H64[i];
      for (int i = 0; i < 6; ++i) b[i] = (float)b64[i];
      bso_solve_ldlt_upper(6, H, b, x);
      float damping = 1.f;
      if (scale == num_scales - 2) damping = 0.5f;
      else if (scale == num_scales - 1) damping = 0.25f;
      float step[6];
      for (int i = 0; i < 6; ++i) step[i] = -damping * x[i];
      bslam_se3f d, next;
      bso_se3_exp(step, &d);
      bso_se3_mul(&estimate, &d, &next);
      estimate = next;
      if (bso_is_scale_n_pose_estimation_converged(x, scaling_factor)) { ++iteration; break; }
    }
    if (iterations_per_scale) iterations_per_scale[scale] = iteration;
  }
  bso_free_tracking_pyramids(num_scales, pyr);
  free(pyr);
  *out_base_T_frame = estimate;
}
