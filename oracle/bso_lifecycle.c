/*
 * bso_lifecycle.c -- ORACLE (test infrastructure only; see bslam_oracle.h).
 *
 * Surfel lifecycle on the CPU: supporting surfels + merge, creation (with the observation-count
 * filter), deletion + radius update, compaction.  BS/ = /root/reference/applications/badslam/src/badslam/
 *
 * The reference resolves cell ownership with atomicCAS races; this restatement fixes the interleaving to
 * "lowest surfel index first" (supporting surfels) and "raster order first" (creation), which is one of the
 * outcomes the reference can produce.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bslam_oracle.h"
#include "bso_math.h"

#define BSO_INVALID_INDEX 0xffffffffu
#define BSO_NAN_BITS 0x7fffffffu   /* CUDART_NAN_F */

static int surfel_is_deleted(const bslam_buffer2d* s, uint32_t i) {
  uint32_t bits;
  memcpy(&bits, &BSO_AT(float, s, BSLAM_SURFEL_X, i), 4);
  return bits == BSO_NAN_BITS;
}
static void surfel_mark_deleted(const bslam_buffer2d* s, uint32_t i) {
  const uint32_t bits = BSO_NAN_BITS;
  memcpy(&BSO_AT(float, s, BSLAM_SURFEL_X, i), &bits, 4);
}

/* IsAssociatedWithPixel<true, false> (BS/surfel_projection_nvcc_only.cuh:49-127): the free-space variant. */
static int is_associated_fs(const bslam_buffer2d* surfels, uint32_t surfel_index, bso_f3 local_position,
                            const bslam_mat3x4* frame_T_global, const bslam_buffer2d* normals_buffer, int px, int py,
                            const bslam_depth_params* dp, uint16_t measured_depth, const bso_unprojector* unproj,
                            int* is_free_space_violation) {
  if (measured_depth & BSLAM_INVALID_DEPTH_BIT) return 0;
  float calibrated_depth = bso_raw_to_calibrated_depth(
      dp->a, BSO_AT(float, &dp->cfactor_buffer, py / dp->sparse_surfel_cell_size, px / dp->sparse_surfel_cell_size),
      dp->raw_to_float_depth, measured_depth);
  bso_f3 local_normal_s = bso_rotate34(frame_T_global, bso_surfel_normal(surfels, surfel_index));
  float stddev = bso_depth_stddev(bso_unproj_nx(unproj, px), bso_unproj_ny(unproj, py), calibrated_depth, local_normal_s, dp->baseline_fx);
  const float thr = BSO_DEPTH_TUKEY * stddev;
  float depth_difference = calibrated_depth - local_position.z;
  if (depth_difference > thr) { *is_free_space_violation = 1; return 0; }
  else if (depth_difference < -thr) return 0;
  float surfel_distance = bso_norm(local_position);
  float dot_angle = (1.0f / surfel_distance) * bso_dot(local_position, local_normal_s);
  if (dot_angle > 0) return 0;
  bso_f3 local_normal = bso_u16_to_image_space_normal(BSO_AT(uint16_t, normals_buffer, py, px));
  if (bso_dot(local_normal_s, local_normal) < BSO_COS_NORMAL_COMPAT) return 0;
  return 1;
}

/* SurfelProjectsToAssociatedPixel(SurfelProjectionResultXYFreeSpace) (:482-511) */
static int projects_fs(uint32_t i, const bslam_buffer2d* surfels, const bslam_keyframe_view* kf, const bslam_depth_params* dp,
                       const bslam_camera4f* cam, const bso_unprojector* unproj, int* px, int* py, int* fsv) {
  *fsv = 0;
  bso_f3 local;
  if (!bso_mul34_if_z_positive(&kf->frame_T_global, bso_surfel_position(surfels, i), &local)) return 0;
  bso_f2 pxy;
  if (!bso_project_surfel_to_image(kf->depth.width, kf->depth.height, cam, local, px, py, &pxy)) return 0;
  return is_associated_fs(surfels, i, local, &kf->frame_T_global, &kf->normals, *px, *py, dp, BSO_AT(uint16_t, &kf->depth, *py, *px), unproj, fsv);
}

void bso_determine_supporting_surfels(
    int merge, float merge_dist_factor, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    const bslam_keyframe_view* kf, uint32_t surfels_size, const bslam_buffer2d* surfels,
    uint32_t* supporting0, uint32_t* supporting1, uint32_t* supporting2, uint32_t* surfel_count) {
  const int cell = dp->sparse_surfel_cell_size;
  const int cw = (kf->depth.width - 1) / cell + 1, ch = (kf->depth.height - 1) / cell + 1;
  uint32_t* sup[3] = {supporting0, supporting1, supporting2};
  for (int b = 0; b < 3; ++b) if (sup[b]) memset(sup[b], 0xff, sizeof(uint32_t) * (size_t)cw * ch);   /* Clear(kInvalidIndex) */
  if (surfels_size == 0) return;
  const float cell_merge_dist_squared = cell * cell * merge_dist_factor * merge_dist_factor;   /* BS/kernel_supporting_surfels.cc:74-76 */
  const float cos_thr = BSO_COS_NORMAL_COMPAT;
  bso_unprojector unproj = bso_make_unprojector(depth_camera);
  uint32_t deleted_count = 0;
  for (uint32_t i = 0; i < surfels_size; ++i) {                     /* :45-97, lowest index first */
    bso_projection r;
    if (!bso_surfel_projects_to_associated_pixel(i, surfels_size, surfels, &kf->depth, &kf->normals, dp, depth_camera, &unproj, &kf->frame_T_global, &r)) continue;
    const size_t c = (size_t)(r.py / cell) * cw + (r.px / cell);
    int deleted = 0;
    for (int b = 0; b < 3; ++b) {
      if (!sup[b]) break;
      const uint32_t sup_index = sup[b][c];
      if (sup_index == BSO_INVALID_INDEX) { sup[b][c] = i; break; }
      else if (merge) {
        bso_f3 sup_normal = bso_surfel_normal(surfels, sup_index);
        bso_f3 this_normal = bso_surfel_normal(surfels, i);
        if (bso_dot(sup_normal, this_normal) > cos_thr) {
          bso_f3 sp = bso_surfel_position(surfels, sup_index), tp = bso_surfel_position(surfels, i);
          float sr = BSO_AT(float, surfels, BSLAM_SURFEL_RADIUS_SQUARED, sup_index), tr = BSO_AT(float, surfels, BSLAM_SURFEL_RADIUS_SQUARED, i);
          float min_r = fminf(sr, tr);
          bso_f3 d = bso_sub(sp, tp);
          if (bso_sqlen(d) < min_r * cell_merge_dist_squared) { surfel_mark_deleted(surfels, i); deleted = 1; }   /* no break: quirk kept */
        }
      }
    }
    deleted_count += (uint32_t)deleted;
  }
  if (merge && surfel_count) *surfel_count -= deleted_count;
}

/* colour texture, one channel (same filter model as bso_tex_w) */
static float tex_channel(const bslam_buffer2d* color, float x, float y, int mode, int chn) {
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fx = floorf(xb), fy = floorf(yb);
  float a = xb - fx, b = yb - fy;
  if (mode == BSLAM_TEX_FIXED_POINT_1_8) {
    a = floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f);
    b = floorf(b * 256.0f + 0.5f) * (1.0f / 256.0f);
  }
  const int i = (int)fminf(fmaxf(fx, -2.0f), (float)color->width);
  const int j = (int)fminf(fmaxf(fy, -2.0f), (float)color->height);
  float t[4];
  for (int q = 0; q < 4; ++q) {
    int ix = i + (q & 1), iy = j + (q >> 1);
    if (ix < 0) ix = 0;
    if (iy < 0) iy = 0;
    if (ix > color->width - 1) ix = color->width - 1;
    if (iy > color->height - 1) iy = color->height - 1;
    t[q] = ((const uint8_t*)color->address + (size_t)iy * color->pitch + 4 * (size_t)ix)[chn] * (1.0f / 255.0f);
  }
  return (((1.0f - a) * (1.0f - b) * t[0] + a * (1.0f - b) * t[1]) + (1.0f - a) * b * t[2]) + a * b * t[3];
}

uint32_t bso_create_surfels_for_keyframe_ex(
    int filter_new_surfels, int min_observation_count,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp, const bslam_keyframe_view* kf, const bslam_mat3x4* global_T_frame,
    int covis_count, const bslam_keyframe_view* covis_keyframes, const bslam_mat3x4* covis_T_frame,
    uint32_t* surfels_size, uint32_t max_surfels, const bslam_buffer2d* surfels, int tex_mode) {
  const int w = kf->depth.width, h = kf->depth.height;
  const int cell = dp->sparse_surfel_cell_size;
  const int cw = (w - 1) / cell + 1, ch = (h - 1) / cell + 1;
  bso_unprojector unproj = bso_make_unprojector(depth_camera);
  bso_depth_to_color d2c = bso_make_depth_to_color(depth_camera, color_camera);

  /* DetermineSupportingSurfelsCUDA (BS/direct_ba.cc:349-358): only "cell occupied or not" matters below */
  uint32_t* sup0 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cw * ch);
  bso_determine_supporting_surfels(0, 0.f, depth_camera, dp, kf, *surfels_size, surfels, sup0, NULL, NULL, NULL);

  /* CreateSurfelsForKeyframeCUDASerializingKernel BS/kernel_create_surfels.cu:41-72: raster order wins the cell */
  uint8_t* flag = (uint8_t*)calloc((size_t)w * h, 1);
  uint32_t count = 0;
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < w; ++x) {
      const int kBorder = 1;
      if (!(x >= kBorder && y >= kBorder && x < w - kBorder && y < h - kBorder)) continue;
      if (BSO_AT(uint16_t, &kf->depth, y, x) & BSLAM_INVALID_DEPTH_BIT) continue;
      uint32_t* occ = &sup0[(size_t)(y / cell) * cw + (x / cell)];
      if (*occ != BSO_INVALID_INDEX) continue;
      *occ = 0;
      flag[(size_t)y * w + x] = 1;
      ++count;
    }
  }
  free(sup0);
  if (count == 0) { free(flag); return 0; }

  if (filter_new_surfels) {   /* BS/kernel_create_surfels.cc:84-136 */
    uint32_t n = 0;
    for (size_t seq = 0; seq < (size_t)w * h; ++seq) {
      if (!flag[seq]) continue;
      const int y = (int)(seq / w), x = (int)(seq - (size_t)y * w);
      uint32_t observations = 1, violations = 0;   /* WriteNewSurfelIndexAndInitializeObservations :163-181 */
      float calibrated_depth = bso_raw_to_calibrated_depth(dp->a, BSO_AT(float, &dp->cfactor_buffer, y / cell, x / cell), dp->raw_to_float_depth,
                                                           BSO_AT(uint16_t, &kf->depth, y, x));
      bso_f3 input_position = bso_unproject(&unproj, x, y, calibrated_depth);
      for (int c = 0; c < covis_count; ++c) {     /* CountObservationsForNewSurfelsCUDAKernel :206-262 */
        const bslam_keyframe_view* ck = &covis_keyframes[c];
        bso_f3 local;
        if (!bso_mul34_if_z_positive(&covis_T_frame[c], input_position, &local)) continue;
        int px, py;
        bso_f2 pxy;
        if (!bso_project_surfel_to_image(ck->depth.width, ck->depth.height, depth_camera, local, &px, &py, &pxy)) continue;
        /* IsAssociatedWithPixel<true>(…image normals…) BS/surfel_projection_nvcc_only.cuh:130-215 */
        const uint16_t measured = BSO_AT(uint16_t, &ck->depth, py, px);
        if (measured & BSLAM_INVALID_DEPTH_BIT) continue;
        float pixel_depth = bso_raw_to_calibrated_depth(dp->a, BSO_AT(float, &dp->cfactor_buffer, py / cell, px / cell), dp->raw_to_float_depth, measured);
        bso_f3 n_local = bso_rotate34(&covis_T_frame[c], bso_u16_to_image_space_normal(BSO_AT(uint16_t, &kf->normals, y, x)));
        float stddev = bso_depth_stddev(bso_unproj_nx(&unproj, px), bso_unproj_ny(&unproj, py), pixel_depth, n_local, dp->baseline_fx);
        const float thr = BSO_DEPTH_TUKEY * stddev;
        float diff = pixel_depth - local.z;
        if (diff > thr) { ++violations; continue; }
        else if (diff < -thr) continue;
        float dist = bso_norm(local);
        if ((1.0f / dist) * bso_dot(local, n_local) > 0) continue;
        bso_f3 pn = bso_u16_to_image_space_normal(BSO_AT(uint16_t, &ck->normals, py, px));
        if (bso_dot(n_local, pn) < BSO_COS_NORMAL_COMPAT) continue;
        ++observations;
      }
      /* u16 counters in the reference; FilterNewSurfelsCUDAKernel :288-305 */
      if ((uint16_t)observations < (uint16_t)min_observation_count || (uint16_t)violations > (uint16_t)observations) flag[seq] = 0;
      else ++n;
    }
    count = n;
    if (count == 0) { free(flag); return 0; }
  }
  if ((uint64_t)*surfels_size + count > max_surfels) { free(flag); return 0; }   /* BS/kernel_create_surfels.cc:162-165 */

  /* CreateSurfelsForKeyframeCUDACreationAppendKernel :357-385 + CreateNewSurfel :96-161 (append in raster order = scan order) */
  uint32_t created = 0;
  for (size_t seq = 0; seq < (size_t)w * h; ++seq) {
    if (!flag[seq]) continue;
    const int y = (int)(seq / w), x = (int)(seq - (size_t)y * w);
    const uint32_t si = *surfels_size + created;
    ++created;
    float calibrated_depth = bso_raw_to_calibrated_depth(dp->a, BSO_AT(float, &dp->cfactor_buffer, y / cell, x / cell), dp->raw_to_float_depth,
                                                         BSO_AT(uint16_t, &kf->depth, y, x));
    bso_f3 gp = bso_mul34(global_T_frame, bso_unproject(&unproj, x, y, calibrated_depth));
    bso_surfel_set_position(surfels, si, gp);
    bso_f3 gn = bso_rotate34(global_T_frame, bso_u16_to_image_space_normal(BSO_AT(uint16_t, &kf->normals, y, x)));
    bso_surfel_set_normal(surfels, si, gn);
    float radius_squared = bso_half_to_float(BSO_AT(uint16_t, &kf->radius, y, x));
    BSO_AT(float, surfels, BSLAM_SURFEL_RADIUS_SQUARED, si) = radius_squared;
    bso_f2 pc = {x + 0.5f, y + 0.5f};
    bso_f2 color_pxy;
    bso_depth_to_color_pxy(pc, &d2c, &color_pxy);
    /* tex2D<float4> -> make_uchar4(255.f * c): truncating conversion (:150-154) */
    uint8_t col[4];
    for (int chn = 0; chn < 3; ++chn) col[chn] = (uint8_t)bso_f2i(255.f * tex_channel(&kf->color, color_pxy.x, color_pxy.y, tex_mode, chn));
    col[3] = 0;
    memcpy(&BSO_AT(float, surfels, BSLAM_SURFEL_COLOR, si), col, 4);
    bso_f2 t1, t2;
    /* note: the UNQUANTISED global normal gn is used here, as in the reference (:124-131) */
    bso_tangent_projections(gp, gn, radius_squared, &kf->frame_T_global, color_camera->fx, color_camera->fy, color_camera->cx, color_camera->cy, &t1, &t2);
    float d1, d2;
    bso_raw_descriptor_residual(&kf->color, tex_mode, color_pxy, t1, t2, 0, 0, &d1, &d2);
    BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR1, si) = d1;
    BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR2, si) = d2;
  }
  free(flag);
  *surfels_size += created;
  return created;
}

uint32_t bso_create_surfels_for_keyframe(
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp, const bslam_keyframe_view* kf, const bslam_se3f* global_T_frame,
    uint32_t* surfels_size, uint32_t max_surfels, const bslam_buffer2d* surfels, int tex_mode) {
  bslam_mat3x4 M;
  bso_se3_matrix3x4(global_T_frame, &M);
  return bso_create_surfels_for_keyframe_ex(0, 0, color_camera, depth_camera, dp, kf, &M, 0, NULL, NULL, surfels_size, max_surfels, surfels, tex_mode);
}

void bso_delete_surfels_and_update_radii(
    int min_observation_count, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    int keyframe_count, const bslam_keyframe_view* keyframes, uint32_t* surfel_count, uint32_t surfels_size,
    const bslam_buffer2d* surfels) {
  bso_unprojector unproj = bso_make_unprojector(depth_camera);
  uint32_t deleted_count = 0;
  for (uint32_t i = 0; i < surfels_size; ++i) {
    float observations = 0, violations = 0, min_radius = INFINITY;   /* Reset… BS/kernel_delete_surfels.cu:45-58 */
    for (int k = 0; k < keyframe_count; ++k) {                        /* CountObservationsAndFreeSpaceViolations :85-103 */
      const bslam_keyframe_view* kf = &keyframes[k];
      int px, py, fsv;
      if (projects_fs(i, surfels, kf, dp, depth_camera, &unproj, &px, &py, &fsv)) {
        observations += 1.f;
        min_radius = fminf(min_radius, bso_half_to_float(BSO_AT(uint16_t, &kf->radius, py, px)));
      } else if (fsv) {
        violations += 1.f;
      }
    }
    if (observations < min_observation_count || violations > observations) {   /* MarkDeletedSurfels :134-160 */
      if (!surfel_is_deleted(surfels, i)) { surfel_mark_deleted(surfels, i); ++deleted_count; }
    } else {
      BSO_AT(float, surfels, BSLAM_SURFEL_RADIUS_SQUARED, i) = min_radius;
    }
  }
  *surfel_count -= deleted_count;
}

void bso_compact_surfels(uint32_t surfel_count, uint32_t* surfels_size, const bslam_buffer2d* surfels,
                         const bslam_buffer2d* active_surfels) {
  const uint32_t n = *surfels_size;
  if (n == surfel_count) return;                                    /* BS/kernel_compact_surfels.cu:186-188 */
  const uint32_t free_spot_count = n - surfel_count;
  uint32_t* free_list = (uint32_t*)malloc(sizeof(uint32_t) * (free_spot_count ? free_spot_count : 1));
  uint8_t* invalid = (uint8_t*)malloc(n);                           /* FlagInvalidSurfelsInAccum2CUDAKernel :97-106 */
  for (uint32_t i = 0; i < n; ++i) invalid[i] = (uint8_t)surfel_is_deleted(surfels, i);
  uint32_t f = 0;
  for (uint32_t i = 0; i < n; ++i)                                  /* forward exclusive scan of the invalid flags + free spot list (:203-232) */
    if (invalid[i]) { if (f < free_spot_count) free_list[f] = i; ++f; }
  uint32_t valid_after = 0;                                         /* reverse exclusive scan of the valid flags (:234-243) */
  for (uint32_t ii = n; ii-- > 0;) {
    if (invalid[ii]) continue;
    const uint32_t reverse_index = valid_after++;
    if (reverse_index < free_spot_count) {                          /* CompactSurfelsCUDAKernel :151-176 */
      const uint32_t spot = free_list[reverse_index];
      if (spot < ii) {
        for (int row = 0; row < BSLAM_SURFEL_DATA_ATTRIBUTE_COUNT; ++row) BSO_AT(float, surfels, row, spot) = BSO_AT(float, surfels, row, ii);
        if (active_surfels) BSO_AT(uint8_t, active_surfels, 0, spot) = BSO_AT(uint8_t, active_surfels, 0, ii);
      }
    }
  }
  free(free_list);
  free(invalid);
  *surfels_size = surfel_count;
}
