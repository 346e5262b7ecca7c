/*
 * bso_pcg.c -- ORACLE (test infrastructure only; see bslam_oracle.h).
 *
 * Serial restatement of BS/kernel_pcg.cu (BS/ = /root/reference/applications/badslam/src/badslam/).
 * Block reductions + atomicAdd of the reference become fp32 additions in
 * (keyframe, surfel-index) order; per-surfel entries are written in place.
 */
#include <math.h>

#include "bslam_oracle.h"
#include "bso_math.h"

#define K_DIAG_EPSILON 1e-8f   /* BS/kernel_pcg.cu:44 (PCGScalar = float, BS/kernels.cuh:62) */
#define K_A_PRIOR_WEIGHT 10.f  /* BS/kernel_pcg.cu:48 */
#define INVALID_INDEX 0xffffffffu

/* float64 sum of the same fp32 alpha_d terms of the last bso_pcg_step1 call: the fp32 serial sum
 * over ~1e5..1e7 terms carries ~1e-4 relative rounding error itself, so parity of the (tree-summed)
 * device value is judged against this one. */
static double g_alpha_d64;
double bso_pcg_last_alpha_d64(void) { return g_alpha_d64; }

/* get_kf_pose_unknown_index BS/direct_ba_pcg.cc:329-337 */
static uint32_t kf_pose_unknown_index(const bslam_pcg_layout* l, int keyframe_id) {
  if (keyframe_id == l->gauge_keyframe_id) return INVALID_INDEX;
  if (keyframe_id < l->gauge_keyframe_id) return (uint32_t)(6 * keyframe_id);
  return (uint32_t)(6 * (keyframe_id - 1));
}

static void sum_r_and_m(const bslam_pcg_vectors* v, uint32_t idx, float jacobian, float weight, float raw_residual) {
  /* BlockedAtomicSumRAndM / AtomicSumRAndM BS/kernel_pcg.cu:100-177 */
  const float weighted_jacobian = weight * jacobian;
  v->r[idx] += -1 * weighted_jacobian * raw_residual;
  v->M[idx] += jacobian * weighted_jacobian;
}
static void sum_r_and_m2(const bslam_pcg_vectors* v, uint32_t idx, float j1, float w1, float r1, float j2, float w2, float r2) {
  /* BlockedAtomicSumRAndM2 BS/kernel_pcg.cu:127-155 */
  const float wj1 = w1 * j1;
  const float wj2 = w2 * j2;
  v->r[idx] += -1 * wj1 * r1 + -1 * wj2 * r2;
  v->M[idx] += j1 * wj1 + j2 * wj2;
}

typedef struct {
  bso_unprojector unproj;
  bso_depth_to_color d2c;
} pcg_cams;

/* PCGInitCUDAKernel BS/kernel_pcg.cu:179-513 for one keyframe */
static void pcg_init_keyframe(const bslam_pcg_layout* l, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
                              const bslam_depth_params* dp, const bslam_keyframe_view* kf, uint32_t surfels_size,
                              const bslam_buffer2d* surfels, const bslam_pcg_vectors* v, int tex_mode, const pcg_cams* c) {
  const uint32_t kf_idx = kf_pose_unknown_index(l, kf->id);
  const int optimize_poses = (kf->id == l->gauge_keyframe_id) ? 0 : l->optimize_poses;   /* BS/direct_ba_pcg.cc:355 */
  const int per_surfel = l->use_descriptor_residuals ? 3 : 1;
  for (uint32_t i = 0; i < surfels_size; ++i) {
    bso_projection r;
    int visible = bso_surfel_projects_to_associated_pixel(i, surfels_size, surfels, &kf->depth, &kf->normals, dp, depth_camera, &c->unproj, &kf->frame_T_global, &r);
    if (!visible) continue;
    bso_f3 rn = bso_rotate34(&kf->frame_T_global, r.surfel_normal);
    if (l->use_depth_residuals) {
      float inv_stddev = bso_depth_inv_stddev(bso_unproj_nx(&c->unproj, r.px), bso_unproj_ny(&c->unproj, r.py), r.calibrated_depth, rn, dp->baseline_fx);
      bso_f3 lu = bso_unproject(&c->unproj, r.px, r.py, r.calibrated_depth);
      float raw = bso_depth_residual(inv_stddev, rn, lu, r.local_position);
      const float weight = bso_depth_weight(raw);
      int vis = visible;
      if (l->optimize_geometry) {                                           /* :217-221 */
        const float jp = bso_jac_depth_position(inv_stddev);
        v->r[l->surfel_unknown_start_index + per_surfel * i] -= jp * weight * raw;
        v->M[l->surfel_unknown_start_index + per_surfel * i] += jp * weight * jp;
      }
      if (optimize_poses) {                                                 /* :224-255 */
        float J[6];
        bso_jac_depth_pose(inv_stddev, rn, lu, J);
        for (int q = 0; q < 6; ++q) sum_r_and_m(v, kf_idx + q, J[q], weight, raw);
      }
      if (l->optimize_depth_intrinsics) {                                   /* :258-322 */
        int sparse_px = r.px / dp->sparse_surfel_cell_size;
        int sparse_py = r.py / dp->sparse_surfel_cell_size;
        float cfactor = BSO_AT(float, &dp->cfactor_buffer, sparse_py, sparse_px);
        float raw_inv_depth = 1.0f / (dp->raw_to_float_depth * BSO_AT(uint16_t, &kf->depth, r.py, r.px));
        float dj[6];
        const float corrected_inv_depth = bso_jac_depth_intrinsics(inv_stddev, r.calibrated_depth, r.px, r.py, bso_unproj_nx(&c->unproj, r.px),
                                                                   bso_unproj_ny(&c->unproj, r.py), r.surfel_normal, kf->frame_T_global.m, rn,
                                                                   cfactor, dp->a, raw_inv_depth, dj);
        if (fabsf(corrected_inv_depth) < 1e-4f) vis = 0;                    /* NOTE: stays false for the descriptor part too (:272) */
        if (vis) {
          const uint32_t d0 = l->depth_intrinsics_unknown_start_index;
          sum_r_and_m(v, d0 + 2, dj[2], weight, raw);
          sum_r_and_m(v, d0 + 3, dj[3], weight, raw);
          sum_r_and_m(v, d0 + 0, dj[0], weight, raw);
          sum_r_and_m(v, d0 + 1, dj[1], weight, raw);
          sum_r_and_m(v, d0 + 4, dj[4], weight, raw);
          sum_r_and_m(v, d0 + 5 + sparse_px + sparse_py * dp->cfactor_buffer.width, dj[5], weight, raw);
        }
        visible = vis;
      }
    }
    if (l->use_descriptor_residuals) {                                      /* :330-511 */
      bso_f2 color_pxy;
      visible = visible && bso_depth_to_color_pxy(r.pxy, &c->d2c, &color_pxy);
      if (!visible) continue;   /* every contribution below is gated by `visible` */
      bso_f2 t1, t2;
      bso_tangent_projections(r.global_position, r.surfel_normal, BSO_AT(float, surfels, BSLAM_SURFEL_RADIUS_SQUARED, i),
                              &kf->frame_T_global, color_camera->fx, color_camera->fy, color_camera->cx, color_camera->cy, &t1, &t2);
      float r1, r2;
      bso_raw_descriptor_residual(&kf->color, tex_mode, color_pxy, t1, t2,
                                  BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR1, i), BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR2, i), &r1, &r2);
      float gx1, gy1, gx2, gy2;
      bso_descriptor_jacobian_wrt_projected_position(&kf->color, color_pxy, t1, t2, &gx1, &gy1, &gx2, &gy2);
      gx1 *= color_camera->fx; gx2 *= color_camera->fx;
      gy1 *= color_camera->fy; gy2 *= color_camera->fy;
      const float w1 = bso_desc_weight(r1), w2 = bso_desc_weight(r2);
      const bso_f3 ls = r.local_position;
      if (l->optimize_geometry) {                                           /* :364-399 */
        float jp1 = bso_jac_desc_position(gx1, gy1, 1.f, 1.f, rn, ls);   /* gx, gy already carry fx, fy */
        float jp2 = bso_jac_desc_position(gx2, gy2, 1.f, 1.f, rn, ls);
        const uint32_t s0 = l->surfel_unknown_start_index + 3 * i;
        v->r[s0 + 0] -= jp1 * w1 * r1 + jp2 * w2 * r2;
        v->M[s0 + 0] += jp1 * w1 * jp1 + jp2 * w2 * jp2;
        const float j11 = -1, j12 = 0, j21 = 0, j22 = -1;
        v->r[s0 + 1] -= j11 * w1 * r1 + j12 * w2 * r2;
        v->M[s0 + 1] += j11 * w1 * j11 + j12 * w2 * j12;
        v->r[s0 + 2] -= j21 * w1 * r1 + j22 * w2 * r2;
        v->M[s0 + 2] += j21 * w1 * j21 + j22 * w2 * j22;
      }
      if (optimize_poses) {                                                 /* :402-459 */
        float J1[6], J2[6];
        bso_jac_desc_pose(gx1, gy1, ls, J1);
        bso_jac_desc_pose(gx2, gy2, ls, J2);
        for (int q = 0; q < 6; ++q) sum_r_and_m2(v, kf_idx + q, J1[q], w1, r1, J2[q], w2, r2);
      }
      if (l->optimize_color_intrinsics) {                                   /* :462-509 */
        const float g_x_1 = gx1 / color_camera->fx, g_y_1 = gy1 / color_camera->fy;
        const float g_x_2 = gx2 / color_camera->fx, g_y_2 = gy2 / color_camera->fy;
        const uint32_t c0 = l->color_intrinsics_unknown_start_index;
        float c1[4], c2[4];
        bso_jac_desc_color_intrinsics(g_x_1, g_y_1, bso_unproj_nx(&c->unproj, r.px), bso_unproj_ny(&c->unproj, r.py), c1);
        bso_jac_desc_color_intrinsics(g_x_2, g_y_2, bso_unproj_nx(&c->unproj, r.px), bso_unproj_ny(&c->unproj, r.py), c2);
        for (int q = 0; q < 4; ++q) sum_r_and_m2(v, c0 + q, c1[q], w1, r1, c2[q], w2, r2);
      }
    }
  }
}

void bso_pcg_init(const bslam_pcg_layout* layout,
                  const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
                  const bslam_depth_params* dp,
                  int keyframe_count, const bslam_keyframe_view* keyframes,
                  uint32_t surfels_size, const bslam_buffer2d* surfels,
                  const bslam_pcg_vectors* v, int tex_mode) {
  for (uint32_t i = 0; i < layout->unknown_count; ++i) { v->r[i] = 0.f; v->M[i] = 0.f; }   /* BS/direct_ba_pcg.cc:315-316 */
  if (surfels_size == 0) return;
  pcg_cams c;
  c.unproj = bso_make_unprojector(depth_camera);
  c.d2c = bso_make_depth_to_color(depth_camera, color_camera);
  for (int k = 0; k < keyframe_count; ++k)
    pcg_init_keyframe(layout, color_camera, depth_camera, dp, &keyframes[k], surfels_size, surfels, v, tex_mode, &c);
}

/* PCGInit2CUDAKernel BS/kernel_pcg.cu:564-606 */
void bso_pcg_init2(const bslam_pcg_layout* l, float a, const bslam_pcg_vectors* v) {
  float alpha_n = 0.f;
  for (uint32_t i = 0; i < l->unknown_count; ++i) {
    v->g[i] = 0;
    float r_value = v->r[i] + ((i == l->a_unknown_index) ? (-K_A_PRIOR_WEIGHT * K_A_PRIOR_WEIGHT * a) : 0);
    float p_value = r_value / (v->M[i] + K_DIAG_EPSILON + ((i == l->a_unknown_index) ? (K_A_PRIOR_WEIGHT * K_A_PRIOR_WEIGHT) : 0));
    v->p[i] = p_value;
    v->delta[i] = 0;
    alpha_n += r_value * p_value;
  }
  *v->alpha_n = alpha_n;
}

/* PCGStep1CUDAKernel BS/kernel_pcg.cu:645-1025 for one keyframe */
static void pcg_step1_keyframe(const bslam_pcg_layout* l, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
                               const bslam_depth_params* dp, const bslam_keyframe_view* kf, uint32_t surfels_size,
                               const bslam_buffer2d* surfels, const bslam_pcg_vectors* v, int tex_mode, const pcg_cams* c) {
  const uint32_t kf_idx = kf_pose_unknown_index(l, kf->id);
  const int optimize_poses = (kf->id == l->gauge_keyframe_id) ? 0 : l->optimize_poses;
  const int per_surfel = l->use_descriptor_residuals ? 3 : 1;
  for (uint32_t i = 0; i < surfels_size; ++i) {
    bso_projection r;
    int visible = bso_surfel_projects_to_associated_pixel(i, surfels_size, surfels, &kf->depth, &kf->normals, dp, depth_camera, &c->unproj, &kf->frame_T_global, &r);
    if (!visible) continue;
    bso_f3 rn = bso_rotate34(&kf->frame_T_global, r.surfel_normal);
    if (l->use_depth_residuals) {
      float inv_stddev = bso_depth_inv_stddev(bso_unproj_nx(&c->unproj, r.px), bso_unproj_ny(&c->unproj, r.py), r.calibrated_depth, rn, dp->baseline_fx);
      bso_f3 lu = bso_unproject(&c->unproj, r.px, r.py, r.calibrated_depth);
      float raw = bso_depth_residual(inv_stddev, rn, lu, r.local_position);
      const float weight = bso_depth_weight(raw);
      float sum = 0;
      float geometry_jacobian = 0;
      float pose_jacobian[6] = {0, 0, 0, 0, 0, 0};
      float dgi[5] = {0, 0, 0, 0, 0};
      float cfactor_entry_jacobian = 0;
      if (l->optimize_geometry) {
        geometry_jacobian = bso_jac_depth_position(inv_stddev);
        sum += geometry_jacobian * v->p[l->surfel_unknown_start_index + per_surfel * i + 0];
      }
      if (optimize_poses) {
        bso_jac_depth_pose(inv_stddev, rn, lu, pose_jacobian);
        for (int q = 0; q < 6; ++q) sum += pose_jacobian[q] * v->p[kf_idx + q];
      }
      int djv = 0;
      uint32_t cfactor_entry_index = 0;
      if (l->optimize_depth_intrinsics) {
        int sparse_px = r.px / dp->sparse_surfel_cell_size;
        int sparse_py = r.py / dp->sparse_surfel_cell_size;
        float cfactor = BSO_AT(float, &dp->cfactor_buffer, sparse_py, sparse_px);
        float raw_inv_depth = 1.0f / (dp->raw_to_float_depth * BSO_AT(uint16_t, &kf->depth, r.py, r.px));
        float dj[6];
        const float corrected_inv_depth = bso_jac_depth_intrinsics(inv_stddev, r.calibrated_depth, r.px, r.py, bso_unproj_nx(&c->unproj, r.px),
                                                                   bso_unproj_ny(&c->unproj, r.py), r.surfel_normal, kf->frame_T_global.m, rn,
                                                                   cfactor, dp->a, raw_inv_depth, dj);
        djv = !(fabsf(corrected_inv_depth) < 1e-4f);
        if (djv) {
          const uint32_t d0 = l->depth_intrinsics_unknown_start_index;
          for (int q = 0; q < 5; ++q) dgi[q] = dj[q];
          sum += dgi[2] * v->p[d0 + 2];
          sum += dgi[3] * v->p[d0 + 3];
          sum += dgi[0] * v->p[d0 + 0];
          sum += dgi[1] * v->p[d0 + 1];
          sum += dgi[4] * v->p[d0 + 4];
          cfactor_entry_index = d0 + 5 + sparse_px + sparse_py * dp->cfactor_buffer.width;
          cfactor_entry_jacobian = dj[5];
          sum += cfactor_entry_jacobian * v->p[cfactor_entry_index];
        }
      }
      *v->alpha_d += sum * weight * sum;
      g_alpha_d64 += (double)(sum * weight * sum);
      sum *= weight;
      if (l->optimize_geometry) v->g[l->surfel_unknown_start_index + per_surfel * i + 0] += geometry_jacobian * sum;
      if (optimize_poses) for (int k = 0; k < 6; ++k) v->g[kf_idx + k] += pose_jacobian[k] * sum;
      if (l->optimize_depth_intrinsics && djv) {
        for (int k = 0; k < 5; ++k) v->g[l->depth_intrinsics_unknown_start_index + k] += dgi[k] * sum;
        v->g[cfactor_entry_index] += cfactor_entry_jacobian * sum;
      }
    }
    if (l->use_descriptor_residuals) {
      bso_f2 color_pxy;
      visible = visible && bso_depth_to_color_pxy(r.pxy, &c->d2c, &color_pxy);
      if (!visible) continue;
      bso_f2 t1, t2;
      bso_tangent_projections(r.global_position, r.surfel_normal, BSO_AT(float, surfels, BSLAM_SURFEL_RADIUS_SQUARED, i),
                              &kf->frame_T_global, color_camera->fx, color_camera->fy, color_camera->cx, color_camera->cy, &t1, &t2);
      float r1, r2;
      bso_raw_descriptor_residual(&kf->color, tex_mode, color_pxy, t1, t2,
                                  BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR1, i), BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR2, i), &r1, &r2);
      float gx1, gy1, gx2, gy2;
      bso_descriptor_jacobian_wrt_projected_position(&kf->color, color_pxy, t1, t2, &gx1, &gy1, &gx2, &gy2);
      gx1 *= color_camera->fx; gx2 *= color_camera->fx;
      gy1 *= color_camera->fy; gy2 *= color_camera->fy;
      const float w1 = bso_desc_weight(r1), w2 = bso_desc_weight(r2);
      const bso_f3 ls = r.local_position;
      float sum_1 = 0, sum_2 = 0;
      float gj1 = 0, gj2 = 0;
      float pj1[6] = {0, 0, 0, 0, 0, 0}, pj2[6] = {0, 0, 0, 0, 0, 0};
      float cj1[4] = {0, 0, 0, 0}, cj2[4] = {0, 0, 0, 0};
      const uint32_t s0 = l->surfel_unknown_start_index + 3 * i;
      if (l->optimize_geometry) {
        gj1 = bso_jac_desc_position(gx1, gy1, 1.f, 1.f, rn, ls);
        gj2 = bso_jac_desc_position(gx2, gy2, 1.f, 1.f, rn, ls);
        float p = v->p[s0 + 0];
        sum_1 += gj1 * p;
        sum_2 += gj2 * p;
        p = v->p[s0 + 1];
        sum_1 += -1.f * p;
        p = v->p[s0 + 2];
        sum_2 += -1.f * p;
      }
      if (optimize_poses) {
        bso_jac_desc_pose(gx1, gy1, ls, pj1);
        bso_jac_desc_pose(gx2, gy2, ls, pj2);
        for (int q = 0; q < 6; ++q) {
          const float p = v->p[kf_idx + q];
          sum_1 += pj1[q] * p;
          sum_2 += pj2[q] * p;
        }
      }
      if (l->optimize_color_intrinsics) {
        const float g_x_1 = gx1 / color_camera->fx, g_y_1 = gy1 / color_camera->fy;
        const float g_x_2 = gx2 / color_camera->fx, g_y_2 = gy2 / color_camera->fy;
        const uint32_t c0 = l->color_intrinsics_unknown_start_index;
        bso_jac_desc_color_intrinsics(g_x_1, g_y_1, bso_unproj_nx(&c->unproj, r.px), bso_unproj_ny(&c->unproj, r.py), cj1);
        bso_jac_desc_color_intrinsics(g_x_2, g_y_2, bso_unproj_nx(&c->unproj, r.px), bso_unproj_ny(&c->unproj, r.py), cj2);
        for (int q = 0; q < 4; ++q) {
          const float p = v->p[c0 + q];
          sum_1 += cj1[q] * p;
          sum_2 += cj2[q] * p;
        }
      }
      *v->alpha_d += sum_1 * w1 * sum_1 + sum_2 * w2 * sum_2;
      g_alpha_d64 += (double)(sum_1 * w1 * sum_1 + sum_2 * w2 * sum_2);
      sum_1 *= w1;
      sum_2 *= w2;
      if (l->optimize_geometry) {
        v->g[s0 + 0] += gj1 * sum_1 + gj2 * sum_2;
        v->g[s0 + 1] += -1.f * sum_1 + 0.f * sum_2;
        v->g[s0 + 2] += 0.f * sum_1 + -1.f * sum_2;
      }
      if (optimize_poses) for (int k = 0; k < 6; ++k) v->g[kf_idx + k] += pj1[k] * sum_1 + pj2[k] * sum_2;
      if (l->optimize_color_intrinsics) for (int k = 0; k < 4; ++k) v->g[l->color_intrinsics_unknown_start_index + k] += cj1[k] * sum_1 + cj2[k] * sum_2;
    }
  }
}

void bso_pcg_step1(const bslam_pcg_layout* layout,
                   const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
                   const bslam_depth_params* dp,
                   int keyframe_count, const bslam_keyframe_view* keyframes,
                   uint32_t surfels_size, const bslam_buffer2d* surfels,
                   const bslam_pcg_vectors* v, int clear_g, int tex_mode) {
  *v->alpha_d = 0.f;                                               /* BS/direct_ba_pcg.cc:383 */
  g_alpha_d64 = 0.0;
  if (clear_g) for (uint32_t i = 0; i < layout->unknown_count; ++i) v->g[i] = 0.f;   /* :393 */
  if (surfels_size == 0) return;
  pcg_cams c;
  c.unproj = bso_make_unprojector(depth_camera);
  c.d2c = bso_make_depth_to_color(depth_camera, color_camera);
  for (int k = 0; k < keyframe_count; ++k) {
    pcg_step1_keyframe(layout, color_camera, depth_camera, dp, &keyframes[k], surfels_size, surfels, v, tex_mode, &c);
    /* AddAlphaDEpsilonTermsCUDAKernel is launched inside PCGStep1CUDA, i.e. once per
     * keyframe (quirk Q7; BS/kernel_pcg.cu:1027-1048,1102-1113) */
    float acc = 0.f;
    for (uint32_t i = 0; i < layout->unknown_count; ++i) {
      float p = v->p[i];
      acc += (K_DIAG_EPSILON + ((i == layout->a_unknown_index) ? (K_A_PRIOR_WEIGHT * K_A_PRIOR_WEIGHT) : 0)) * p * p;
    }
    *v->alpha_d += acc;
    g_alpha_d64 += (double)acc;
  }
}

/* PCGStep2CUDAKernel BS/kernel_pcg.cu:1116-1170 */
void bso_pcg_step2(const bslam_pcg_layout* l, const bslam_pcg_vectors* v, float* beta_n_host) {
  float beta_n = 0.f;
  const float alpha = (*v->alpha_d >= 1e-35f) ? (*v->alpha_n / *v->alpha_d) : 0;
  for (uint32_t i = 0; i < l->unknown_count; ++i) {
    float p_value = v->p[i];
    v->delta[i] += alpha * p_value;
    float r_value = v->r[i];
    r_value -= alpha * (v->g[i] + (K_DIAG_EPSILON + ((i == l->a_unknown_index) ? (K_A_PRIOR_WEIGHT * K_A_PRIOR_WEIGHT) : 0)) * p_value);
    v->r[i] = r_value;
    float z_value = r_value / (v->M[i] + K_DIAG_EPSILON + ((i == l->a_unknown_index) ? (K_A_PRIOR_WEIGHT * K_A_PRIOR_WEIGHT) : 0));
    v->g[i] = z_value;
    beta_n += z_value * r_value;
  }
  *v->beta_n = beta_n;
  if (beta_n_host) *beta_n_host = beta_n;
}

/* PCGStep3CUDAKernel BS/kernel_pcg.cu:1211-1230 */
void bso_pcg_step3(const bslam_pcg_layout* l, const bslam_pcg_vectors* v) {
  const float beta = (*v->alpha_n >= 1e-35f) ? (*v->beta_n / *v->alpha_n) : 0;
  for (uint32_t i = 0; i < l->unknown_count; ++i) v->p[i] = v->g[i] + beta * v->p[i];
}

/* UpdateSurfelsFromPCGDeltaCUDAKernel BS/kernel_pcg.cu:1305-1333 */
void bso_update_surfels_from_pcg_delta(uint32_t surfels_size, const bslam_buffer2d* surfels,
                                       int use_descriptor_residuals, uint32_t s0, const float* delta) {
  const int per = use_descriptor_residuals ? 3 : 1;
  for (uint32_t i = 0; i < surfels_size; ++i) {
    float t = delta[s0 + per * i];
    if (t != 0) {
      bso_f3 p = bso_surfel_position(surfels, i);
      bso_f3 n = bso_surfel_normal(surfels, i);
      bso_surfel_set_position(surfels, i, bso_add(p, bso_scale(t, n)));
    }
    if (use_descriptor_residuals) {
      float d1 = BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR1, i);
      d1 += delta[s0 + 3 * i + 1];
      BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR1, i) = fmaxf(-180.f, fminf(180.f, d1));
      float d2 = BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR2, i);
      d2 += delta[s0 + 3 * i + 2];
      BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR2, i) = fmaxf(-180.f, fminf(180.f, d2));
    }
  }
}

/* UpdateCFactorsFromPCGDeltaCUDAKernel BS/kernel_pcg.cu:1361-1372 */
void bso_update_cfactors_from_pcg_delta(const bslam_buffer2d* cf, uint32_t start, const float* delta) {
  for (int y = 0; y < cf->height; ++y)
    for (int x = 0; x < cf->width; ++x)
      BSO_AT(float, cf, y, x) += delta[start + (uint32_t)(y * cf->width + x)];
}
