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
      float raw = inv_stddev * bso_dot(rn, bso_sub(lu, r.local_position));
      const float weight = bso_depth_weight(raw);
      int vis = visible;
      if (l->optimize_geometry) {                                           /* :217-221 */
        const float jp = -inv_stddev;
        v->r[l->surfel_unknown_start_index + per_surfel * i] -= jp * weight * raw;
        v->M[l->surfel_unknown_start_index + per_surfel * i] += jp * weight * jp;
      }
      if (optimize_poses) {                                                 /* :224-255 */
        sum_r_and_m(v, kf_idx + 0, inv_stddev * rn.x, weight, raw);
        sum_r_and_m(v, kf_idx + 1, inv_stddev * rn.y, weight, raw);
        sum_r_and_m(v, kf_idx + 2, inv_stddev * rn.z, weight, raw);
        sum_r_and_m(v, kf_idx + 3, inv_stddev * (-rn.y * lu.z + rn.z * lu.y), weight, raw);
        sum_r_and_m(v, kf_idx + 4, inv_stddev * (rn.x * lu.z - rn.z * lu.x), weight, raw);
        sum_r_and_m(v, kf_idx + 5, inv_stddev * (-rn.x * lu.y + rn.y * lu.x), weight, raw);
      }
      if (l->optimize_depth_intrinsics) {                                   /* :258-322 */
        int sparse_px = r.px / dp->sparse_surfel_cell_size;
        int sparse_py = r.py / dp->sparse_surfel_cell_size;
        float cfactor = BSO_AT(float, &dp->cfactor_buffer, sparse_py, sparse_px);
        float raw_inv_depth = 1.0f / (dp->raw_to_float_depth * BSO_AT(uint16_t, &kf->depth, r.py, r.px));
        float exp_inv_depth = bso_expf(-dp->a * raw_inv_depth);
        float corrected_inv_depth = cfactor * exp_inv_depth + raw_inv_depth;
        if (fabsf(corrected_inv_depth) < 1e-4f) vis = 0;                    /* NOTE: stays false for the descriptor part too (:272) */
        float nx = bso_unproj_nx(&c->unproj, r.px);
        float ny = bso_unproj_ny(&c->unproj, r.py);
        float dot = bso_dot(bso_make3(nx, ny, 1), rn);
        float jac_base = inv_stddev * dot * exp_inv_depth / (corrected_inv_depth * corrected_inv_depth);
        const float* m = kf->frame_T_global.m;
        float d_cx_inv = inv_stddev * r.calibrated_depth * bso_dot(r.surfel_normal, bso_make3(m[0], m[1], m[2]));
        float d_cy_inv = inv_stddev * r.calibrated_depth * bso_dot(r.surfel_normal, bso_make3(m[4], m[5], m[6]));
        if (vis) {
          const uint32_t d0 = l->depth_intrinsics_unknown_start_index;
          sum_r_and_m(v, d0 + 2, d_cx_inv, weight, raw);
          sum_r_and_m(v, d0 + 3, d_cy_inv, weight, raw);
          sum_r_and_m(v, d0 + 0, r.px * d_cx_inv, weight, raw);
          sum_r_and_m(v, d0 + 1, r.py * d_cy_inv, weight, raw);
          sum_r_and_m(v, d0 + 4, cfactor * raw_inv_depth * jac_base, weight, raw);
          sum_r_and_m(v, d0 + 5 + sparse_px + sparse_py * dp->cfactor_buffer.width, -jac_base, weight, raw);
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
        const float term1 = -(rn.x * ls.z - rn.z * ls.x);
        const float term2 = -(rn.y * ls.z - rn.z * ls.y);
        const float term3 = 1.f / (ls.z * ls.z);
        float jp1 = -(gx1 * term1 + gy1 * term2) * term3;
        float jp2 = -(gx2 * term1 + gy2 * term2) * term3;
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
        float inv_ls_z = 1.f / ls.z;
        float ls_z_sq = ls.z * ls.z;
        float inv_ls_z_sq = inv_ls_z * inv_ls_z;
        sum_r_and_m2(v, kf_idx + 0, -gx1 * inv_ls_z, w1, r1, -gx2 * inv_ls_z, w2, r2);
        sum_r_and_m2(v, kf_idx + 1, -gy1 * inv_ls_z, w1, r1, -gy2 * inv_ls_z, w2, r2);
        sum_r_and_m2(v, kf_idx + 2, (ls.x * gx1 + ls.y * gy1) * inv_ls_z_sq, w1, r1, (ls.x * gx2 + ls.y * gy2) * inv_ls_z_sq, w2, r2);
        float ls_x_y = ls.x * ls.y;
        const float term1 = ls.y * ls.y + ls_z_sq;
        sum_r_and_m2(v, kf_idx + 3, (term1 * gy1 + ls_x_y * gx1) * inv_ls_z_sq, w1, r1, (term1 * gy2 + ls_x_y * gx2) * inv_ls_z_sq, w2, r2);
        const float term2 = ls.x * ls.x + ls_z_sq;
        sum_r_and_m2(v, kf_idx + 4, -(term2 * gx1 + ls_x_y * gy1) * inv_ls_z_sq, w1, r1, -(term2 * gx2 + ls_x_y * gy2) * inv_ls_z_sq, w2, r2);
        sum_r_and_m2(v, kf_idx + 5, -(ls.x * gy1 - ls.y * gx1) * inv_ls_z, w1, r1, -(ls.x * gy2 - ls.y * gx2) * inv_ls_z, w2, r2);
      }
      if (l->optimize_color_intrinsics) {                                   /* :462-509 */
        const float g_x_1 = gx1 / color_camera->fx, g_y_1 = gy1 / color_camera->fy;
        const float g_x_2 = gx2 / color_camera->fx, g_y_2 = gy2 / color_camera->fy;
        const uint32_t c0 = l->color_intrinsics_unknown_start_index;
        sum_r_and_m2(v, c0 + 0, g_x_1 * bso_unproj_nx(&c->unproj, r.px), w1, r1, g_x_2 * bso_unproj_nx(&c->unproj, r.px), w2, r2);
        sum_r_and_m2(v, c0 + 1, g_y_1 * bso_unproj_ny(&c->unproj, r.py), w1, r1, g_y_2 * bso_unproj_ny(&c->unproj, r.py), w2, r2);
        sum_r_and_m2(v, c0 + 2, g_x_1, w1, r1, g_x_2, w2, r2);
        sum_r_and_m2(v, c0 + 3, g_y_1, w1, r1, g_y_2, w2, r2);
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
      float raw = inv_stddev * bso_dot(rn, bso_sub(lu, r.local_position));
      const float weight = bso_depth_weight(raw);
      float sum = 0;
      float geometry_jacobian = 0;
      float pose_jacobian[6] = {0, 0, 0, 0, 0, 0};
      float dgi[5] = {0, 0, 0, 0, 0};
      float cfactor_entry_jacobian = 0;
      if (l->optimize_geometry) {
        geometry_jacobian = -inv_stddev;
        sum += geometry_jacobian * v->p[l->surfel_unknown_start_index + per_surfel * i + 0];
      }
      if (optimize_poses) {
        pose_jacobian[0] = inv_stddev * rn.x;                          sum += pose_jacobian[0] * v->p[kf_idx + 0];
        pose_jacobian[1] = inv_stddev * rn.y;                          sum += pose_jacobian[1] * v->p[kf_idx + 1];
        pose_jacobian[2] = inv_stddev * rn.z;                          sum += pose_jacobian[2] * v->p[kf_idx + 2];
        pose_jacobian[3] = inv_stddev * (-rn.y * lu.z + rn.z * lu.y);  sum += pose_jacobian[3] * v->p[kf_idx + 3];
        pose_jacobian[4] = inv_stddev * (rn.x * lu.z - rn.z * lu.x);   sum += pose_jacobian[4] * v->p[kf_idx + 4];
        pose_jacobian[5] = inv_stddev * (-rn.x * lu.y + rn.y * lu.x);  sum += pose_jacobian[5] * v->p[kf_idx + 5];
      }
      int djv = 0;
      uint32_t cfactor_entry_index = 0;
      if (l->optimize_depth_intrinsics) {
        int sparse_px = r.px / dp->sparse_surfel_cell_size;
        int sparse_py = r.py / dp->sparse_surfel_cell_size;
        float cfactor = BSO_AT(float, &dp->cfactor_buffer, sparse_py, sparse_px);
        float raw_inv_depth = 1.0f / (dp->raw_to_float_depth * BSO_AT(uint16_t, &kf->depth, r.py, r.px));
        float exp_inv_depth = bso_expf(-dp->a * raw_inv_depth);
        float corrected_inv_depth = cfactor * exp_inv_depth + raw_inv_depth;
        djv = !(fabsf(corrected_inv_depth) < 1e-4f);
        if (djv) {
          const uint32_t d0 = l->depth_intrinsics_unknown_start_index;
          float nx = bso_unproj_nx(&c->unproj, r.px);
          float ny = bso_unproj_ny(&c->unproj, r.py);
          float dot = bso_dot(bso_make3(nx, ny, 1), rn);
          float jac_base = inv_stddev * dot * exp_inv_depth / (corrected_inv_depth * corrected_inv_depth);
          const float* m = kf->frame_T_global.m;
          dgi[2] = inv_stddev * r.calibrated_depth * bso_dot(r.surfel_normal, bso_make3(m[0], m[1], m[2]));
          sum += dgi[2] * v->p[d0 + 2];
          dgi[3] = inv_stddev * r.calibrated_depth * bso_dot(r.surfel_normal, bso_make3(m[4], m[5], m[6]));
          sum += dgi[3] * v->p[d0 + 3];
          dgi[0] = r.px * dgi[2];
          sum += dgi[0] * v->p[d0 + 0];
          dgi[1] = r.py * dgi[3];
          sum += dgi[1] * v->p[d0 + 1];
          dgi[4] = cfactor * raw_inv_depth * jac_base;
          sum += dgi[4] * v->p[d0 + 4];
          cfactor_entry_index = d0 + 5 + sparse_px + sparse_py * dp->cfactor_buffer.width;
          cfactor_entry_jacobian = -jac_base;
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
        const float term1 = -(rn.x * ls.z - rn.z * ls.x);
        const float term2 = -(rn.y * ls.z - rn.z * ls.y);
        const float term3 = 1.f / (ls.z * ls.z);
        gj1 = -(gx1 * term1 + gy1 * term2) * term3;
        gj2 = -(gx2 * term1 + gy2 * term2) * term3;
        float p = v->p[s0 + 0];
        sum_1 += gj1 * p;
        sum_2 += gj2 * p;
        p = v->p[s0 + 1];
        sum_1 += -1.f * p;
        p = v->p[s0 + 2];
        sum_2 += -1.f * p;
      }
      if (optimize_poses) {
        float inv_ls_z = 1.f / ls.z;
        float ls_z_sq = ls.z * ls.z;
        float inv_ls_z_sq = inv_ls_z * inv_ls_z;
        float p = v->p[kf_idx + 0];
        pj1[0] = -gx1 * inv_ls_z; sum_1 += pj1[0] * p;
        pj2[0] = -gx2 * inv_ls_z; sum_2 += pj2[0] * p;
        p = v->p[kf_idx + 1];
        pj1[1] = -gy1 * inv_ls_z; sum_1 += pj1[1] * p;
        pj2[1] = -gy2 * inv_ls_z; sum_2 += pj2[1] * p;
        p = v->p[kf_idx + 2];
        pj1[2] = (ls.x * gx1 + ls.y * gy1) * inv_ls_z_sq; sum_1 += pj1[2] * p;
        pj2[2] = (ls.x * gx2 + ls.y * gy2) * inv_ls_z_sq; sum_2 += pj2[2] * p;
        float ls_x_y = ls.x * ls.y;
        p = v->p[kf_idx + 3];
        const float term1 = ls.y * ls.y + ls_z_sq;
        pj1[3] = (term1 * gy1 + ls_x_y * gx1) * inv_ls_z_sq; sum_1 += pj1[3] * p;
        pj2[3] = (term1 * gy2 + ls_x_y * gx2) * inv_ls_z_sq; sum_2 += pj2[3] * p;
        p = v->p[kf_idx + 4];
        const float term2 = ls.x * ls.x + ls_z_sq;
        pj1[4] = -(term2 * gx1 + ls_x_y * gy1) * inv_ls_z_sq; sum_1 += pj1[4] * p;
        pj2[4] = -(term2 * gx2 + ls_x_y * gy2) * inv_ls_z_sq; sum_2 += pj2[4] * p;
        p = v->p[kf_idx + 5];
        pj1[5] = -(ls.x * gy1 - ls.y * gx1) * inv_ls_z; sum_1 += pj1[5] * p;
        pj2[5] = -(ls.x * gy2 - ls.y * gx2) * inv_ls_z; sum_2 += pj2[5] * p;
      }
      if (l->optimize_color_intrinsics) {
        const float g_x_1 = gx1 / color_camera->fx, g_y_1 = gy1 / color_camera->fy;
        const float g_x_2 = gx2 / color_camera->fx, g_y_2 = gy2 / color_camera->fy;
        const uint32_t c0 = l->color_intrinsics_unknown_start_index;
        float p = v->p[c0 + 0];
        cj1[0] = g_x_1 * bso_unproj_nx(&c->unproj, r.px); sum_1 += cj1[0] * p;
        cj2[0] = g_x_2 * bso_unproj_nx(&c->unproj, r.px); sum_2 += cj2[0] * p;
        p = v->p[c0 + 1];
        cj1[1] = g_y_1 * bso_unproj_ny(&c->unproj, r.py); sum_1 += cj1[1] * p;
        cj2[1] = g_y_2 * bso_unproj_ny(&c->unproj, r.py); sum_2 += cj2[1] * p;
        p = v->p[c0 + 2];
        cj1[2] = g_x_1; sum_1 += cj1[2] * p;
        cj2[2] = g_x_2; sum_2 += cj2[2] * p;
        p = v->p[c0 + 3];
        cj1[3] = g_y_1; sum_1 += cj1[3] * p;
        cj2[3] = g_y_2; sum_2 += cj2[3] * p;
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
