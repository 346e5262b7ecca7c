/*
 * bso_intrinsics.c -- ORACLE (test infrastructure only; see bslam_oracle.h).
 *
 * Serial restatement of OptimizeIntrinsicsCUDA (BS/kernel_opt_intrinsics.cc:38-283) and its three
 * kernels (BS/kernel_opt_intrinsics.cu:46-448); BS/ = /root/reference/applications/badslam/src/badslam/.
 * Block reductions + atomics become fp32 additions in (keyframe, surfel) / cell order.
 */
#include <math.h>
#include <stdlib.h>

#include "bslam_oracle.h"
#include "bso_math.h"

#define K_A_ROWS 5

/* Summation of the GLOBAL blocks (A, b1, colour H, b): 0 (default) = fp32 additions in (keyframe, surfel) order, the
 * order a serialised run of the reference's atomics would give; 1 = the same fp32 terms added in float64 and rounded once
 * (the "truth" a tree- or atomics-ordered fp32 sum is compared with: a serial fp32 sum of ~4e6 terms is itself only good to
 * ~1e-3 relative, and the 4x4 / 5x5 solves amplify that). */
static int g_sum64 = 0;
void bso_set_intrinsics_sum64(int enable) { g_sum64 = enable != 0; }

typedef struct { float f[15]; double d[15]; } acc15;
static void acc_add(acc15* a, int i, float v) { a->f[i] += v; a->d[i] += (double)v; }
static float acc_get(const acc15* a, int i) { return g_sum64 ? (float)a->d[i] : a->f[i]; }
static double acc_getd(const acc15* a, int i) { return a->d[i]; }

/* AccumulateGaussNewtonHAndB<size> BS/gauss_newton.cuh:47-95 for a generic size */
static void accumulate_n(int n, float raw, float w, const float* J, acc15* H, acc15* b) {
  int idx = 0;
  for (int row = 0; row < n; ++row)
    for (int col = row; col < n; ++col) { acc_add(H, idx, w * J[row] * J[col]); ++idx; }
  const float wr = w * raw;
  for (int i = 0; i < n; ++i) acc_add(b, i, wr * J[i]);
}

void bso_optimize_intrinsics(
    int optimize_depth_intrinsics, int optimize_color_intrinsics,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    bslam_camera4f* out_color_camera, bslam_camera4f* out_depth_camera, float* a, int tex_mode) {
  if (surfels_size == 0) return;
  const bso_unprojector unproj = bso_make_unprojector(depth_camera);
  const bso_depth_to_color d2c = bso_make_depth_to_color(depth_camera, color_camera);
  const int cw = dp->cfactor_buffer.width;
  const int cells = ((depth_camera->width - 1) / dp->sparse_surfel_cell_size + 1) * ((depth_camera->height - 1) / dp->sparse_surfel_cell_size + 1);
  acc15 accA = {{0}, {0}}, accb1 = {{0}, {0}}, acc_color_H = {{0}, {0}}, acc_color_b = {{0}, {0}};
  float* B = (float*)calloc((size_t)K_A_ROWS * cells, sizeof(float));
  float* D = (float*)calloc((size_t)cells, sizeof(float));
  float* b2 = (float*)calloc((size_t)cells, sizeof(float));
  uint32_t* obs = (uint32_t*)calloc((size_t)cells, sizeof(uint32_t));

  for (int k = 0; k < keyframe_count; ++k) {                       /* :80-108 */
    const bslam_keyframe_view* kf = &keyframes[k];
    for (uint32_t i = 0; i < surfels_size; ++i) {                   /* kernel :46-217 */
      float dj[K_A_ROWS + 1] = {0, 0, 0, 0, 0, 0};
      float raw_depth = 0;
      float j1[4] = {0, 0, 0, 0}, j2[4] = {0, 0, 0, 0};
      float r1 = 0, r2 = 0;
      int cell = -1;
      bso_projection r;
      if (bso_surfel_projects_to_associated_pixel(i, surfels_size, surfels, &kf->depth, &kf->normals, dp, depth_camera, &unproj, &kf->frame_T_global, &r)) {
        const float nx = bso_unproj_nx(&unproj, r.px), ny = bso_unproj_ny(&unproj, r.py);
        if (optimize_depth_intrinsics) {
          const int sparse_px = r.px / dp->sparse_surfel_cell_size, sparse_py = r.py / dp->sparse_surfel_cell_size;
          const float cfactor = BSO_AT(float, &dp->cfactor_buffer, sparse_py, sparse_px);
          const float raw_inv_depth = 1.0f / (dp->raw_to_float_depth * BSO_AT(uint16_t, &kf->depth, r.py, r.px));
          const bso_f3 ln = bso_rotate34(&kf->frame_T_global, r.surfel_normal);
          const float inv_stddev = bso_depth_inv_stddev(nx, ny, r.calibrated_depth, ln, dp->baseline_fx);
          float dj_all[6];
          const float corrected_inv_depth = bso_jac_depth_intrinsics(inv_stddev, r.calibrated_depth, r.px, r.py, nx, ny, r.surfel_normal,
                                                                     kf->frame_T_global.m, ln, cfactor, dp->a, raw_inv_depth, dj_all);
          if (fabsf(corrected_inv_depth) > 1e-4f) {
            for (int q = 0; q < 6; ++q) dj[q] = dj_all[q];
            const bso_f3 lu = bso_make3(r.calibrated_depth * nx, r.calibrated_depth * ny, r.calibrated_depth);
            raw_depth = bso_depth_residual(inv_stddev, ln, lu, r.local_position);
            cell = sparse_px + sparse_py * cw;
          }
        }
        if (optimize_color_intrinsics) {
          bso_f2 color_pxy;
          if (bso_depth_to_color_pxy(r.pxy, &d2c, &color_pxy)) {
            bso_f2 t1, t2;
            bso_tangent_projections(r.global_position, r.surfel_normal, BSO_AT(float, surfels, BSLAM_SURFEL_RADIUS_SQUARED, i),
                                    &kf->frame_T_global, color_camera->fx, color_camera->fy, color_camera->cx, color_camera->cy, &t1, &t2);
            float gx1, gy1, gx2, gy2;
            bso_descriptor_jacobian_wrt_projected_position(&kf->color, color_pxy, t1, t2, &gx1, &gy1, &gx2, &gy2);
            bso_jac_desc_color_intrinsics(gx1, gy1, nx, ny, j1);
            bso_jac_desc_color_intrinsics(gx2, gy2, nx, ny, j2);
            bso_raw_descriptor_residual(&kf->color, tex_mode, color_pxy, t1, t2, BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR1, i),
                                        BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR2, i), &r1, &r2);
          }
        }
      }
      if (optimize_depth_intrinsics && cell >= 0) {                 /* :170-196 */
        const float w = bso_depth_weight(raw_depth);
        accumulate_n(K_A_ROWS, raw_depth, w, dj, &accA, &accb1);
        for (int q = 0; q < K_A_ROWS; ++q) B[(size_t)q * cells + cell] += w * dj[q] * dj[K_A_ROWS];
        D[cell] += w * dj[K_A_ROWS] * dj[K_A_ROWS];
        b2[cell] += w * raw_depth * dj[K_A_ROWS];
        obs[cell] += 1;
      }
      if (optimize_color_intrinsics) {                              /* :198-216: "valid" = residual != 0 */
        if (r1 != 0) accumulate_n(4, r1, bso_desc_weight(r1), j1, &acc_color_H, &acc_color_b);
        if (r2 != 0) accumulate_n(4, r2, bso_desc_weight(r2), j2, &acc_color_H, &acc_color_b);
      }
    }
  }

  float A[15], b1[K_A_ROWS], color_H[10], color_b[4];
  for (int i = 0; i < 15; ++i) A[i] = acc_get(&accA, i);
  for (int i = 0; i < K_A_ROWS; ++i) b1[i] = acc_get(&accb1, i);
  for (int i = 0; i < 10; ++i) color_H[i] = acc_get(&acc_color_H, i);
  for (int i = 0; i < 4; ++i) color_b[i] = acc_get(&acc_color_b, i);
  if (optimize_depth_intrinsics) {
    /* ComputeIntrinsicsIntermediateMatricesCUDAKernel :265-340 */
    double A64[15] = {0}, b164[K_A_ROWS] = {0};   /* sum64 mode: the Schur corrections summed in float64, applied once */
    for (int p = 0; p < cells; ++p) {
      const float D_inverse = 1.0f / D[p];
      if (!(D_inverse < 1e12f)) { D[p] = NAN; continue; }
      const float D_inv_b2 = D_inverse * b2[p];
      D[p] = D_inv_b2;
      int idx = 0;
      for (int row = 0; row < K_A_ROWS; ++row)
        for (int col = row; col < K_A_ROWS; ++col) {
          const float v = -1.f * (B[(size_t)row * cells + p] * D_inverse * B[(size_t)col * cells + p]);
          if (g_sum64) A64[idx] += (double)v; else A[idx] += v;
          ++idx;
        }
      for (int row = 0; row < K_A_ROWS; ++row) {
        const float v = -1.f * (B[(size_t)row * cells + p] * D_inv_b2);
        if (g_sum64) b164[row] += (double)v; else b1[row] += v;
      }
      for (int row = 0; row < K_A_ROWS; ++row) B[(size_t)row * cells + p] = D_inverse * B[(size_t)row * cells + p];
    }
    if (g_sum64) {
      for (int i = 0; i < 15; ++i) A[i] = (float)(acc_getd(&accA, i) + A64[i]);
      for (int i = 0; i < K_A_ROWS; ++i) b1[i] = (float)(acc_getd(&accb1, i) + b164[i]);
    }
    /* host solve BS/kernel_opt_intrinsics.cc:129-186 */
    const float kAPriorWeight = 10;
    A[14] += kAPriorWeight * kAPriorWeight;            /* cpu_matrix(4, 4) */
    b1[4] += kAPriorWeight * kAPriorWeight * (*a);
    float x1[K_A_ROWS];
    bso_solve_ldlt_upper(K_A_ROWS, A, b1, x1);
    const float new_fx = 1.0f / (unproj.fx_inv - x1[0]);
    const float new_fy = 1.0f / (unproj.fy_inv - x1[1]);
    const float new_cx = -(new_fx * (unproj.cx_inv - x1[2])) + 0.5f;
    const float new_cy = -(new_fy * (unproj.cy_inv - x1[3])) + 0.5f;
    out_depth_camera->fx = new_fx; out_depth_camera->fy = new_fy; out_depth_camera->cx = new_cx; out_depth_camera->cy = new_cy;
    out_depth_camera->width = depth_camera->width; out_depth_camera->height = depth_camera->height;
    *a -= x1[4];
    /* SolveForPixelIntrinsicsUpdateCUDAKernel :374-420 */
    for (int p = 0; p < cells; ++p) {
      float offset = D[p];
      if (isnan(offset)) offset = 0;
      else for (int row = 0; row < K_A_ROWS; ++row) offset -= B[(size_t)row * cells + p] * x1[row];
      const int y = p / cw, x = p - y * cw;
      float cfactor = BSO_AT(float, &dp->cfactor_buffer, y, x) - offset;
      if (obs[p] == 0) cfactor = 0;
      BSO_AT(float, &dp->cfactor_buffer, y, x) = cfactor;
    }
  }
  if (optimize_color_intrinsics) {                                    /* :251-280 */
    float x[4];
    bso_solve_ldlt_upper(4, color_H, color_b, x);
    out_color_camera->fx = color_camera->fx - x[0]; out_color_camera->fy = color_camera->fy - x[1];
    out_color_camera->cx = color_camera->cx - x[2]; out_color_camera->cy = color_camera->cy - x[3];
    out_color_camera->width = color_camera->width; out_color_camera->height = color_camera->height;
  }
  free(B); free(D); free(b2); free(obs);
}
