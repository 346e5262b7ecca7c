/*
 * bslam_oracle.c -- ORACLE (test infrastructure only; see bslam_oracle.h).
 *
 * Serial CPU restatement of the reference's BA hot path.  Each function cites the
 * reference file:line it follows (BS/ = /root/reference/applications/badslam/src/badslam/).
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include "bslam_oracle.h"

#include <float.h>
#include <stdlib.h>

#include "bso_math.h"

/* ========================================================================== */
/* Sophus::SE3f restated (libvis/third_party/sophus/sophus/so3.hpp, se3.hpp)   */
/* ========================================================================== */

#define BSO_SOPHUS_EPS 1e-5f /* Constants<float>::epsilon(), sophus/common.hpp:146-148 */

typedef struct { float x, y, z, w; } bso_quat;

static bso_quat q_load(const bslam_se3f* T) { bso_quat q = {T->q[0], T->q[1], T->q[2], T->q[3]}; return q; }
static void q_store(bslam_se3f* T, bso_quat q) { T->q[0] = q.x; T->q[1] = q.y; T->q[2] = q.z; T->q[3] = q.w; }

/* Eigen::Quaternion product (Eigen/src/Geometry/Quaternion.h, quat_product) */
static bso_quat q_mul(bso_quat a, bso_quat b) {
  bso_quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

/* Eigen QuaternionBase::_transformVector: uv = vec x v; uv += uv; v + w*uv + vec x uv */
static bso_f3 q_rotate(bso_quat q, bso_f3 v) {
  bso_f3 qv = bso_make3(q.x, q.y, q.z);
  bso_f3 uv = bso_make3(qv.y * v.z - qv.z * v.y, qv.z * v.x - qv.x * v.z, qv.x * v.y - qv.y * v.x);
  uv = bso_add(uv, uv);
  bso_f3 c = bso_make3(qv.y * uv.z - qv.z * uv.y, qv.z * uv.x - qv.x * uv.z, qv.x * uv.y - qv.y * uv.x);
  return bso_make3(v.x + q.w * uv.x + c.x, v.y + q.w * uv.y + c.y, v.z + q.w * uv.z + c.z);
}

/* Eigen QuaternionBase::toRotationMatrix */
static void q_to_matrix(bso_quat q, float R[9]) {
  const float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
  R[3] = txy + twz;          R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = 1.0f - (txx + tyy);
}

void bso_se3_identity(bslam_se3f* T) {
  T->q[0] = T->q[1] = T->q[2] = 0.f; T->q[3] = 1.f;
  T->t[0] = T->t[1] = T->t[2] = 0.f;
}

/* SO3::expAndTheta so3.hpp:282-318 */
static bso_quat so3_exp(bso_f3 omega, float* theta) {
  float theta_sq = bso_sqlen(omega);
  *theta = sqrtf(theta_sq);
  float half_theta = 0.5f * (*theta);
  float imag_factor, real_factor;
  if ((*theta) < BSO_SOPHUS_EPS) {
    float theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * theta_po4;
    real_factor = 1.f - 0.5f * theta_sq + (float)(1.0 / 384.0) * theta_po4;
  } else {
    float sin_half_theta = sinf(half_theta);
    imag_factor = sin_half_theta / (*theta);
    real_factor = cosf(half_theta);
  }
  bso_quat q = {imag_factor * omega.x, imag_factor * omega.y, imag_factor * omega.z, real_factor};
  return q;
}

static void mat3_mul(const float A[9], const float B[9], float C[9]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      C[3 * r + c] = A[3 * r + 0] * B[0 + c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

/* SE3::exp se3.hpp:293-313 */
void bso_se3_exp(const float a[6], bslam_se3f* out) {
  bso_f3 omega = bso_make3(a[3], a[4], a[5]);
  float theta;
  bso_quat q = so3_exp(omega, &theta);
  /* SO3::hat */
  float Omega[9] = {0.f, -omega.z, omega.y, omega.z, 0.f, -omega.x, -omega.y, omega.x, 0.f};
  float Omega_sq[9];
  mat3_mul(Omega, Omega, Omega_sq);
  float V[9];
  if (theta < BSO_SOPHUS_EPS) {
    q_to_matrix(q, V);
  } else {
    float theta_sq = theta * theta;
    float c1 = (1.f - cosf(theta)) / (theta_sq);
    float c2 = (theta - sinf(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; ++i) {
      float id = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
      V[i] = (id + c1 * Omega[i]) + c2 * Omega_sq[i];
    }
  }
  q_store(out, q);
  out->t[0] = V[0] * a[0] + V[1] * a[1] + V[2] * a[2];
  out->t[1] = V[3] * a[0] + V[4] * a[1] + V[5] * a[2];
  out->t[2] = V[6] * a[0] + V[7] * a[1] + V[8] * a[2];
}

/* SO3::logAndTheta so3.hpp:421-466 */
static bso_f3 so3_log(bso_quat q, float* theta) {
  float squared_n = q.x * q.x + q.y * q.y + q.z * q.z;
  float n = sqrtf(squared_n);
  float w = q.w;
  float two_atan_nbyw_by_n;
  if (n < BSO_SOPHUS_EPS) {
    float squared_w = w * w;
    two_atan_nbyw_by_n = 2.f / w - 2.f * (squared_n) / (w * squared_w);
  } else {
    if (fabsf(w) < BSO_SOPHUS_EPS) {
      if (w > 0.f) two_atan_nbyw_by_n = (float)M_PI / n;
      else two_atan_nbyw_by_n = -(float)M_PI / n;
    } else {
      two_atan_nbyw_by_n = 2.f * atanf(n / w) / n;
    }
  }
  *theta = two_atan_nbyw_by_n * n;
  return bso_make3(two_atan_nbyw_by_n * q.x, two_atan_nbyw_by_n * q.y, two_atan_nbyw_by_n * q.z);
}

/* SE3::log se3.hpp:435-466 */
void bso_se3_log(const bslam_se3f* T, float out[6]) {
  float theta;
  bso_f3 omega = so3_log(q_load(T), &theta);
  float Omega[9] = {0.f, -omega.z, omega.y, omega.z, 0.f, -omega.x, -omega.y, omega.x, 0.f};
  float Omega_sq[9];
  mat3_mul(Omega, Omega, Omega_sq);
  float V_inv[9];
  float k;
  if (fabsf(theta) < BSO_SOPHUS_EPS) {
    k = (float)(1. / 12.);
  } else {
    float half_theta = 0.5f * theta;
    k = (1.f - theta * cosf(half_theta) / (2.f * sinf(half_theta))) / (theta * theta);
  }
  for (int i = 0; i < 9; ++i) {
    float id = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
    V_inv[i] = (id - 0.5f * Omega[i]) + k * Omega_sq[i];
  }
  out[0] = V_inv[0] * T->t[0] + V_inv[1] * T->t[1] + V_inv[2] * T->t[2];
  out[1] = V_inv[3] * T->t[0] + V_inv[4] * T->t[1] + V_inv[5] * T->t[2];
  out[2] = V_inv[6] * T->t[0] + V_inv[7] * T->t[1] + V_inv[8] * T->t[2];
  out[3] = omega.x; out[4] = omega.y; out[5] = omega.z;
}

/* SE3Base::operator*= : translation() += so3() * other.translation(); so3() *= other.so3();
 * SO3Base::operator*= so3.hpp:215-232 (first-order renormalisation). */
void bso_se3_mul(const bslam_se3f* a, const bslam_se3f* b, bslam_se3f* out) {
  bso_quat qa = q_load(a), qb = q_load(b);
  bso_f3 rt = q_rotate(qa, bso_make3(b->t[0], b->t[1], b->t[2]));
  float t0 = a->t[0] + rt.x, t1 = a->t[1] + rt.y, t2 = a->t[2] + rt.z;
  bso_quat q = q_mul(qa, qb);
  float squared_norm = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  if (squared_norm != 1.0f) {
    float f = 2.0f / (1.0f + squared_norm);
    q.x *= f; q.y *= f; q.z *= f; q.w *= f;
  }
  q_store(out, q);
  out->t[0] = t0; out->t[1] = t1; out->t[2] = t2;
}

/* SE3Base::inverse: invR = so3().inverse() (conjugate); SE3(invR, invR * (translation() * -1)) */
void bso_se3_inverse(const bslam_se3f* T, bslam_se3f* out) {
  bso_quat q = q_load(T);
  bso_quat qi = {-q.x, -q.y, -q.z, q.w};
  bso_f3 t = q_rotate(qi, bso_make3(T->t[0] * -1.f, T->t[1] * -1.f, T->t[2] * -1.f));
  q_store(out, qi);
  out->t[0] = t.x; out->t[1] = t.y; out->t[2] = t.z;
}

void bso_se3_matrix3x4(const bslam_se3f* T, bslam_mat3x4* out) {
  float R[9];
  q_to_matrix(q_load(T), R);
  for (int r = 0; r < 3; ++r) {
    out->m[4 * r + 0] = R[3 * r + 0];
    out->m[4 * r + 1] = R[3 * r + 1];
    out->m[4 * r + 2] = R[3 * r + 2];
    out->m[4 * r + 3] = T->t[r];
  }
}

void bso_se3_rotation(const bslam_se3f* T, bslam_mat3x3* out) { q_to_matrix(q_load(T), out->m); }

/* Keyframe::set_global_T_frame (BS/keyframe.h:160-173): caches frame_T_global (3x4)
 * and global_R_frame = global_T_frame.rotationMatrix(). */
void bso_keyframe_set_pose(bslam_keyframe_view* kf, const bslam_se3f* global_T_frame) {
  bslam_se3f inv;
  bso_se3_inverse(global_T_frame, &inv);
  bso_se3_matrix3x4(&inv, &kf->frame_T_global);
  bso_se3_rotation(global_T_frame, &kf->global_R_frame);
}

/* BS/convergence_analysis.h:45-52 */
int bso_is_scale1_pose_estimation_converged(const float x[6]) {
  const float translation_threshold = 1e-06f;
  const float rotation_threshold = 1e-07f;
  const float s = translation_threshold / rotation_threshold;
  float scaled[6] = {x[0], x[1], x[2], x[3] * s, x[4] * s, x[5] * s};
  /* Eigen squaredNorm of a 6-vector: plain left-to-right sum is what the unrolled
   * redux produces for this size up to association; the comparison is far from ties. */
  float n = 0.f;
  for (int i = 0; i < 6; ++i) n += scaled[i] * scaled[i];
  return n < translation_threshold;
}

/* Eigen 3.3 LDLT (Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked + solve):
 * symmetric pivoting on the largest |diagonal| entry, then x = P^T L^-T D^-1 L^-1 P b
 * with D entries below the smallest normal double treated as zero. */
void bso_solve_ldlt_upper(int n, const float* H_upper, const float* b, float* x) {
  double A[36];
  int idx = 0;
  for (int r = 0; r < n; ++r)
    for (int c = r; c < n; ++c) {
      A[r * n + c] = (double)H_upper[idx];
      A[c * n + r] = (double)H_upper[idx];
      ++idx;
    }
  int perm[6];
  for (int k = 0; k < n; ++k) {
    /* pivot: biggest |diagonal| in the remaining sub-matrix */
    int p = k;
    double biggest = fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i) {
      double v = fabs(A[i * n + i]);
      if (v > biggest) { biggest = v; p = i; }
    }
    perm[k] = p;
    if (p != k) {
      /* symmetric row/column swap of the full matrix */
      for (int c = 0; c < n; ++c) { double t = A[k * n + c]; A[k * n + c] = A[p * n + c]; A[p * n + c] = t; }
      for (int r = 0; r < n; ++r) { double t = A[r * n + k]; A[r * n + k] = A[r * n + p]; A[r * n + p] = t; }
    }
    /* A[k][k] -= sum_j L[k][j]^2 D[j]; column below = (A[i][k] - sum_j L[i][j] D[j] L[k][j]) / D[k]
     * (L stored in the strict lower part, D on the diagonal) */
    double temp[6];
    for (int j = 0; j < k; ++j) temp[j] = A[j * n + j] * A[k * n + j];
    double acc = 0.0;
    for (int j = 0; j < k; ++j) acc += A[k * n + j] * temp[j];
    double dk = A[k * n + k] - acc;
    A[k * n + k] = dk;
    for (int i = k + 1; i < n; ++i) {
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += A[i * n + j] * temp[j];
      double v = A[i * n + k] - s;
      A[i * n + k] = (fabs(dk) > 0.0) ? v / dk : 0.0;
    }
  }
  double y[6];
  for (int i = 0; i < n; ++i) y[i] = (double)b[i];
  for (int k = 0; k < n; ++k) if (perm[k] != k) { double t = y[k]; y[k] = y[perm[k]]; y[perm[k]] = t; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) y[i] -= A[i * n + j] * y[j];
  for (int i = 0; i < n; ++i) {
    double d = A[i * n + i];
    y[i] = (fabs(d) > DBL_MIN) ? y[i] / d : 0.0;
  }
  for (int i = n - 1; i >= 0; --i) for (int j = i + 1; j < n; ++j) y[i] -= A[j * n + i] * y[j];
  for (int k = n - 1; k >= 0; --k) if (perm[k] != k) { double t = y[k]; y[k] = y[perm[k]]; y[perm[k]] = t; }
  for (int i = 0; i < n; ++i) x[i] = (float)y[i];
}

/* ========================================================================== */
/* association                                                                 */
/* ========================================================================== */

void bso_association(const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
                     const bslam_keyframe_view* kf, uint32_t surfels_size, const bslam_buffer2d* surfels,
                     uint32_t* out_pixel) {
  bso_unprojector unproj = bso_make_unprojector(depth_camera);
  for (uint32_t i = 0; i < surfels_size; ++i) {
    bso_projection r;
    if (bso_surfel_projects_to_associated_pixel(i, surfels_size, surfels, &kf->depth, &kf->normals, dp, depth_camera, &unproj,
                                                &kf->frame_T_global, &r)) {
      out_pixel[i] = (uint32_t)(r.py * kf->depth.width + r.px);
    } else {
      out_pixel[i] = 0xffffffffu;
    }
  }
}

/* Evaluation-shape switch of bso_math.h (0: twin of the HIP kernels, 1: literal transcription of the reference). */
int bso_literal_mode = 0;
void bso_set_literal_mode(int mode) { bso_literal_mode = mode != 0; }
int bso_get_literal_mode(void) { return bso_literal_mode; }

/* Decision margins of the association chain for every surfel, evaluated in float64 from the same inputs (diagnosis aid of
 * tests/test_oracle_literal.py: a surfel whose integer output differs between the two evaluation shapes must sit on a
 * threshold).  out[5 * i + ...] = { local z,
 *   distance of (u, v) to the nearest pixel boundary or image border [px],
 *   (threshold - |z - depth|) / threshold of the depth test (BS/surfel_projection_nvcc_only.cuh:88-100),
 *   -cos(angle between viewing ray and surfel normal)            (:103-108),
 *   dot(surfel normal, pixel normal) - cos 40 deg               (:111-121) };
 * entries after the first failed stage are NaN. */
void bso_association_margins(const bslam_camera4f* depth_camera, const bslam_depth_params* dp, const bslam_keyframe_view* kf,
                             uint32_t surfels_size, const bslam_buffer2d* surfels, double* out) {
  const double fx = depth_camera->fx, fy = depth_camera->fy, cx = depth_camera->cx, cy = depth_camera->cy;
  const float* m = kf->frame_T_global.m;
  for (uint32_t i = 0; i < surfels_size; ++i) {
    double* o = out + 5 * (size_t)i;
    for (int k = 0; k < 5; ++k) o[k] = NAN;
    const bso_f3 gpf = bso_surfel_position(surfels, i);
    const double g[3] = {gpf.x, gpf.y, gpf.z};
    double l[3];
    for (int r = 0; r < 3; ++r) l[r] = (double)m[4 * r] * g[0] + (double)m[4 * r + 1] * g[1] + (double)m[4 * r + 2] * g[2] + (double)m[4 * r + 3];
    o[0] = l[2];
    if (!(l[2] > 0)) continue;
    const double u = fx * (l[0] / l[2]) + cx, v = fy * (l[1] / l[2]) + cy;
    const double du = fmin(u - floor(u), ceil(u) - u), dv = fmin(v - floor(v), ceil(v) - v);
    o[1] = fmin(fmin(du, dv), fmin(fmin(u, v), fmin(kf->depth.width - u, kf->depth.height - v)));
    if (u < 0 || v < 0 || u >= kf->depth.width || v >= kf->depth.height) continue;
    const int px = (int)u, py = (int)v;
    const uint16_t measured = BSO_AT(uint16_t, &kf->depth, py, px);
    if (measured & BSLAM_INVALID_DEPTH_BIT) continue;
    const double cf = BSO_AT(float, &dp->cfactor_buffer, py / dp->sparse_surfel_cell_size, px / dp->sparse_surfel_cell_size);
    const double inv_depth = 1.0 / ((double)dp->raw_to_float_depth * measured);
    const double depth = 1.0 / (inv_depth + cf * exp(-(double)dp->a * inv_depth));
    const bso_f3 nf = bso_surfel_normal(surfels, i);
    double n[3];
    for (int r = 0; r < 3; ++r) n[r] = (double)m[4 * r] * nf.x + (double)m[4 * r + 1] * nf.y + (double)m[4 * r + 2] * nf.z;
    const double fx_inv = 1.0 / fx, fy_inv = 1.0 / fy;
    const double nx = fx_inv * px - (cx - 0.5) * fx_inv, ny = fy_inv * py - (cy - 0.5) * fy_inv;
    const double thr = 10.0 * (0.1 * fabs(n[0] * nx + n[1] * ny + n[2]) * depth * depth) / (double)dp->baseline_fx;
    o[2] = (thr - fabs(l[2] - depth)) / thr;
    if (fabs(l[2] - depth) > thr) continue;
    o[3] = -(l[0] * n[0] + l[1] * n[1] + l[2] * n[2]) / sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
    if (o[3] < 0) continue;
    const uint16_t pn = BSO_AT(uint16_t, &kf->normals, py, px);
    const double pnx = (int8_t)(pn & 0xff) / 127.0, pny = (int8_t)(pn >> 8) / 127.0;
    const double pz2 = 1.0 - pnx * pnx - pny * pny;
    const double pnz = -sqrt(pz2 > 0 ? pz2 : 0);
    o[4] = (n[0] * pnx + n[1] * pny + n[2] * pnz) - (double)BSO_COS_NORMAL_COMPAT;
  }
}

/* Point-wise evaluation of the Jacobian formulas of bso_math.h (the ones every loop of this oracle calls) for
 * tests/test_jacobian_golden.py.  Layouts (floats per point, in -> out); the HIP entry bslam_debug_jacobians uses the same:
 *   kind 0 depth / pose          [inv_stddev, n_local(3), lu(3), ls(3)]                                   -> [raw residual, J(6)]
 *   kind 1 depth / position      [inv_stddev]                                                              -> [j]
 *   kind 2 depth / intrinsics    [inv_stddev, calibrated depth, px, py, nx, ny, n_global(3), frame_T_global row 0 (3), row 1 (3),
 *                                 n_local(3), cfactor, a, raw_inv_depth]                                   -> [corrected_inv_depth, dj(6)]
 *   kind 3 descriptor / pose     [tl, tr, bl, br, tx, ty, fx, fy, ls(3)]                                   -> [bilinear value, gx fx, gy fy, J(6)]
 *   kind 4 descriptor / position [tl, tr, bl, br, tx, ty, fx, fy, rn(3), ls(3)]                           -> [j]
 *   kind 5 descriptor / colour intrinsics [tl, tr, bl, br, tx, ty, nx, ny]                                 -> [j(4)]
 * (tl .. br = the 2x2 texel footprint, tx, ty = fractional offsets of the sample inside it). */
static void probe_gradient(const float* a, float* dx, float* dy) {   /* the gradient form of the active evaluation shape */
  if (bso_literal_mode) bso_bilinear_gradient(a[0], a[1], a[2], a[3], a[4], a[5], dx, dy);
  else bso_bilinear_gradient_bytes(a[0], a[1], a[2], a[3], a[4], a[5], dx, dy);
}

int bso_jacobian_probe(int kind, int count, const float* in, float* out) {
  static const int kIn[6] = {10, 1, 21, 11, 14, 8}, kOut[6] = {7, 1, 7, 9, 1, 4};
  if (kind < 0 || kind > 5) return -1;
  for (int i = 0; i < count; ++i) {
    const float* a = in + (size_t)i * kIn[kind];
    float* o = out + (size_t)i * kOut[kind];
    switch (kind) {
      case 0: {
        const bso_f3 nl = bso_make3(a[1], a[2], a[3]), lu = bso_make3(a[4], a[5], a[6]), ls = bso_make3(a[7], a[8], a[9]);
        o[0] = bso_depth_residual(a[0], nl, lu, ls);
        bso_jac_depth_pose(a[0], nl, lu, o + 1);
        break;
      }
      case 1: o[0] = bso_jac_depth_position(a[0]); break;
      case 2: {
        const float m[12] = {a[9], a[10], a[11], 0, a[12], a[13], a[14], 0, 0, 0, 0, 0};
        o[0] = bso_jac_depth_intrinsics(a[0], a[1], (int)a[2], (int)a[3], a[4], a[5], bso_make3(a[6], a[7], a[8]), m, bso_make3(a[15], a[16], a[17]),
                                        a[18], a[19], a[20], o + 1);
        break;
      }
      case 3: {
        float dx, dy;
        probe_gradient(a, &dx, &dy);
        const float w00 = (1.0f - a[4]) * (1.0f - a[5]), w10 = a[4] * (1.0f - a[5]), w01 = (1.0f - a[4]) * a[5], w11 = a[4] * a[5];
        o[0] = bso_literal_mode ? ((w00 * a[0] + w10 * a[1]) + w01 * a[2]) + w11 * a[3]       /* the filter of bso_tex_w */
                                : bso_bilinear_bytes(a[0], a[1], a[2], a[3], a[4], a[5]);
        o[1] = dx * a[6];
        o[2] = dy * a[7];
        bso_jac_desc_pose(o[1], o[2], bso_make3(a[8], a[9], a[10]), o + 3);
        break;
      }
      case 4: {
        float dx, dy;
        probe_gradient(a, &dx, &dy);
        o[0] = bso_jac_desc_position(dx, dy, a[6], a[7], bso_make3(a[8], a[9], a[10]), bso_make3(a[11], a[12], a[13]));
        break;
      }
      case 5: {
        float dx, dy;
        probe_gradient(a, &dx, &dy);
        bso_jac_desc_color_intrinsics(dx, dy, a[6], a[7], o);
        break;
      }
    }
  }
  return 0;
}

/* ========================================================================== */
/* pose optimisation                                                           */
/* ========================================================================== */

/* AccumulateGaussNewtonHAndB<6> BS/gauss_newton.cuh:47-95 (serial sum, index order) */
static void accumulate_h_and_b(float raw_residual, float weight, const float* J, float* H, float* b, double* H64, double* b64) {
  int idx = 0;
  for (int row = 0; row < 6; ++row)
    for (int col = row; col < 6; ++col) {
      const float v = weight * J[row] * J[col];
      H[idx] += v;
      if (H64) H64[idx] += (double)v;
      ++idx;
    }
  const float weighted_raw_residual = weight * raw_residual;
  for (int i = 0; i < 6; ++i) {
    const float v = weighted_raw_residual * J[i];
    b[i] += v;
    if (b64) b64[i] += (double)v;
  }
}

void bso_accumulate_pose_estimation_coeffs(
    int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp,
    const bslam_buffer2d* depth_buffer, const bslam_buffer2d* normals_buffer, const bslam_buffer2d* color_buffer,
    const bslam_mat3x4* frame_T_global, uint32_t surfels_size, const bslam_buffer2d* surfels,
    int tex_mode, uint32_t* residual_count, float* residual_sum,
    float* H, float* b, double* H64, double* b64, float* per_surfel) {
  for (int i = 0; i < 21; ++i) { H[i] = 0.f; if (H64) H64[i] = 0.0; }
  for (int i = 0; i < 6; ++i) { b[i] = 0.f; if (b64) b64[i] = 0.0; }
  uint32_t count = 0;
  float cost = 0.f;
  bso_unprojector unproj = bso_make_unprojector(depth_camera);           /* CreatePixelCenterUnprojector(depth_camera) */
  bso_depth_to_color d2c = bso_make_depth_to_color(depth_camera, color_camera);
  /* CreatePixelCenterProjector(color_camera): fx, fy, cx-0.5, cy-0.5 -- only fx, fy are used */
  const float color_center_fx = color_camera->fx, color_center_fy = color_camera->fy;

  for (uint32_t i = 0; i < surfels_size; ++i) {
    float* ps = per_surfel ? per_surfel + 8 * (size_t)i : NULL;
    if (ps) for (int k = 0; k < 8; ++k) ps[k] = 0.f;
    bso_projection r;
    int visible = bso_surfel_projects_to_associated_pixel(i, surfels_size, surfels, depth_buffer, normals_buffer, dp, depth_camera,
                                                          &unproj, frame_T_global, &r);
    if (!visible) continue;
    uint32_t flags = 1;
    float J[6];
    float raw_residual;
    /* --- depth residual BS/kernel_opt_pose.cu:283-317 --- */
    if (use_depth_residuals) {
      bso_f3 n_local = bso_rotate34(frame_T_global, r.surfel_normal);
      float inv_stddev = bso_depth_inv_stddev(bso_unproj_nx(&unproj, r.px), bso_unproj_ny(&unproj, r.py), r.calibrated_depth, n_local, dp->baseline_fx);
      /* ComputeRawDepthResidualAndJacobian BS/kernel_opt_pose.cu:45-94 */
      bso_f3 local_unproj = bso_unproject(&unproj, r.px, r.py, r.calibrated_depth);
      raw_residual = bso_depth_residual(inv_stddev, n_local, local_unproj, r.local_position);
      bso_jac_depth_pose(inv_stddev, n_local, local_unproj, J);
      float w = bso_depth_weight(raw_residual);
      accumulate_h_and_b(raw_residual, w, J, H, b, H64, b64);
      count += 1;
      cost += bso_weighted_depth_residual(raw_residual);
      if (ps) { ps[0] = raw_residual; ps[1] = w; }
    }
    /* --- descriptor residual BS/kernel_opt_pose.cu:320-382 --- */
    if (use_descriptor_residuals) {
      bso_f2 color_pxy;
      if (bso_depth_to_color_pxy(r.pxy, &d2c, &color_pxy)) {
        bso_f2 t1, t2;
        bso_tangent_projections(r.global_position, r.surfel_normal, BSO_AT(float, surfels, BSLAM_SURFEL_RADIUS_SQUARED, i),
                                frame_T_global, color_camera->fx, color_camera->fy, color_camera->cx, color_camera->cy, &t1, &t2);
        float r1, r2;
        bso_raw_descriptor_residual(color_buffer, tex_mode, color_pxy, t1, t2,
                                    BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR1, i), BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR2, i), &r1, &r2);
        float gx1, gy1, gx2, gy2;
        bso_descriptor_jacobian_wrt_projected_position(color_buffer, color_pxy, t1, t2, &gx1, &gy1, &gx2, &gy2);
        gx1 *= color_center_fx; gx2 *= color_center_fx;
        gy1 *= color_center_fy; gy2 *= color_center_fy;
        float J1[6], J2[6];
        bso_jac_desc_pose(gx1, gy1, r.local_position, J1);
        bso_jac_desc_pose(gx2, gy2, r.local_position, J2);
        float w1 = bso_desc_weight(r1), w2 = bso_desc_weight(r2);
        accumulate_h_and_b(r1, w1, J1, H, b, H64, b64);
        accumulate_h_and_b(r2, w2, J2, H, b, H64, b64);
        /* debug accumulation counts only the first descriptor residual (quirk Q1) */
        count += 1;
        cost += bso_weighted_desc_residual(r1);
        flags |= 2;
        if (ps) { ps[2] = r1; ps[3] = w1; ps[4] = r2; ps[5] = w2; }
      }
    }
    if (ps) ps[6] = (float)flags;
  }
  if (residual_count) *residual_count = count;
  if (residual_sum) *residual_sum = cost;
}

void bso_estimate_frame_pose(
    int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp,
    const bslam_buffer2d* depth_buffer, const bslam_buffer2d* normals_buffer, const bslam_buffer2d* color_buffer,
    const bslam_se3f* global_T_frame_initial, uint32_t surfels_size, const bslam_buffer2d* surfels,
    int tex_mode, int max_iterations, bslam_se3f* out_global_T_frame, int* iterations_done, int* converged_out) {
  bslam_se3f est = *global_T_frame_initial;
  int converged = 0;
  int iteration;
  for (iteration = 0; iteration < max_iterations; ++iteration) {   /* BS/direct_ba_alternating.cc:133 */
    bslam_se3f frame_T_global;
    bso_se3_inverse(&est, &frame_T_global);
    bslam_mat3x4 M;
    bso_se3_matrix3x4(&frame_T_global, &M);
    float H[21], b[6];
    if (surfels_size == 0) {
      for (int i = 0; i < 21; ++i) H[i] = 0.f;
      for (int i = 0; i < 6; ++i) b[i] = 0.f;
    } else {
      bso_accumulate_pose_estimation_coeffs(use_depth_residuals, use_descriptor_residuals, color_camera, depth_camera, dp,
                                            depth_buffer, normals_buffer, color_buffer, &M, surfels_size, surfels, tex_mode,
                                            NULL, NULL, H, b, NULL, NULL, NULL);
    }
    float x[6];
    bso_solve_ldlt_upper(6, H, b, x);                              /* :206 */
    float neg[6];
    for (int i = 0; i < 6; ++i) neg[i] = -1.f * x[i];              /* :213-214, kDamping = 1 */
    bslam_se3f d, next;
    bso_se3_exp(neg, &d);
    bso_se3_mul(&est, &d, &next);
    est = next;
    converged = bso_is_scale1_pose_estimation_converged(x);        /* :231 */
    if (converged) { ++iteration; break; }
  }
  *out_global_T_frame = est;
  if (iterations_done) *iterations_done = iteration;
  if (converged_out) *converged_out = converged;
}

/* ========================================================================== */
/* activation / geometry                                                       */
/* ========================================================================== */

#define ACTIVE(buf, i) BSO_AT(uint8_t, buf, 0, i)

/* visited (optional): number of (surfel, keyframe) pairs evaluated -- a surfel leaves the walk at its first associated
 * active keyframe (bench accounting). */
void bso_update_surfel_activation_counted(
    const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels, const bslam_buffer2d* active_surfels, uint64_t* visited) {
  if (visited) *visited = 0;
  if (surfels_size == 0) return;
  bso_unprojector unproj = bso_make_unprojector(depth_camera);
  for (uint32_t i = 0; i < surfels_size; ++i) ACTIVE(active_surfels, i) &= (uint8_t)~BSLAM_SURFEL_ACTIVE_FLAG;   /* BS/kernel_surfel_activation.cu:38-46 */
  uint64_t count = 0;
  for (int k = 0; k < keyframe_count; ++k) {
    const bslam_keyframe_view* kf = &keyframes[k];
    if (kf->activation != BSLAM_KF_ACTIVE) continue;              /* BS/kernel_surfel_activation.cc:60 */
    for (uint32_t i = 0; i < surfels_size; ++i) {                 /* BS/kernel_surfel_activation.cu:64-79 */
      if (ACTIVE(active_surfels, i) & BSLAM_SURFEL_ACTIVE_FLAG) continue;
      ++count;
      bso_projection r;
      if (bso_surfel_projects_to_associated_pixel(i, surfels_size, surfels, &kf->depth, &kf->normals, dp, depth_camera, &unproj, &kf->frame_T_global, &r))
        ACTIVE(active_surfels, i) = BSLAM_SURFEL_ACTIVE_FLAG;
    }
  }
  if (visited) *visited = count;
}

void bso_update_surfel_activation(
    const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels, const bslam_buffer2d* active_surfels) {
  bso_update_surfel_activation_counted(depth_camera, dp, keyframe_count, keyframes, surfels_size, surfels, active_surfels, NULL);
}

#define ACC(s, row, i) BSO_AT(float, s, BSLAM_SURFEL_ACCUM0 + (row), i)

/* ---- AssignColorsCUDA (BS/kernel_assign_colors.cc:40-80, BS/kernel_assign_colors.cu:42-125) ---- */
static void bso_texel_rgba(const bslam_buffer2d* color, int ix, int iy, float out[4]) {   /* clamp addressing, u8 -> [0, 1] */
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (ix > color->width - 1) ix = color->width - 1;
  if (iy > color->height - 1) iy = color->height - 1;
  const uint8_t* px = (const uint8_t*)color->address + (size_t)iy * color->pitch + 4 * (size_t)ix;
  for (int ch = 0; ch < 4; ++ch) out[ch] = px[ch] * (1.0f / 255.0f);
}
/* tex2D<float4> with the texture model of bso_tex_w (bso_math.h) */
static void bso_tex_rgba(const bslam_buffer2d* color, float x, float y, int mode, float out[4]) {
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fx = floorf(xb), fy = floorf(yb);
  float a = xb - fx, b = yb - fy;
  if (mode == BSLAM_TEX_FIXED_POINT_1_8) {
    a = floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f);
    b = floorf(b * 256.0f + 0.5f) * (1.0f / 256.0f);
  }
  const int i = (int)fminf(fmaxf(fx, -2.0f), (float)color->width);
  const int j = (int)fminf(fmaxf(fy, -2.0f), (float)color->height);
  float t00[4], t10[4], t01[4], t11[4];
  bso_texel_rgba(color, i, j, t00);
  bso_texel_rgba(color, i + 1, j, t10);
  bso_texel_rgba(color, i, j + 1, t01);
  bso_texel_rgba(color, i + 1, j + 1, t11);
  const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
  for (int ch = 0; ch < 4; ++ch) out[ch] = ((w00 * t00[ch] + w10 * t10[ch]) + w01 * t01[ch]) + w11 * t11[ch];
}
static uint8_t bso_float_to_u8(float v) {   /* cvt.rzi.u8.f32: truncate, saturate */
  int i = bso_f2i(v);
  if (i < 0) i = 0;
  if (i > 255) i = 255;
  return (uint8_t)i;
}

void bso_assign_colors(
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    int keyframe_count, const bslam_keyframe_view* keyframes, int tex_mode,
    uint32_t surfels_size, const bslam_buffer2d* surfels) {
  if (surfels_size == 0) return;                                   /* BS/kernel_assign_colors.cc:49-51 */
  bso_unprojector unproj = bso_make_unprojector(depth_camera);
  bso_depth_to_color d2c = bso_make_depth_to_color(depth_camera, color_camera);
  for (uint32_t i = 0; i < surfels_size; ++i)                      /* ResetSurfelForColorAssignmentKernel .cu:42-55 */
    for (int r = 0; r < 5; ++r) ACC(surfels, r, i) = 0;
  for (int k = 0; k < keyframe_count; ++k) {                       /* every listed keyframe, whatever its activation (.cc:60-73) */
    const bslam_keyframe_view* kf = &keyframes[k];
    for (uint32_t i = 0; i < surfels_size; ++i) {                  /* AccumulateColorObservationsCUDAKernel .cu:74-93 */
      bso_projection r;
      if (!bso_surfel_projects_to_associated_pixel(i, surfels_size, surfels, &kf->depth, &kf->normals, dp, depth_camera, &unproj, &kf->frame_T_global, &r)) continue;
      bso_f2 color_pxy;
      if (!bso_depth_to_color_pxy(r.pxy, &d2c, &color_pxy)) continue;
      float c[4];
      bso_tex_rgba(&kf->color, color_pxy.x, color_pxy.y, tex_mode, c);
      ACC(surfels, 0, i) += 1.f;
      for (int ch = 0; ch < 4; ++ch) ACC(surfels, 1 + ch, i) += c[ch];
    }
  }
  for (uint32_t i = 0; i < surfels_size; ++i) {                    /* AssignColorsCUDAKernel .cu:111-125 */
    const float n = ACC(surfels, 0, i);
    if (!(n > 0)) continue;
    uint8_t* out = (uint8_t*)&BSO_AT(float, surfels, BSLAM_SURFEL_COLOR, i);
    for (int ch = 0; ch < 4; ++ch) out[ch] = bso_float_to_u8(255.f * ACC(surfels, 1 + ch, i) / n + 0.5f);
  }
}

void bso_update_surfel_normals(
    const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels, const bslam_buffer2d* active_surfels) {
  if (surfels_size == 0) return;
  bso_unprojector unproj = bso_make_unprojector(depth_camera);
  for (uint32_t i = 0; i < surfels_size; ++i) {                   /* ResetSurfelAccum0to3 BS/kernel_opt_geometry.cu:82-98 */
    if (!(ACTIVE(active_surfels, i) & BSLAM_SURFEL_ACTIVE_FLAG)) continue;
    for (int r = 0; r < 4; ++r) ACC(surfels, r, i) = 0;
  }
  for (int k = 0; k < keyframe_count; ++k) {
    const bslam_keyframe_view* kf = &keyframes[k];
    if (kf->activation == BSLAM_KF_INACTIVE) continue;            /* BS/kernel_opt_geometry.cc:59 */
    for (uint32_t i = 0; i < surfels_size; ++i) {                 /* BS/kernel_opt_geometry.cu:527-557 */
      if (!(ACTIVE(active_surfels, i) & BSLAM_SURFEL_ACTIVE_FLAG)) continue;
      bso_projection r;
      if (bso_surfel_projects_to_associated_pixel(i, surfels_size, surfels, &kf->depth, &kf->normals, dp, depth_camera, &unproj, &kf->frame_T_global, &r)) {
        bso_f3 local_normal = bso_u16_to_image_space_normal(BSO_AT(uint16_t, &kf->normals, r.py, r.px));
        bso_f3 global_normal = bso_mul33(&kf->global_R_frame, local_normal);
        ACC(surfels, 0, i) += global_normal.x;
        ACC(surfels, 1, i) += global_normal.y;
        ACC(surfels, 2, i) += global_normal.z;
        ACC(surfels, 3, i) += 1.f;
      }
    }
  }
  for (uint32_t i = 0; i < surfels_size; ++i) {                   /* BS/kernel_opt_geometry.cu:577-597 */
    if (!(ACTIVE(active_surfels, i) & BSLAM_SURFEL_ACTIVE_FLAG)) continue;
    float count = ACC(surfels, 3, i);
    if (count >= 1) {
      bso_f3 sum = bso_make3(ACC(surfels, 0, i), ACC(surfels, 1, i), ACC(surfels, 2, i));
      bso_surfel_set_normal(surfels, i, bso_scale(1.f / count, sum));
    }
  }
}

void bso_optimize_geometry_iteration(
    int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels, const bslam_buffer2d* active_surfels,
    int tex_mode) {
  if (surfels_size == 0) return;
  /* --- normals BS/kernel_opt_geometry.cc:104-134 --- */
  bso_update_surfel_normals(depth_camera, dp, keyframe_count, keyframes, surfels_size, surfels, active_surfels);

  bso_unprojector unproj = bso_make_unprojector(depth_camera);
  bso_depth_to_color d2c = bso_make_depth_to_color(depth_camera, color_camera);

  if (!use_descriptor_residuals) {
    /* --- position BS/kernel_opt_geometry.cc:137-169 --- */
    for (uint32_t i = 0; i < surfels_size; ++i) {
      if (!(ACTIVE(active_surfels, i) & BSLAM_SURFEL_ACTIVE_FLAG)) continue;
      ACC(surfels, 0, i) = 0; ACC(surfels, 1, i) = 0;
    }
    for (int k = 0; k < keyframe_count; ++k) {
      const bslam_keyframe_view* kf = &keyframes[k];
      if (kf->activation == BSLAM_KF_INACTIVE) continue;
      for (uint32_t i = 0; i < surfels_size; ++i) {               /* BS/kernel_opt_geometry.cu:417-459 */
        if (!(ACTIVE(active_surfels, i) & BSLAM_SURFEL_ACTIVE_FLAG)) continue;
        bso_projection r;
        if (!bso_surfel_projects_to_associated_pixel(i, surfels_size, surfels, &kf->depth, &kf->normals, dp, depth_camera, &unproj, &kf->frame_T_global, &r)) continue;
        bso_f3 rn = bso_rotate34(&kf->frame_T_global, r.surfel_normal);
        float inv_stddev = bso_depth_inv_stddev(bso_unproj_nx(&unproj, r.px), bso_unproj_ny(&unproj, r.py), r.calibrated_depth, rn, dp->baseline_fx);
        const float depth_jacobian = bso_jac_depth_position(inv_stddev);
        bso_f3 local_unproj = bso_unproject(&unproj, r.px, r.py, r.calibrated_depth);
        float raw = bso_depth_residual(inv_stddev, rn, local_unproj, r.local_position);
        const float w = bso_depth_weight(raw);
        float weighted_jacobian = w * depth_jacobian;
        ACC(surfels, 0, i) += weighted_jacobian * depth_jacobian;
        ACC(surfels, 1, i) += weighted_jacobian * raw;
      }
    }
    for (uint32_t i = 0; i < surfels_size; ++i) {                 /* BS/kernel_opt_geometry.cu:487-507 */
      if (!(ACTIVE(active_surfels, i) & BSLAM_SURFEL_ACTIVE_FLAG)) continue;
      float Hs = ACC(surfels, 0, i);
      const float kEpsilon = 1e-6f;
      if (Hs > kEpsilon) {
        bso_f3 p = bso_surfel_position(surfels, i);
        float t = -1.f * ACC(surfels, 1, i) / Hs;
        bso_f3 n = bso_surfel_normal(surfels, i);
        bso_surfel_set_position(surfels, i, bso_add(p, bso_scale(t, n)));
      }
    }
  } else {
    /* --- position + descriptors BS/kernel_opt_geometry.cc:170-200 --- */
    for (uint32_t i = 0; i < surfels_size; ++i) {
      if (!(ACTIVE(active_surfels, i) & BSLAM_SURFEL_ACTIVE_FLAG)) continue;
      for (int r = 0; r < 9; ++r) ACC(surfels, r, i) = 0;
    }
    for (int k = 0; k < keyframe_count; ++k) {
      const bslam_keyframe_view* kf = &keyframes[k];
      if (kf->activation == BSLAM_KF_INACTIVE) continue;
      for (uint32_t i = 0; i < surfels_size; ++i) {               /* BS/kernel_opt_geometry.cu:118-231 */
        if (!(ACTIVE(active_surfels, i) & BSLAM_SURFEL_ACTIVE_FLAG)) continue;
        bso_projection r;
        if (!bso_surfel_projects_to_associated_pixel(i, surfels_size, surfels, &kf->depth, &kf->normals, dp, depth_camera, &unproj, &kf->frame_T_global, &r)) continue;
        bso_f3 rn = bso_rotate34(&kf->frame_T_global, r.surfel_normal);
        if (use_depth_residuals) {
          float inv_stddev = bso_depth_inv_stddev(bso_unproj_nx(&unproj, r.px), bso_unproj_ny(&unproj, r.py), r.calibrated_depth, rn, dp->baseline_fx);
          const float depth_jacobian = bso_jac_depth_position(inv_stddev);
          bso_f3 local_unproj = bso_unproject(&unproj, r.px, r.py, r.calibrated_depth);
          float raw = bso_depth_residual(inv_stddev, rn, local_unproj, r.local_position);
          const float w = bso_depth_weight(raw);
          ACC(surfels, 0, i) += w * depth_jacobian * depth_jacobian;
          ACC(surfels, 6, i) += w * raw * depth_jacobian;
        }
        bso_f2 color_pxy;
        if (bso_depth_to_color_pxy(r.pxy, &d2c, &color_pxy)) {
          bso_f2 t1, t2;
          bso_tangent_projections(r.global_position, r.surfel_normal, BSO_AT(float, surfels, BSLAM_SURFEL_RADIUS_SQUARED, i),
                                  &kf->frame_T_global, color_camera->fx, color_camera->fy, color_camera->cx, color_camera->cy, &t1, &t2);
          const float d1 = BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR1, i);
          const float d2 = BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR2, i);
          float r1, r2;
          bso_raw_descriptor_residual(&kf->color, tex_mode, color_pxy, t1, t2, d1, d2, &r1, &r2);
          float gx1, gy1, gx2, gy2;
          bso_descriptor_jacobian_wrt_projected_position(&kf->color, color_pxy, t1, t2, &gx1, &gy1, &gx2, &gy2);
          float jp1 = bso_jac_desc_position(gx1, gy1, color_camera->fx, color_camera->fy, rn, r.local_position);
          float jp2 = bso_jac_desc_position(gx2, gy2, color_camera->fx, color_camera->fy, rn, r.local_position);
          const float jd = -1.f;
          const float w1 = bso_desc_weight(r1);
          const float wr1 = w1 * r1;
          const float w2 = bso_desc_weight(r2);
          const float wr2 = w2 * r2;
          ACC(surfels, 0, i) += w1 * jp1 * jp1 + w2 * jp2 * jp2;
          ACC(surfels, 1, i) += w1 * jp1 * jd;
          ACC(surfels, 3, i) += w1 * jd * jd;
          ACC(surfels, 6, i) += wr1 * jp1 + wr2 * jp2;
          ACC(surfels, 7, i) += wr1 * jd;
          ACC(surfels, 2, i) += w2 * jp2 * jd;
          ACC(surfels, 5, i) += w2 * jd * jd;
          ACC(surfels, 8, i) += wr2 * jd;
        }
      }
    }
    for (uint32_t i = 0; i < surfels_size; ++i) {                 /* BS/kernel_opt_geometry.cu:273-361 */
      if (!(ACTIVE(active_surfels, i) & BSLAM_SURFEL_ACTIVE_FLAG)) continue;
      float H_0_0 = ACC(surfels, 0, i), H_0_1 = ACC(surfels, 1, i), H_0_2 = ACC(surfels, 2, i);
      float H_1_1 = ACC(surfels, 3, i), H_1_2 = ACC(surfels, 4, i), H_2_2 = ACC(surfels, 5, i);
      const float kEpsilon = 1e-6f;
      H_0_0 += kEpsilon; H_1_1 += kEpsilon; H_2_2 += kEpsilon;
      H_0_0 = sqrtf(H_0_0);
      H_0_1 = H_0_1 / H_0_0;
      H_1_1 = sqrtf(H_1_1 - H_0_1 * H_0_1);
      H_0_2 = H_0_2 / H_0_0;
      H_1_2 = (H_1_2 - H_0_2 * H_0_1) / H_1_1;
      H_2_2 = sqrtf(H_2_2 - H_0_2 * H_0_2 - H_1_2 * H_1_2);
      const float b0 = ACC(surfels, 6, i), b1 = ACC(surfels, 7, i), b2 = ACC(surfels, 8, i);
      float y0 = b0 / H_0_0;
      float y1 = (b1 - H_0_1 * y0) / H_1_1;
      float y2 = (b2 - H_0_2 * y0 - H_1_2 * y1) / H_2_2;
      float x2 = y2 / H_2_2;
      float x1 = (y1 - H_1_2 * x2) / H_1_1;
      float x0 = (y0 - H_0_2 * x2 - H_0_1 * x1) / H_0_0;
      if (x0 != 0) {
        bso_f3 p = bso_surfel_position(surfels, i);
        bso_f3 n = bso_surfel_normal(surfels, i);
        bso_surfel_set_position(surfels, i, bso_sub(p, bso_scale(x0, n)));
      }
      if (x1 != 0) {
        float d = BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR1, i);
        d -= x1;
        BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR1, i) = fmaxf(-180.f, fminf(180.f, d));
      }
      if (x2 != 0) {
        float d = BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR2, i);
        d -= x2;
        BSO_AT(float, surfels, BSLAM_SURFEL_DESCRIPTOR2, i) = fmaxf(-180.f, fminf(180.f, d));
      }
    }
  }
}
