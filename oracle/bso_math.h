/*
 * bso_math.h -- ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Plain-C restatement of the device math of the reference's bundle-adjustment hot
 * path.  Every function cites the reference file:line it follows.  The arithmetic is
 * fp32 and exists in TWO evaluation shapes, selected at run time by bso_literal_mode:
 *
 *   twin mode (0, default): the shape the HIP kernels use -- sums of products as explicit
 *     fmaf chains (BSO_FMA below), ONE correctly rounded reciprocal per projection
 *     (p.x * (1 / p.z)), "* fl(1 / baseline_fx)", the Cephes-style bso_expf.  Compiled with
 *     -ffp-contract=off, so the only fused operations are the ones spelled out; the kernels
 *     spell out the same ones, which is what makes the integer outputs (associated pixel,
 *     activation mask, packed normals) comparable bit for bit between CPU and GPU.
 *   literal mode (1): the reference's source expressions transcribed operator by operator in
 *     its operand order, unfused and with IEEE division: fx * (x / z) + cx
 *     (BS/surfel_projection.cuh:52-55), a.x*b.x + a.y*b.y + a.z*b.z (BS/cuda_util.cuh:52-58),
 *     row.x*p.x + row.y*p.y + row.z*p.z + row.w (BS/cuda_matrix.cuh:98-137),
 *     (...) / baseline_fx (BS/cost_function.cuh:81-83), 1 - x*x - y*y (BS/util.cuh:126), libm expf.
 *     tests/test_oracle_literal.py measures how far the twin has moved from this transcription:
 *     integer outputs may differ only on threshold ties, float outputs by rounding.
 *
 * BS/ = /root/reference/applications/badslam/src/badslam/
 *
 * Known, documented deviations of BOTH modes from what the CUDA build computes:
 *   - the reference is compiled with -use_fast_math (approximate division / sqrtf / expf and
 *     FMA contraction wherever nvcc likes), so its own last bits are not defined by its source;
 *   - tex2D() bilinear filtering is modelled in software (bso_tex_w below).
 */
#ifndef BSO_MATH_H_
#define BSO_MATH_H_

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/badslam_hip.h"

/* 0: twin of the HIP kernels (default); 1: literal transcription of the reference's expressions.  Defined in
 * bslam_oracle.c, set with bso_set_literal_mode(). */
extern int bso_literal_mode;
#define BSO_LITERAL __builtin_expect(bso_literal_mode != 0, 0)

typedef struct { float x, y, z; } bso_f3;
typedef struct { float x, y; } bso_f2;

static inline bso_f3 bso_make3(float x, float y, float z) { bso_f3 r = {x, y, z}; return r; }

/* float -> int as the GPU converts (CUDA cvt.rzi.s32.f32 and gfx950 v_cvt_i32_f32 both
 * truncate, saturate out-of-range values and map NaN to 0; x86 would return INT_MIN). */
static inline int bso_f2i(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int)v;
}

/* ---- BS/cuda_util.cuh:52-107 ------------------------------------------------ */
/* Sums of products are written as explicit fused multiply-adds, in ONE fixed shape shared with the HIP kernels
 * (csrc/device_math.hpp): the reference's nvcc build contracts a * b + c into FMAs wherever it likes, so no
 * unfused evaluation order is "the" reference; an explicit fmaf chain is exactly defined on both CPU and GPU
 * (so the integer outputs stay bit-comparable) and costs the kernels half the instructions.  -ffp-contract=off stays:
 * nothing is fused implicitly. */
#define BSO_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
static inline float bso_sqlen(bso_f3 v) {                                                        /* :52 */
  if (BSO_LITERAL) return v.x * v.x + v.y * v.y + v.z * v.z;
  return BSO_FMA(v.z, v.z, BSO_FMA(v.y, v.y, v.x * v.x));
}
static inline float bso_dot(bso_f3 a, bso_f3 b) {                                                /* :57 */
  if (BSO_LITERAL) return a.x * b.x + a.y * b.y + a.z * b.z;
  return BSO_FMA(a.z, b.z, BSO_FMA(a.y, b.y, a.x * b.x));
}
/* one row of a rigid transform / of a rotation applied to p (BS/cuda_matrix.cuh:98-137) */
static inline float bso_tr_row(float a, float b, float c, float d, bso_f3 p) {
  if (BSO_LITERAL) return a * p.x + b * p.y + c * p.z + d;
  return BSO_FMA(c, p.z, BSO_FMA(b, p.y, BSO_FMA(a, p.x, d)));
}
static inline float bso_rot_row(float a, float b, float c, bso_f3 p) {
  if (BSO_LITERAL) return a * p.x + b * p.y + c * p.z;
  return BSO_FMA(c, p.z, BSO_FMA(b, p.y, a * p.x));
}
static inline bso_f3 bso_cross(bso_f3 a, bso_f3 b) {                                            /* :78 */
  return bso_make3(a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y);
}
static inline float bso_norm(bso_f3 v) { return sqrtf(bso_sqlen(v)); }                           /* :85 */
static inline bso_f3 bso_add(bso_f3 a, bso_f3 b) { return bso_make3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline bso_f3 bso_sub(bso_f3 a, bso_f3 b) { return bso_make3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline bso_f3 bso_scale(float m, bso_f3 b) { return bso_make3(m * b.x, m * b.y, m * b.z); }

/* ---- BS/cuda_matrix.cuh:98-137 ---------------------------------------------- */
static inline bso_f3 bso_mul34(const bslam_mat3x4* T, bso_f3 p) {                               /* :98 */
  const float* m = T->m;
  return bso_make3(bso_tr_row(m[0], m[1], m[2], m[3], p), bso_tr_row(m[4], m[5], m[6], m[7], p), bso_tr_row(m[8], m[9], m[10], m[11], p));
}
static inline int bso_mul34_if_z_positive(const bslam_mat3x4* T, bso_f3 p, bso_f3* out) {       /* :113 */
  const float* m = T->m;
  out->z = bso_tr_row(m[8], m[9], m[10], m[11], p);
  if (out->z <= 0.f) return 0;
  out->x = bso_tr_row(m[0], m[1], m[2], m[3], p);
  out->y = bso_tr_row(m[4], m[5], m[6], m[7], p);
  return 1;
}
static inline bso_f3 bso_rotate34(const bslam_mat3x4* T, bso_f3 p) {                            /* :129 */
  const float* m = T->m;
  return bso_make3(bso_rot_row(m[0], m[1], m[2], p), bso_rot_row(m[4], m[5], m[6], p), bso_rot_row(m[8], m[9], m[10], p));
}
static inline bso_f3 bso_mul33(const bslam_mat3x3* R, bso_f3 p) {                               /* :64 */
  const float* m = R->m;
  return bso_make3(bso_rot_row(m[0], m[1], m[2], p), bso_rot_row(m[3], m[4], m[5], p), bso_rot_row(m[6], m[7], m[8], p));
}

/* ---- pitched buffers (libvis/src/libvis/cuda/cuda_buffer.cuh:61-83) ---------- */
#define BSO_AT(type, buf, y, x) \
  (*((type*)((char*)(buf)->address + (size_t)(y) * (buf)->pitch) + (x)))

/* ---- camera helpers (BS/surfel_projection.h:42-124, BS/surfel_projection.cuh) */
typedef struct { float fx_inv, fy_inv, cx_inv, cy_inv; } bso_unprojector;  /* PixelCenterUnprojector */
typedef struct { float fx, fy, cx, cy; int width, height; } bso_depth_to_color;  /* DepthToColorPixelCorner */

static inline bso_unprojector bso_make_unprojector(const bslam_camera4f* c) {   /* BS/surfel_projection.h:58-67 */
  bso_unprojector u;
  u.fx_inv = 1.0f / c->fx;
  u.fy_inv = 1.0f / c->fy;
  const float cx_pixel_center = c->cx - 0.5f;
  const float cy_pixel_center = c->cy - 0.5f;
  u.cx_inv = -cx_pixel_center * u.fx_inv;
  u.cy_inv = -cy_pixel_center * u.fy_inv;
  return u;
}
static inline float bso_unproj_nx(const bso_unprojector* u, float px) {   /* BS/surfel_projection.cuh:116 */
  if (BSO_LITERAL) return u->fx_inv * px + u->cx_inv;
  return BSO_FMA(u->fx_inv, px, u->cx_inv);
}
static inline float bso_unproj_ny(const bso_unprojector* u, float py) {
  if (BSO_LITERAL) return u->fy_inv * py + u->cy_inv;
  return BSO_FMA(u->fy_inv, py, u->cy_inv);
}
static inline bso_f3 bso_unproject(const bso_unprojector* u, int x, int y, float depth) {   /* BS/surfel_projection.cuh:110-114 */
  return bso_make3(depth * bso_unproj_nx(u, (float)x), depth * bso_unproj_ny(u, (float)y), depth);
}
static inline bso_f2 bso_project(float fx, float fy, float cx, float cy, bso_f3 p) {        /* BS/surfel_projection.cuh:52-55 */
  /* The reference writes p.x / p.z and p.y / p.z and builds with -use_fast_math (BS/CMakeLists.txt:67), under which
     nvcc evaluates a / b as a * rcp(b).  The restatement takes that shape with a correctly rounded reciprocal, so
     that CPU and GPU agree to the last bit at one division per projection. */
  bso_f2 r;
  if (BSO_LITERAL) {
    r.x = fx * (p.x / p.z) + cx;
    r.y = fy * (p.y / p.z) + cy;
    return r;
  }
  const float inv_z = 1.0f / p.z;
  r.x = BSO_FMA(fx, p.x * inv_z, cx);
  r.y = BSO_FMA(fy, p.y * inv_z, cy);
  return r;
}
static inline bso_depth_to_color bso_make_depth_to_color(const bslam_camera4f* depth, const bslam_camera4f* color) {
  /* BS/surfel_projection.h:105-124 */
  bso_depth_to_color r;
  r.width = color->width;
  r.height = color->height;
  r.fx = color->fx / depth->fx;
  r.cx = -1 * color->fx * depth->cx / depth->fx + color->cx;
  r.fy = color->fy / depth->fy;
  r.cy = -1 * color->fy * depth->cy / depth->fy + color->cy;
  return r;
}
static inline int bso_depth_to_color_pxy(bso_f2 pxy, const bso_depth_to_color* d, bso_f2* out) {   /* BS/surfel_projection.cuh:196-207 */
  out->x = d->fx * pxy.x + d->cx;
  out->y = d->fy * pxy.y + d->cy;
  return out->x >= 0 && out->y >= 0 && bso_f2i(out->x) < d->width && bso_f2i(out->y) < d->height;
}

/* exp(x) from plain fp32 multiplies and adds (Cephes expf: Cody-Waite reduction by ln 2, degree-5 polynomial on
 * |r| <= ln2 / 2, ~1 ulp), so that this oracle and the HIP kernels agree bit for bit; libm's and the device
 * library's expf do not.  (The reference itself evaluates exp with -use_fast_math, i.e. ex2.approx.)  exp(0) = 1. */
static inline float bso_expf(float x) {
  if (x < -87.0f) return 0.f;
  if (x > 88.0f) return INFINITY;
  const float n = floorf(x * 1.44269504088896341f + 0.5f);
  const float r = (x - n * 0.693359375f) - n * -2.12194440e-4f;
  float p = 1.9875691500e-4f;
  p = p * r + 1.3981999507e-3f;
  p = p * r + 8.3334519073e-3f;
  p = p * r + 4.1665795894e-2f;
  p = p * r + 1.6666665459e-1f;
  p = p * r + 5.0000001201e-1f;
  const float y = (p * r) * r + r + 1.0f;
  const uint32_t bits = (uint32_t)((int32_t)n + 127) << 23;
  float two_n;
  memcpy(&two_n, &bits, 4);
  return y * two_n;
}

/* ---- BS/util.cuh:46-53 ------------------------------------------------------- */
static inline float bso_raw_to_calibrated_depth(float a, float cfactor, float raw_to_float_depth, uint16_t measured_depth) {
  const float inv_depth = 1.0f / (raw_to_float_depth * measured_depth);
  if (BSO_LITERAL) return 1.f / (inv_depth + cfactor * expf(-a * inv_depth));
  return 1.f / (inv_depth + cfactor * bso_expf(-a * inv_depth));
}

/* ---- BS/util.cuh:101-130 ----------------------------------------------------- */
static inline int8_t bso_small_float_to_s8(float value) {                                    /* :102 */
  return (int8_t)(value * ((1 << 7) - 1) + ((value > 0) ? 0.5f : -0.5f));
}
static inline float bso_s8_to_small_float(int8_t value) { return value * (1.0f / ((1 << 7) - 1)); }   /* :107 */
static inline uint16_t bso_image_space_normal_to_u16(float x, float y) {                     /* :114 */
  return (uint16_t)(((uint16_t)(uint8_t)bso_small_float_to_s8(x)) << 0) |
         (uint16_t)(((uint16_t)(uint8_t)bso_small_float_to_s8(y)) << 8);
}
static inline bso_f3 bso_u16_to_image_space_normal(uint16_t value) {                         /* :120 */
  bso_f3 r;
  r.x = bso_s8_to_small_float((int8_t)(value & 0x00ff));
  r.y = bso_s8_to_small_float((int8_t)((value & 0xff00) >> 8));
  if (BSO_LITERAL) r.z = 1 - r.x * r.x - r.y * r.y;                                         /* :126 */
  else r.z = BSO_FMA(-r.y, r.y, BSO_FMA(-r.x, r.x, 1.0f));
  r.z = -sqrtf((r.z > 0.f) ? r.z : 0.f);
  return r;
}

/* ---- BS/util_nvcc_only.cuh:52-115 -------------------------------------------- */
static inline bso_f3 bso_surfel_position(const bslam_buffer2d* s, uint32_t i) {               /* :59 */
  return bso_make3(BSO_AT(float, s, BSLAM_SURFEL_X, i), BSO_AT(float, s, BSLAM_SURFEL_Y, i), BSO_AT(float, s, BSLAM_SURFEL_Z, i));
}
static inline void bso_surfel_set_position(const bslam_buffer2d* s, uint32_t i, bso_f3 p) {   /* :52 */
  BSO_AT(float, s, BSLAM_SURFEL_X, i) = p.x;
  BSO_AT(float, s, BSLAM_SURFEL_Y, i) = p.y;
  BSO_AT(float, s, BSLAM_SURFEL_Z, i) = p.z;
}
static inline uint32_t bso_small_float_to_s10(float value) {                                  /* :68 */
  return 0x03ff & (uint16_t)(int16_t)(value * ((1 << 9) - 1) + ((value > 0) ? 0.5f : -0.5f));
}
static inline float bso_s10_to_small_float(uint32_t value) {                                  /* :74 */
  uint16_t temp = (uint16_t)(((0x0200 & value) ? 0xfc00 : 0) | (0x03ff & value));
  int16_t s;
  memcpy(&s, &temp, 2);
  return s * (1.0f / ((1 << 9) - 1));
}
static inline void bso_surfel_set_normal(const bslam_buffer2d* s, uint32_t i, bso_f3 n) {     /* :80 */
  uint32_t v = (bso_small_float_to_s10(n.x) << 0) | (bso_small_float_to_s10(n.y) << 10) | (bso_small_float_to_s10(n.z) << 20);
  memcpy(&BSO_AT(float, s, BSLAM_SURFEL_NORMAL, i), &v, 4);
}
static inline bso_f3 bso_surfel_normal(const bslam_buffer2d* s, uint32_t i) {                 /* :88 */
  uint32_t value;
  memcpy(&value, &BSO_AT(float, s, BSLAM_SURFEL_NORMAL, i), 4);
  bso_f3 n = bso_make3(bso_s10_to_small_float(value >> 0), bso_s10_to_small_float(value >> 10), bso_s10_to_small_float(value >> 20));
  float factor = 1.0f / bso_norm(n);
  return bso_scale(factor, n);
}

/* ---- BS/robust_weighting.cuh:39-86 ------------------------------------------- */
static inline float bso_tukey_residual(float r, float k) {
  if (fabsf(r) < k) {
    const float quot = r / k;
    const float term = 1.f - quot * quot;
    return (1 / 6.f) * k * k * (1 - term * term * term);
  } else {
    return (1 / 6.f) * k * k;
  }
}
static inline float bso_tukey_weight(float r, float k) {
  if (fabsf(r) < k) {
    const float quot = r / k;
    const float term = 1.f - quot * quot;
    return term * term;
  } else {
    return 0.f;
  }
}
static inline float bso_huber_residual(float r, float k) {
  const float a = fabsf(r);
  if (a < k) return 0.5f * r * r;
  return k * (a - 0.5f * k);
}
static inline float bso_huber_weight(float r, float k) {
  const float a = fabsf(r);
  return (a < k) ? 1.f : (k / a);
}

/* ---- BS/cost_function.cuh:44-103 (depth residual) ----------------------------- */
#define BSO_DEPTH_RESIDUAL_WEIGHT 1.f          /* :44 */
#define BSO_DEPTH_TUKEY 10.f                   /* :48 */
#define BSO_DEPTH_UNCERTAINTY_FACTOR 0.1f      /* :52 */
#define BSO_COS_NORMAL_COMPAT 0.76604f         /* BS/kernels.cuh:58 */

static inline float bso_depth_stddev(float nx, float ny, float depth, bso_f3 n, float baseline_fx) {      /* :81-83 */
  if (BSO_LITERAL) return (BSO_DEPTH_UNCERTAINTY_FACTOR * fabsf(n.x * nx + n.y * ny + n.z) * (depth * depth)) / baseline_fx;
  /* ".../ baseline_fx" in the reference; same -use_fast_math shape as bso_project: times the rounded reciprocal */
  const float inv_baseline_fx = 1.0f / baseline_fx;
  return (BSO_DEPTH_UNCERTAINTY_FACTOR * fabsf(BSO_FMA(n.x, nx, BSO_FMA(n.y, ny, n.z))) * (depth * depth)) * inv_baseline_fx;
}
static inline float bso_depth_inv_stddev(float nx, float ny, float depth, bso_f3 n, float baseline_fx) {  /* :86-88 */
  if (BSO_LITERAL) return baseline_fx / (BSO_DEPTH_UNCERTAINTY_FACTOR * fabsf(n.x * nx + n.y * ny + n.z) * (depth * depth));
  return baseline_fx / (BSO_DEPTH_UNCERTAINTY_FACTOR * fabsf(BSO_FMA(n.x, nx, BSO_FMA(n.y, ny, n.z))) * (depth * depth));
}
static inline float bso_depth_weight(float r) { return BSO_DEPTH_RESIDUAL_WEIGHT * bso_tukey_weight(r, 1.f * BSO_DEPTH_TUKEY); }           /* :91-93 */
static inline float bso_weighted_depth_residual(float r) { return BSO_DEPTH_RESIDUAL_WEIGHT * bso_tukey_residual(r, 1.f * BSO_DEPTH_TUKEY); }  /* :96-98 */

/* ---- BS/cost_function.cuh:105-181 (descriptor residual) ------------------------ */
#define BSO_DESC_RESIDUAL_WEIGHT 1e-2f         /* :105 */
#define BSO_DESC_HUBER 10.f                    /* :109 */
static inline float bso_desc_weight(float r) { return 1.f * BSO_DESC_RESIDUAL_WEIGHT * bso_huber_weight(r, BSO_DESC_HUBER); }            /* :177-179 */
static inline float bso_weighted_desc_residual(float r) { return 1.f * BSO_DESC_RESIDUAL_WEIGHT * bso_huber_residual(r, BSO_DESC_HUBER); }  /* :183-185 */

/* ---- Jacobians ---------------------------------------------------------------------------------------
 * One definition per formula, used by every loop of the oracle (pose, geometry, PCG, intrinsics, odometry) and pinned
 * against the reference's own symbolic derivation (applications/badslam/scripts/jacobians_derivation.py, evaluated into
 * tests/golden/jacobian_golden.npz; tests/test_jacobian_golden.py).  The reference repeats these expressions in each
 * kernel; the citations name the first occurrence. */

/* Depth residual and its Jacobian wrt. the pose delta (BS/kernel_opt_pose.cu:45-94, BS/cost_function.cuh:56-68):
 * n_local = surfel normal in the frame, lu = unprojected pixel, ls = surfel position in the frame. */
static inline float bso_depth_residual(float inv_stddev, bso_f3 n_local, bso_f3 lu, bso_f3 ls) {
  return inv_stddev * bso_dot(n_local, bso_sub(lu, ls));
}
static inline void bso_jac_depth_pose(float inv_stddev, bso_f3 n_local, bso_f3 lu, float* J) {
  J[0] = inv_stddev * n_local.x;
  J[1] = inv_stddev * n_local.y;
  J[2] = inv_stddev * n_local.z;
  J[3] = inv_stddev * (-n_local.y * lu.z + n_local.z * lu.y);
  J[4] = inv_stddev * (n_local.x * lu.z - n_local.z * lu.x);
  J[5] = inv_stddev * (-n_local.x * lu.y + n_local.y * lu.x);
}
/* Depth residual wrt. a surfel move of t along its (unit) normal (BS/kernel_opt_geometry.cu:440, BS/kernel_pcg.cu:217). */
static inline float bso_jac_depth_position(float inv_stddev) { return -inv_stddev; }

/* Descriptor residual wrt. the pose delta (BS/kernel_opt_pose.cu:100-144); gx, gy = image gradient of the residual in
 * intensity per pixel TIMES the colour camera's fx, fy; ls = surfel position in the frame. */
static inline void bso_jac_desc_pose(float gx, float gy, bso_f3 ls, float* J) {
  float inv_ls_z = 1.f / ls.z;
  float ls_z_sq = ls.z * ls.z;
  float inv_ls_z_sq = inv_ls_z * inv_ls_z;
  J[0] = -gx * inv_ls_z;
  J[1] = -gy * inv_ls_z;
  J[2] = (ls.x * gx + ls.y * gy) * inv_ls_z_sq;
  float ls_x_y = ls.x * ls.y;
  J[3] = ((ls.y * ls.y + ls_z_sq) * gy + ls_x_y * gx) * inv_ls_z_sq;
  J[4] = -((ls.x * ls.x + ls_z_sq) * gx + ls_x_y * gy) * inv_ls_z_sq;
  J[5] = -(ls.x * gy - ls.y * gx) * inv_ls_z;
}
/* Descriptor residual wrt. a surfel move along its normal (BS/kernel_opt_geometry.cu:175-189, BS/kernel_pcg.cu:364-372):
 * rn = normal in the frame, ls = position in the frame; the gradient (gx, gy) is multiplied by (fx, fy) here (the PCG
 * kernels pass gradients that already carry the focal lengths and fx = fy = 1). */
static inline float bso_jac_desc_position(float gx, float gy, float fx, float fy, bso_f3 rn, bso_f3 ls) {
  const float term1 = -fx * (rn.x * ls.z - rn.z * ls.x);
  const float term2 = -fy * (rn.y * ls.z - rn.z * ls.y);
  const float term3 = 1.f / (ls.z * ls.z);
  return -(gx * term1 + gy * term2) * term3;
}
/* Depth residual wrt. (fx_inv, fy_inv, cx_inv, cy_inv, a, cfactor of the pixel's cell), dj[0..5]
 * (BS/kernel_opt_intrinsics.cu:82-118 = BS/kernel_pcg.cu:258-322).  m = frame_T_global (3x4 row-major), ln = normal in
 * the frame.  Returns corrected_inv_depth, on which the callers base their validity tests (:90, BS/kernel_pcg.cu:272). */
static inline float bso_jac_depth_intrinsics(float inv_stddev, float calibrated_depth, int px, int py, float nx, float ny, bso_f3 n_global,
                                             const float* m, bso_f3 ln, float cfactor, float a, float raw_inv_depth, float* dj) {
  const float exp_inv_depth = BSO_LITERAL ? expf(-a * raw_inv_depth) : bso_expf(-a * raw_inv_depth);
  const float corrected_inv_depth = cfactor * exp_inv_depth + raw_inv_depth;
  const float dot = bso_dot(bso_make3(nx, ny, 1), ln);
  const float jac_base = inv_stddev * dot * exp_inv_depth / (corrected_inv_depth * corrected_inv_depth);
  dj[2] = inv_stddev * calibrated_depth * bso_dot(n_global, bso_make3(m[0], m[1], m[2]));
  dj[3] = inv_stddev * calibrated_depth * bso_dot(n_global, bso_make3(m[4], m[5], m[6]));
  dj[0] = px * dj[2];
  dj[1] = py * dj[3];
  dj[4] = cfactor * raw_inv_depth * jac_base;
  dj[5] = -jac_base;
  return corrected_inv_depth;
}
/* Descriptor residual wrt. the colour camera's (fx, fy, cx, cy) (BS/kernel_opt_intrinsics.cu:141-149, BS/kernel_pcg.cu:462-509):
 * gx, gy = image gradient of the residual (no focal length), nx, ny = normalised image coordinates of the pixel. */
static inline void bso_jac_desc_color_intrinsics(float gx, float gy, float nx, float ny, float* j) {
  j[0] = gx * nx;
  j[1] = gy * ny;
  j[2] = gx;
  j[3] = gy;
}
/* Image gradient of a bilinear sample from its 2x2 texel footprint and the sample's fractional offsets (tx, ty)
 * (one block of BS/cost_function.cuh:200-239). */
static inline void bso_bilinear_gradient(float top_left, float top_right, float bottom_left, float bottom_right, float tx, float ty,
                                         float* dx, float* dy) {
  *dx = (bottom_right - bottom_left) * ty + (top_right - top_left) * (1 - ty);
  *dy = (bottom_right - top_right) * tx + (bottom_left - top_left) * (1 - tx);
}

/* ---- software model of the colour texture (BS/keyframe.cc:67-73) --------------
 * pitch2D uchar4, clamp addressing, cudaFilterModeLinear, cudaReadModeNormalizedFloat,
 * unnormalised coordinates; only the .w channel (luma) is ever read on this path.
 * Linear filtering per the CUDA programming guide (texture fetching appendix):
 *   xB = x - 0.5, i = floor(xB), alpha = frac(xB) (9-bit fixed point, 8 fractional bits)
 *   tex = (1-a)(1-b) T[i,j] + a(1-b) T[i+1,j] + (1-a) b T[i,j+1] + a b T[i+1,j+1].
 * mode BSLAM_TEX_FIXED_POINT_1_8 rounds alpha/beta to multiples of 1/256,
 * mode BSLAM_TEX_EXACT_FLOAT keeps the fp32 fractions.  The exact rounding inside
 * NVIDIA's texture unit is not published: PARITY UNPINNED for this one function. */
static inline float bso_texel_w(const bslam_buffer2d* color, int ix, int iy) {
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (ix > color->width - 1) ix = color->width - 1;
  if (iy > color->height - 1) iy = color->height - 1;
  const uint8_t* px = (const uint8_t*)color->address + (size_t)iy * color->pitch + 4 * (size_t)ix;
  return px[3] * (1.0f / 255.0f);
}
/* Twin shape of the sampling arithmetic (shared with csrc/device_math.hpp; the literal shape is the CUDA programming guide's
 * formula as written, above and in bso_tex_w / bso_point_gradient below):
 *   - texels stay raw bytes 0..255 (exact in fp32; their differences are exact too) and the 1 / 255 normalisation is folded
 *     into the residual's intensity scale: 180 * (i1 - i0) = (180 / 255) * (b1 - b0);
 *   - bilinear value as nested interpolation with explicit fused multiply-adds: top = a (tr - tl) + tl, bot = a (br - bl) + bl,
 *     value = b (bot - top) + top  (3 roundings instead of 11);
 *   - gradient dx = ty ((br - bl) - (tr - tl)) + (tr - tl), dy = tx ((br - tr) - (bl - tl)) + (bl - tl)  (1 rounding each).
 * Only the residual path (descriptor residual and its image gradient) takes this shape; colours and downsampled images,
 * whose outputs are integers, keep the literal filter in both modes. */
#define BSO_DESC_SCALE (180.f / 255.f)
static inline float bso_texel_b(const bslam_buffer2d* color, int ix, int iy) {     /* luma byte 0..255, clamp addressing */
  if (ix < 0) ix = 0;
  if (iy < 0) iy = 0;
  if (ix > color->width - 1) ix = color->width - 1;
  if (iy > color->height - 1) iy = color->height - 1;
  return (float)((const uint8_t*)color->address + (size_t)iy * color->pitch + 4 * (size_t)ix)[3];
}
static inline float bso_quantise_weight(float a, int mode) {
  if (mode != BSLAM_TEX_FIXED_POINT_1_8) return a;
  return floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f);
}
static inline float bso_bilinear_bytes(float tl, float tr, float bl, float br, float a, float b) {
  const float top = BSO_FMA(a, tr - tl, tl);
  const float bot = BSO_FMA(a, br - bl, bl);
  return BSO_FMA(b, bot - top, top);
}
static inline void bso_bilinear_gradient_bytes(float tl, float tr, float bl, float br, float tx, float ty, float* dx, float* dy) {
  *dx = BSO_FMA(ty, (br - bl) - (tr - tl), tr - tl);
  *dy = BSO_FMA(tx, (br - tr) - (bl - tl), bl - tl);
}
/* twin: bilinear luma in BYTE units at pixel-corner coordinates (x, y) */
static inline float bso_tex_b(const bslam_buffer2d* color, float x, float y, int mode) {
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fx = floorf(xb), fy = floorf(yb);
  const float a = bso_quantise_weight(xb - fx, mode), b = bso_quantise_weight(yb - fy, mode);
  const int i = (int)fminf(fmaxf(fx, -2.0f), (float)color->width);
  const int j = (int)fminf(fmaxf(fy, -2.0f), (float)color->height);
  return bso_bilinear_bytes(bso_texel_b(color, i, j), bso_texel_b(color, i + 1, j), bso_texel_b(color, i, j + 1), bso_texel_b(color, i + 1, j + 1), a, b);
}
static inline float bso_tex_w(const bslam_buffer2d* color, float x, float y, int mode) {
  const float xb = x - 0.5f;
  const float yb = y - 0.5f;
  const float fx = floorf(xb);
  const float fy = floorf(yb);
  float a = xb - fx;
  float b = yb - fy;
  if (!BSO_LITERAL) return bso_tex_b(color, x, y, mode) * (1.0f / 255.0f);
  a = bso_quantise_weight(a, mode);
  b = bso_quantise_weight(b, mode);
  /* clamp in float first so that huge coordinates cannot overflow the int conversion */
  const float fxc = fminf(fmaxf(fx, -2.0f), (float)color->width);
  const float fyc = fminf(fmaxf(fy, -2.0f), (float)color->height);
  const int i = (int)fxc;
  const int j = (int)fyc;
  const float t00 = bso_texel_w(color, i, j);
  const float t10 = bso_texel_w(color, i + 1, j);
  const float t01 = bso_texel_w(color, i, j + 1);
  const float t11 = bso_texel_w(color, i + 1, j + 1);
  const float w00 = (1.0f - a) * (1.0f - b);
  const float w10 = a * (1.0f - b);
  const float w01 = (1.0f - a) * b;
  const float w11 = a * b;
  return ((w00 * t00 + w10 * t10) + w01 * t01) + w11 * t11;
}

/* BS/cost_function.cuh:115-136 */
static inline void bso_tangent_projections(bso_f3 gp, bso_f3 gn, float radius_squared, const bslam_mat3x4* frame_T_global,
                                           float cfx, float cfy, float ccx, float ccy, bso_f2* t1_pxy, bso_f2* t2_pxy) {
  const float kTangentScaling = 2.0f;
  bso_f3 t1 = bso_cross(gn, (fabsf(gn.x) > 0.9f) ? bso_make3(0, 1, 0) : bso_make3(1, 0, 0));
  t1 = bso_scale(sqrtf(radius_squared / fmaxf(1e-12f, bso_sqlen(t1))), bso_scale(kTangentScaling, t1));
  *t1_pxy = bso_project(cfx, cfy, ccx, ccy, bso_mul34(frame_T_global, bso_add(gp, t1)));
  bso_f3 t2 = bso_cross(gn, t1);
  t2 = bso_scale(sqrtf(radius_squared / fmaxf(1e-12f, bso_sqlen(t2))), bso_scale(kTangentScaling, t2));
  *t2_pxy = bso_project(cfx, cfy, ccx, ccy, bso_mul34(frame_T_global, bso_add(gp, t2)));
}

/* BS/cost_function.cuh:140-156 */
static inline void bso_raw_descriptor_residual(const bslam_buffer2d* color, int tex_mode, bso_f2 pxy, bso_f2 t1, bso_f2 t2,
                                               float d1, float d2, float* r1, float* r2) {
  if (!BSO_LITERAL) {
    const float b0 = bso_tex_b(color, pxy.x, pxy.y, tex_mode), b1 = bso_tex_b(color, t1.x, t1.y, tex_mode), b2 = bso_tex_b(color, t2.x, t2.y, tex_mode);
    *r1 = BSO_FMA(BSO_DESC_SCALE, b1 - b0, -d1);
    *r2 = BSO_FMA(BSO_DESC_SCALE, b2 - b0, -d2);
    return;
  }
  float intensity = bso_tex_w(color, pxy.x, pxy.y, tex_mode);
  float t1_intensity = bso_tex_w(color, t1.x, t1.y, tex_mode);
  float t2_intensity = bso_tex_w(color, t2.x, t2.y, tex_mode);
  *r1 = (180.f * (t1_intensity - intensity)) - d1;
  *r2 = (180.f * (t2_intensity - intensity)) - d2;
}

/* one of the three blocks of BS/cost_function.cuh:200-239: finite-difference image
 * gradient at p from the four texel centres around it (tex2D at texel centres
 * returns the texel itself, for both filter models). */
static inline void bso_point_gradient(const bslam_buffer2d* color, bso_f2 p, float* dx, float* dy) {
  int ix = bso_f2i(fmaxf(0.f, p.x - 0.5f));
  int iy = bso_f2i(fmaxf(0.f, p.y - 0.5f));
  float tx = fmaxf(0.f, fminf(1.f, p.x - 0.5f - ix));
  float ty = fmaxf(0.f, fminf(1.f, p.y - 0.5f - iy));
  /* clamp before the +1 so that a saturated index cannot overflow (same texels as clamp addressing) */
  if (ix > color->width - 1) ix = color->width - 1;
  if (iy > color->height - 1) iy = color->height - 1;
  if (!BSO_LITERAL) {   /* twin: gradient in BYTE units per pixel (bso_descriptor_jacobian_wrt_projected_position applies 180 / 255) */
    bso_bilinear_gradient_bytes(bso_texel_b(color, ix, iy), bso_texel_b(color, ix + 1, iy), bso_texel_b(color, ix, iy + 1), bso_texel_b(color, ix + 1, iy + 1),
                                tx, ty, dx, dy);
    return;
  }
  float top_left = bso_texel_w(color, ix, iy);
  float top_right = bso_texel_w(color, ix + 1, iy);
  float bottom_left = bso_texel_w(color, ix, iy + 1);
  float bottom_right = bso_texel_w(color, ix + 1, iy + 1);
  bso_bilinear_gradient(top_left, top_right, bottom_left, bottom_right, tx, ty, dx, dy);
}

/* BS/cost_function.cuh:191-254 (the three unused fetches :241-243 are dropped, quirk Q3) */
static inline void bso_descriptor_jacobian_wrt_projected_position(const bslam_buffer2d* color, bso_f2 c, bso_f2 t1, bso_f2 t2,
                                                                  float* gx1, float* gy1, float* gx2, float* gy2) {
  float cdx, cdy, t1dx, t1dy, t2dx, t2dy;
  bso_point_gradient(color, c, &cdx, &cdy);
  bso_point_gradient(color, t1, &t1dx, &t1dy);
  bso_point_gradient(color, t2, &t2dx, &t2dy);
  const float scale = BSO_LITERAL ? 180.f : BSO_DESC_SCALE;   /* the twin's gradients are in byte units */
  *gx1 = scale * (t1dx - cdx);
  *gy1 = scale * (t1dy - cdy);
  *gx2 = scale * (t2dx - cdx);
  *gy2 = scale * (t2dy - cdy);
}

/* ---- association (BS/surfel_projection_nvcc_only.cuh:49-127, 302-332) ---------- */
typedef struct {
  bso_f3 global_position;
  bso_f3 local_position;
  bso_f3 surfel_normal;        /* global */
  float calibrated_depth;
  int px, py;
  bso_f2 pxy;
} bso_projection;

/* BS/util.cuh:67-99 */
static inline int bso_project_surfel_to_image(int width, int height, const bslam_camera4f* cam, bso_f3 local, int* px, int* py, bso_f2* pxy) {
  *pxy = bso_project(cam->fx, cam->fy, cam->cx, cam->cy, local);
  *px = bso_f2i(pxy->x);
  *py = bso_f2i(pxy->y);
  if (pxy->x < 0 || pxy->y < 0 || *px >= width || *py >= height) return 0;
  return 1;
}

/* IsAssociatedWithPixel<false, true> BS/surfel_projection_nvcc_only.cuh:49-127 */
static inline int bso_is_associated_with_pixel(const bslam_buffer2d* surfels, uint32_t surfel_index, bso_f3 local_position,
                                               const bslam_mat3x4* frame_T_global, const bslam_buffer2d* normals_buffer,
                                               int px, int py, const bslam_depth_params* dp, uint16_t measured_depth,
                                               float depth_tukey_parameter, const bso_unprojector* unproj,
                                               bso_f3* surfel_normal, float* out_calibrated_depth) {
  if (measured_depth & BSLAM_INVALID_DEPTH_BIT) return 0;
  float calibrated_depth = bso_raw_to_calibrated_depth(
      dp->a, BSO_AT(float, &dp->cfactor_buffer, py / dp->sparse_surfel_cell_size, px / dp->sparse_surfel_cell_size),
      dp->raw_to_float_depth, measured_depth);
  *out_calibrated_depth = calibrated_depth;
  *surfel_normal = bso_surfel_normal(surfels, surfel_index);
  bso_f3 local_normal_s = bso_rotate34(frame_T_global, *surfel_normal);
  float stddev = bso_depth_stddev(bso_unproj_nx(unproj, px), bso_unproj_ny(unproj, py), calibrated_depth, local_normal_s, dp->baseline_fx);
  const float thr = depth_tukey_parameter * stddev;
  if (fabsf(local_position.z - calibrated_depth) > thr) return 0;
  float surfel_distance = bso_norm(local_position);
  float dot_angle = (1.0f / surfel_distance) * bso_dot(local_position, local_normal_s);
  if (dot_angle > 0) return 0;
  bso_f3 local_normal = bso_u16_to_image_space_normal(BSO_AT(uint16_t, normals_buffer, py, px));
  float dot2 = bso_dot(local_normal_s, local_normal);
  if (dot2 < BSO_COS_NORMAL_COMPAT) return 0;
  return 1;
}

/* SurfelProjectsToAssociatedPixel (all result variants) BS/surfel_projection_nvcc_only.cuh:302-360,416-511 */
static inline int bso_surfel_projects_to_associated_pixel(uint32_t surfel_index, uint32_t surfels_size, const bslam_buffer2d* surfels,
                                                          const bslam_buffer2d* depth_buffer, const bslam_buffer2d* normals_buffer,
                                                          const bslam_depth_params* dp, const bslam_camera4f* depth_camera,
                                                          const bso_unprojector* unproj, const bslam_mat3x4* frame_T_global,
                                                          bso_projection* r) {
  if (surfel_index >= surfels_size) return 0;
  r->global_position = bso_surfel_position(surfels, surfel_index);
  if (!bso_mul34_if_z_positive(frame_T_global, r->global_position, &r->local_position)) return 0;
  if (!bso_project_surfel_to_image(depth_buffer->width, depth_buffer->height, depth_camera, r->local_position, &r->px, &r->py, &r->pxy)) return 0;
  return bso_is_associated_with_pixel(surfels, surfel_index, r->local_position, frame_T_global, normals_buffer, r->px, r->py, dp,
                                      BSO_AT(uint16_t, depth_buffer, r->py, r->px), BSO_DEPTH_TUKEY, unproj,
                                      &r->surfel_normal, &r->calibrated_depth);
}

#endif /* BSO_MATH_H_ */
