// device_math.hpp -- gfx950 device functions of the BA hot path.
//
// Projection, association, residuals and Jacobians of the reference's cost function,
// written for CDNA4.  Built with -ffp-contract=off and IEEE division / sqrt so that the
// predicate chain that decides the integer outputs (associated pixel, activation mask)
// rounds the same way on every run and matches the CPU oracle bit for bit.
//
// Reference formulas: BS/ = applications/badslam/src/badslam/ of pangfumin/badslam;
// each function names the file:line whose behaviour it reproduces.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/badslam_hip.h"

namespace bslam {

struct f3 { float x, y, z; };
struct f2 { float x, y; };

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
// Sums of products are explicit fused multiply-adds in ONE fixed shape shared with the CPU oracle: the reference's nvcc
// build contracts a * b + c wherever it likes, so no unfused order is "the" reference; an explicit fma chain is exactly
// defined on CPU and GPU alike (integer outputs stay bit-comparable) at half the instructions.  -ffp-contract=off
// stays: nothing is fused implicitly.
__device__ __forceinline__ float sqlen(f3 v) { return __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)); }           // BS/cuda_util.cuh:52
__device__ __forceinline__ float dot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }       // BS/cuda_util.cuh:57
// one row of a rigid transform / of a rotation applied to p
__device__ __forceinline__ float tr_row(float a, float b, float c, float d, f3 p) { return __builtin_fmaf(c, p.z, __builtin_fmaf(b, p.y, __builtin_fmaf(a, p.x, d))); }
__device__ __forceinline__ float rot_row(float a, float b, float c, f3 p) { return __builtin_fmaf(c, p.z, __builtin_fmaf(b, p.y, a * p.x)); }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {                                                     // BS/cuda_util.cuh:78
  return mk3(a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ float norm3(f3 v) { return sqrtf(sqlen(v)); }                            // BS/cuda_util.cuh:85
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 scale3(float m, f3 b) { return mk3(m * b.x, m * b.y, m * b.z); }

// 3x4 rigid transform held in registers / SGPRs (= CUDAMatrix3x4, BS/cuda_matrix.cuh:98-137).
struct M34 { float m[12]; };

__device__ __forceinline__ f3 mul34(const M34& T, f3 p) {
  return mk3(tr_row(T.m[0], T.m[1], T.m[2], T.m[3], p), tr_row(T.m[4], T.m[5], T.m[6], T.m[7], p), tr_row(T.m[8], T.m[9], T.m[10], T.m[11], p));
}
__device__ __forceinline__ f3 rot34(const M34& T, f3 p) {
  return mk3(rot_row(T.m[0], T.m[1], T.m[2], p), rot_row(T.m[4], T.m[5], T.m[6], p), rot_row(T.m[8], T.m[9], T.m[10], p));
}

// Per-launch camera constants, computed on the host exactly as the reference's factories do
// (BS/surfel_projection.h:42-124) and passed by value.
struct CamConsts {
  // depth camera, pixel-corner projector
  float fx, fy, cx, cy;
  int width, height;
  // PixelCenterUnprojector of the depth camera
  float fx_inv, fy_inv, cx_inv, cy_inv;
  // DepthToColorPixelCorner
  float d2c_fx, d2c_fy, d2c_cx, d2c_cy;
  int color_width, color_height;
  // colour camera (corner convention; the centre projector only differs in cx, cy which are unused)
  float cfx, cfy, ccx, ccy;
  // DepthParameters scalars
  float a, raw_to_float_depth, baseline_fx, inv_baseline_fx;
  int cell;
  const float* cfactor;   // device image
  uint32_t cfactor_pitch; // bytes
  int cfactor_width;
  int tex_mode;
  int d2c_identity;       // d2c_fx = d2c_fy = 1, d2c_cx = d2c_cy = 0 and equal image sizes (set by make_cam_consts)
};

struct __attribute__((aligned(16))) PixelRecord { float depth, nx, ny, nz; };

// Device view of one keyframe (pointers + pose), kept in a device array and read with scalar loads.
struct KfDev {
  const uint8_t* depth;    uint32_t depth_pitch;
  const uint8_t* normals;  uint32_t normals_pitch;
  const uint8_t* color;    uint32_t color_pitch;
  const uint8_t* radius;   uint32_t radius_pitch;   // half radius^2 image (only the surfel lifecycle reads it)
  // derived per-pixel records {f32 calibrated depth (0: no measurement), decoded pixel normal x, y, z}, row pitch in
  // records = image width (library-owned, rebuilt by build_records_kernel)
  const PixelRecord* records;
  // derived luma quads: quads[(j + 1) * (color_width + 1) + (i + 1)] packs the 2x2 texel footprint
  // {L(i,j), L(i+1,j), L(i,j+1), L(i+1,j+1)} (clamp addressing, i in [-1, w-1], j in [-1, h-1]) so that a
  // bilinear sample is ONE 4-byte gather instead of four byte gathers (library-owned, build_quads_kernel)
  const uint32_t* quads;
  M34 frame_T_global;
  float global_R_frame[9];
  int activation;
  int id;
};

__device__ __forceinline__ float nx_of(const CamConsts& c, float px) { return __builtin_fmaf(c.fx_inv, px, c.cx_inv); }   // BS/surfel_projection.cuh:116
__device__ __forceinline__ float ny_of(const CamConsts& c, float py) { return __builtin_fmaf(c.fy_inv, py, c.cy_inv); }
__device__ __forceinline__ f3 unproject(const CamConsts& c, int x, int y, float depth) {                    // BS/surfel_projection.cuh:110
  return mk3(depth * nx_of(c, (float)x), depth * ny_of(c, (float)y), depth);
}
// Correctly rounded 1 / x for 2^-96 < |x| < 2^96: the compiler's expansion of an IEEE division (v_div_scale x 2, v_rcp, four
// fmas, a multiply, v_div_fmas, v_div_fixup) with the parts that only serve numerators != 1, denormal scaling and special
// operands removed -- the same v_rcp seed and the same fma chain, so the same bits as 1.0f / x there, in 7 instead of 11
// instructions.  (tests/test_gpu_parity.py::test_midrange_reciprocal_is_correctly_rounded sweeps it against 1.0f / x.)
__device__ __forceinline__ float rcp_rn_midrange(float x) {
  const float r0 = __builtin_amdgcn_rcpf(x);
  const float e0 = __builtin_fmaf(-x, r0, 1.0f);
  const float r1 = __builtin_fmaf(e0, r0, r0);
  const float e1 = __builtin_fmaf(-x, r1, 1.0f);
  const float r2 = __builtin_fmaf(e1, r1, r1);
  const float e2 = __builtin_fmaf(-x, r2, 1.0f);
  return __builtin_fmaf(e2, r1, r2);
}

__device__ __forceinline__ f2 project(float fx, float fy, float cx, float cy, f3 p) {                       // BS/surfel_projection.cuh:52
  // one correctly rounded reciprocal + two multiplies (the CPU oracle's projection has the same shape; the reference's
  // -use_fast_math build evaluates p.x / p.z as p.x * rcp(p.z))
  const float inv_z = rcp_rn_midrange(p.z);
  return f2{__builtin_fmaf(fx, p.x * inv_z, cx), __builtin_fmaf(fy, p.y * inv_z, cy)};
}

// exp(x) from plain fp32 multiplies and adds (Cephes expf: Cody-Waite reduction by ln 2, degree-5 polynomial,
// ~1 ulp).  The device library's expf and the CPU's libm do not agree to the last bit; the depth deformation
// exp(-a / d) feeds the association predicate, whose integer outputs are compared bit for bit with the oracle, so
// both sides evaluate this same sequence.  (The reference evaluates exp under -use_fast_math.)
__device__ __forceinline__ float det_expf(float x) {
  if (x < -87.0f) return 0.f;
  if (x > 88.0f) return __uint_as_float(0x7f800000u);
  const float n = floorf(x * 1.44269504088896341f + 0.5f);
  const float r = (x - n * 0.693359375f) - n * -2.12194440e-4f;
  float p = 1.9875691500e-4f;
  p = p * r + 1.3981999507e-3f;
  p = p * r + 8.3334519073e-3f;
  p = p * r + 4.1665795894e-2f;
  p = p * r + 1.6666665459e-1f;
  p = p * r + 5.0000001201e-1f;
  const float y = (p * r) * r + r + 1.0f;
  return y * __uint_as_float((uint32_t)((int32_t)n + 127) << 23);
}

// BS/util.cuh:46-53.  exp(-a * inv_depth) is exactly 1 for a == 0 (the default, no depth
// deformation), so that case skips the evaluation without changing a bit.
__device__ __forceinline__ float raw_to_calibrated_depth(float a, float cfactor, float raw_to_float_depth, uint32_t measured_depth) {
  const float inv_depth = 1.0f / (raw_to_float_depth * (float)measured_depth);
  const float e = (a == 0.f) ? 1.0f : det_expf(-a * inv_depth);
  return 1.f / (inv_depth + cfactor * e);
}

// Correctly rounded sqrt for x == 0 or 2^-96 <= x < 2^96: v_sqrt_f32 (<= 1 ulp) plus the two residual tests of the
// compiler's general expansion; its rescaling of tiny inputs and the final class test are dead for such x (7 of 15
// instructions).  Same bits as sqrtf() there.
__device__ __forceinline__ float sqrt_rn_midrange(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
  const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
  const float r_dn = __builtin_fmaf(-s_dn, s, x);
  const float r_up = __builtin_fmaf(-s_up, s, x);
  float r = (r_dn <= 0.f) ? s_dn : s;
  r = (r_up > 0.f) ? s_up : r;
  return r;
}

// BS/util.cuh:107-130
__device__ __forceinline__ f3 u16_to_image_space_normal(uint32_t value) {
  f3 r;
  r.x = (float)(int8_t)(value & 0xff) * (1.0f / 127);
  r.y = (float)(int8_t)((value >> 8) & 0xff) * (1.0f / 127);
  r.z = __builtin_fmaf(-r.y, r.y, __builtin_fmaf(-r.x, r.x, 1.0f));
  r.z = -sqrt_rn_midrange((r.z > 0.f) ? r.z : 0.f);   // 0, or >= 2^-25: a difference of numbers in [0, 1] rounded to 2^-24
  return r;
}

// BS/util_nvcc_only.cuh:67-95
__device__ __forceinline__ uint32_t small_float_to_s10(float value) {
  return 0x03ffu & (uint32_t)(uint16_t)(int16_t)(value * 511 + ((value > 0) ? 0.5f : -0.5f));
}
__device__ __forceinline__ float s10_to_small_float(uint32_t value) {
  const int32_t s = ((int32_t)(value << 22)) >> 22;   // sign-extend 10 bits
  return (float)s * (1.0f / 511);
}
__device__ __forceinline__ uint32_t pack_normal(f3 n) {
  return (small_float_to_s10(n.x) << 0) | (small_float_to_s10(n.y) << 10) | (small_float_to_s10(n.z) << 20);
}
__device__ __forceinline__ f3 unpack_normal(uint32_t value) {
  f3 n = mk3(s10_to_small_float(value >> 0), s10_to_small_float(value >> 10), s10_to_small_float(value >> 20));
  const float factor = 1.0f / norm3(n);
  return scale3(factor, n);
}

// Division / reciprocal / square root of the RESIDUAL math.  The reference builds its kernels with
// -use_fast_math (BS/CMakeLists.txt:67), i.e. with approximate division and square root throughout; here
//  - everything that feeds an INTEGER output (project_and_associate, normal quantisation) uses the correctly
//    rounded forms, so associations / counts / packed normals are bit-identical to the oracle's;
//  - robust weights, inverse stddev and Jacobian factors (rdiv, rrcp) use the 1-ulp v_rcp_f32: smooth functions, a
//    1-ulp input change is a ~1e-7 relative output change;
//  - the tangent sample points (sdiv, srcp, ssqrt) stay correctly rounded: with the 1.8 fixed-point texture weights a
//    1-ulp move of a sample point can flip a quantised weight (a ~1e-4..1e-3 jump of one descriptor residual;
//    measured in round 2 with the raw v_rcp_f32 there).
__device__ __forceinline__ float rdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float rrcp(float b) { return __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float sdiv(float a, float b) { return a / b; }
__device__ __forceinline__ float srcp(float b) { return rcp_rn_midrange(b); }
__device__ __forceinline__ float ssqrt(float x) { return sqrtf(x); }

// BS/robust_weighting.cuh:39-86
__device__ __forceinline__ float tukey_weight(float r, float k) {
  if (fabsf(r) < k) { const float q = rdiv(r, k); const float t = 1.f - q * q; return t * t; }
  return 0.f;
}
__device__ __forceinline__ float tukey_residual(float r, float k) {
  if (fabsf(r) < k) { const float q = rdiv(r, k); const float t = 1.f - q * q; return (1 / 6.f) * k * k * (1 - t * t * t); }
  return (1 / 6.f) * k * k;
}
__device__ __forceinline__ float huber_weight(float r, float k) { const float a = fabsf(r); return (a < k) ? 1.f : rdiv(k, a); }
__device__ __forceinline__ float huber_residual(float r, float k) {
  const float a = fabsf(r);
  return (a < k) ? 0.5f * r * r : k * (a - 0.5f * k);
}

// BS/cost_function.cuh:44-98, 105-185
constexpr float kDepthTukey = 10.f;
constexpr float kDepthUncertainty = 0.1f;
constexpr float kCosNormalCompat = 0.76604f;   // BS/kernels.cuh:58
constexpr float kDescWeight = 1e-2f;
constexpr float kDescHuber = 10.f;

__device__ __forceinline__ float depth_stddev(float nx, float ny, float depth, f3 n, float inv_baseline_fx) {   // inv_baseline_fx = fl(1 / baseline_fx)
  return (kDepthUncertainty * fabsf(__builtin_fmaf(n.x, nx, __builtin_fmaf(n.y, ny, n.z))) * (depth * depth)) * inv_baseline_fx;
}
__device__ __forceinline__ float depth_inv_stddev(float nx, float ny, float depth, f3 n, float baseline_fx) {
  return rdiv(baseline_fx, kDepthUncertainty * fabsf(__builtin_fmaf(n.x, nx, __builtin_fmaf(n.y, ny, n.z))) * (depth * depth));
}
__device__ __forceinline__ float depth_weight(float r) { return 1.f * tukey_weight(r, 1.f * kDepthTukey); }
__device__ __forceinline__ float weighted_depth_residual(float r) { return 1.f * tukey_residual(r, 1.f * kDepthTukey); }
__device__ __forceinline__ float desc_weight(float r) { return 1.f * kDescWeight * huber_weight(r, kDescHuber); }
__device__ __forceinline__ float weighted_desc_residual(float r) { return 1.f * kDescWeight * huber_residual(r, kDescHuber); }

// float -> int, truncating and saturating (v_cvt_i32_f32; CUDA's cvt.rzi.s32.f32 behaves the same)
__device__ __forceinline__ int f2i(float v) { return (int)v; }

// ---------------------------------------------------------------------------------------------
// Colour sampling.  MI355X has no texture unit; clamp addressing, u8 -> [0,1] normalisation and
// bilinear filtering of the reference's texture (BS/keyframe.cc:67-73) are ALU code over plain
// global loads of the luma byte (.w of the uchar4 pixel).
// ---------------------------------------------------------------------------------------------
// Loads through the global address space: pointers read out of the keyframe table are generic
// pointers to the compiler, which would otherwise emit flat_load.
template <class T>
__device__ __forceinline__ T gload(const T* p) {
  return *(const __attribute__((address_space(1))) T*)(p);
}

// The same at a 32-bit BYTE offset from a wave-uniform base: the load takes the base from an SGPR pair and the offset from one
// VGPR (global_load ... v_off, s[base:base+1]) -- no 64-bit address arithmetic per lane, one VGPR per address instead of two.
template <class T>
__device__ __forceinline__ T gload_at(const T* base, uint32_t byte_offset) {
  return *(const __attribute__((address_space(1))) T*)((const __attribute__((address_space(1))) char*)base + byte_offset);
}

__device__ __forceinline__ uint2 gload_u2(const uint2* p) {   // one 8-byte load
  const unsigned long long v = gload((const unsigned long long*)p);
  return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ PixelRecord gload_record(const PixelRecord* p) {   // one 16-byte load
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = gload((const v4f*)p);
  return PixelRecord{v.x, v.y, v.z, v.w};
}

struct LumaQuad { float tl, tr, bl, br; };   // texels (i,j), (i+1,j), (i,j+1), (i+1,j+1) in [0,1]

__device__ __forceinline__ LumaQuad unpack_quad(uint32_t q) {
  LumaQuad r;
  r.tl = (float)(q & 0xffu) * (1.0f / 255.0f);
  r.tr = (float)((q >> 8) & 0xffu) * (1.0f / 255.0f);
  r.bl = (float)((q >> 16) & 0xffu) * (1.0f / 255.0f);
  r.br = (float)(q >> 24) * (1.0f / 255.0f);
  return r;
}

// i in [-1, w-1], j in [-1, h-1]
__device__ __forceinline__ uint32_t quad_at(const KfDev& kf, const CamConsts& c, int i, int j) {
  // byte offset inside one keyframe's table: < 2^26, 32-bit arithmetic (v_mul_u32_u24, v_lshl_add_u32) on top of the uniform base
  // (j + 1) * pitch + (i + 1) * 4 as  j * pitch + (4 i + (pitch + 4)):  v_lshl_add_u32 + v_mad_i32_i24 (j and i may be -1; the sum is not)
  const int pitch = 4 * (c.color_width + 1);
  const uint32_t t = ((uint32_t)i << 2) + (uint32_t)(pitch + 4);
  uint32_t off;   // written out: the compiler splits the constant off again and spends a third instruction on it
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(off) : "v"(j), "s"(pitch), "v"(t));
  return gload_at(kf.quads, off);
}

// Bilinear footprint of a sample at pixel-corner coordinates (x, y): base texel (clamped to the quad
// table's range, which reproduces clamp addressing exactly) and the two filter weights.
struct TexFootprint {
  int i, j;
  float a, b;        // filter weights (quantised to 1/256 in the fixed-point texture mode)
  float ua, ub;      // the unquantised fractions
  bool interior;     // 0 <= x - 0.5 < w - 1 and 0 <= y - 0.5 < h - 1: no clamping anywhere in the 2x2 footprint
};
// the filter weights of a footprint from its fractions (quantised to 1/256 in the fixed-point texture mode)
__device__ __forceinline__ void tex_weights(const CamConsts& c, TexFootprint* f) {
  f->a = f->ua;
  f->b = f->ub;
  if (c.tex_mode == BSLAM_TEX_FIXED_POINT_1_8) {
    // frac * 256 is exact in fp32 (a power-of-two scaling of a value in [0, 1)): fma(frac, 256, 0.5) rounds once, exactly where
    // frac * 256 + 0.5 rounds -- the same bits, one instruction less per weight
    f->a = floorf(__builtin_fmaf(f->ua, 256.0f, 0.5f)) * (1.0f / 256.0f);
    f->b = floorf(__builtin_fmaf(f->ub, 256.0f, 0.5f)) * (1.0f / 256.0f);
  }
}
template <bool kWeights = true>
__device__ __forceinline__ TexFootprint tex_footprint(const CamConsts& c, float x, float y) {
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fx = floorf(xb), fy = floorf(yb);
  TexFootprint f;
  f.a = f.ua = xb - fx;
  f.b = f.ub = yb - fy;
  if constexpr (kWeights) tex_weights(c, &f);
  // clamp = v_med3_f32 (same selection as fminf(fmaxf(.)) for the finite arguments that reach this point, without the
  // two NaN-quieting moves fminf / fmaxf cost each)
  f.i = (int)__builtin_amdgcn_fmed3f(fx, -1.0f, (float)(c.color_width - 1));
  f.j = (int)__builtin_amdgcn_fmed3f(fy, -1.0f, (float)(c.color_height - 1));
  // 0 <= xb < w - 1  <=>  0 <= floor(xb) <= w - 2 (w - 1 is an integer)  <=>  the clamped base texel lies in [0, w - 2]: two
  // unsigned compares instead of four float compares
  f.interior = ((uint32_t)f.i < (uint32_t)(c.color_width - 1)) & ((uint32_t)f.j < (uint32_t)(c.color_height - 1));
  return f;
}
__device__ __forceinline__ float tex_filter(const LumaQuad& t, float a, float b) {
  const float w00 = (1.0f - a) * (1.0f - b);
  const float w10 = a * (1.0f - b);
  const float w01 = (1.0f - a) * b;
  const float w11 = a * b;
  return ((w00 * t.tl + w10 * t.tr) + w01 * t.bl) + w11 * t.br;
}
// Sampling arithmetic of the RESIDUAL path (descriptor residual and its image gradient; colours and downsampled images, whose
// outputs are integers, keep tex_filter above):
//   - texels stay raw bytes 0..255 (exact in fp32, and so are their differences); the 1 / 255 normalisation is folded into the
//     residual's intensity scale, 180 * (i1 - i0) = (180 / 255) * (b1 - b0);
//   - the bilinear value is a nested interpolation with explicit fused multiply-adds (3 roundings instead of 11), the gradient
//     one fma per component (1 rounding each).
// The CPU checker evaluates the same sequences, so the two agree to the bit; the reference's formula as written is its
// "literal" mode.
constexpr float kDescScale = 180.f / 255.f;
__device__ __forceinline__ LumaQuad unpack_quad_bytes(uint32_t q) {
  LumaQuad r;
  r.tl = (float)(q & 0xffu);
  r.tr = (float)((q >> 8) & 0xffu);
  r.bl = (float)((q >> 16) & 0xffu);
  r.br = (float)(q >> 24);
  return r;
}
__device__ __forceinline__ float bilinear_bytes(const LumaQuad& t, float a, float b) {
  const float top = __builtin_fmaf(a, t.tr - t.tl, t.tl);
  const float bot = __builtin_fmaf(a, t.br - t.bl, t.bl);
  return __builtin_fmaf(b, bot - top, top);
}
__device__ __forceinline__ void bilinear_gradient_bytes(const LumaQuad& t, float tx, float ty, float* dx, float* dy) {
  *dx = __builtin_fmaf(ty, (t.br - t.bl) - (t.tr - t.tl), t.tr - t.tl);
  *dy = __builtin_fmaf(tx, (t.br - t.tr) - (t.bl - t.tl), t.bl - t.tl);
}
// bilinear luma in byte units at pixel-corner coordinates (x, y)
__device__ __forceinline__ float tex_b(const KfDev& kf, const CamConsts& c, float x, float y) {
  const TexFootprint f = tex_footprint(c, x, y);
  return bilinear_bytes(unpack_quad_bytes(quad_at(kf, c, f.i, f.j)), f.a, f.b);
}

// BS/cost_function.cuh:115-136
__device__ __forceinline__ f2 project_sample(float fx, float fy, float cx, float cy, f3 p) {
  const float inv_z = srcp(p.z);
  return f2{__builtin_fmaf(fx, p.x * inv_z, cx), __builtin_fmaf(fy, p.y * inv_z, cy)};
}
// The two tangent sample points gp + t1, gp + t2 depend on the surfel only; the per-pair work is their projection.
__device__ __forceinline__ void tangent_points(f3 gp, f3 gn, float radius_squared, f3* p1, f3* p2) {
  f3 t1 = cross(gn, (fabsf(gn.x) > 0.9f) ? mk3(0, 1, 0) : mk3(1, 0, 0));
  t1 = scale3(ssqrt(sdiv(radius_squared, fmaxf(1e-12f, sqlen(t1)))), scale3(2.0f, t1));
  *p1 = add3(gp, t1);
  f3 t2 = cross(gn, t1);
  t2 = scale3(ssqrt(sdiv(radius_squared, fmaxf(1e-12f, sqlen(t2)))), scale3(2.0f, t2));
  *p2 = add3(gp, t2);
}
__device__ __forceinline__ void project_tangent_points(f3 p1, f3 p2, const M34& T, const CamConsts& c, f2* t1_pxy, f2* t2_pxy) {
  *t1_pxy = project_sample(c.cfx, c.cfy, c.ccx, c.ccy, mul34(T, p1));
  *t2_pxy = project_sample(c.cfx, c.cfy, c.ccx, c.ccy, mul34(T, p2));
}
__device__ __forceinline__ void tangent_projections(f3 gp, f3 gn, float radius_squared, const M34& T, const CamConsts& c, f2* t1_pxy, f2* t2_pxy) {
  f3 p1, p2;
  tangent_points(gp, gn, radius_squared, &p1, &p2);
  project_tangent_points(p1, p2, T, c, t1_pxy, t2_pxy);
}

// BS/cost_function.cuh:140-156
__device__ __forceinline__ void raw_descriptor_residual(const KfDev& kf, const CamConsts& c, f2 pxy, f2 t1, f2 t2, float d1, float d2, float* r1, float* r2) {
  const float b0 = tex_b(kf, c, pxy.x, pxy.y);
  const float b1 = tex_b(kf, c, t1.x, t1.y);
  const float b2 = tex_b(kf, c, t2.x, t2.y);
  *r1 = __builtin_fmaf(kDescScale, b1 - b0, -d1);
  *r2 = __builtin_fmaf(kDescScale, b2 - b0, -d2);
}

// one block of BS/cost_function.cuh:200-239: base texel and weights of the gradient at p
struct GradFootprint { int ix, iy; float tx, ty; };
__device__ __forceinline__ GradFootprint grad_footprint(const CamConsts& c, f2 p) {
  GradFootprint g;
  const float big = 3.0e38f;
  g.ix = f2i(__builtin_amdgcn_fmed3f(p.x - 0.5f, 0.f, big));   // fmaxf(0, .)
  g.iy = f2i(__builtin_amdgcn_fmed3f(p.y - 0.5f, 0.f, big));
  g.tx = __builtin_amdgcn_fmed3f(p.x - 0.5f - (float)g.ix, 0.f, 1.f);
  g.ty = __builtin_amdgcn_fmed3f(p.y - 0.5f - (float)g.iy, 0.f, 1.f);
  g.ix = min(g.ix, c.color_width - 1);
  g.iy = min(g.iy, c.color_height - 1);
  return g;
}
// the reference's formula as written, on texels in [0, 1] (bslam_debug_jacobians compares it with the fma form above)
__device__ __forceinline__ void grad_filter(const LumaQuad& t, const GradFootprint& g, float* dx, float* dy) {
  *dx = (t.br - t.bl) * g.ty + (t.tr - t.tl) * (1 - g.ty);
  *dy = (t.br - t.tr) * g.tx + (t.bl - t.tl) * (1 - g.tx);
}

// raw_descriptor_residual + descriptor_jacobian_wrt_projected_position sharing their gathers, in two steps: the three
// footprints and the three gathers first (DescSamples: the loads are in flight when it returns), the filters later -- so
// that a kernel can issue the gathers of several surfels before it waits for the first.
// Inside the image the gradient's footprint (BS/cost_function.cuh:200-211: ix = int(max(0, x - 0.5)), tx = clamp(x - 0.5 - ix, 0, 1))
// IS the sample's: ix = floor(x - 0.5) and tx = the unquantised fraction, bit for bit (the same subtraction).  Only in the
// half-pixel border strip does a gradient read another quad than its sample; that rarely taken path asks the caller for the
// sample positions again (`sample_points`: a callable filling f2[3] = {centre, tangent point 1, tangent point 2}, recomputed
// from what the kernel holds anyway) instead of carrying a second set of footprints through every pair.
struct DescSamples {
  TexFootprint f[3];
  uint32_t q[3];
};
__device__ __forceinline__ DescSamples descriptor_samples_issue(const KfDev& kf, const CamConsts& c, f2 cp, f2 t1, f2 t2) {
  DescSamples d;
  const f2 pts[3] = {cp, t1, t2};
  // base texels first, the three gathers next, the filter weights while they are in flight (the asm pins the fractions behind the
  // loads): photometric geometry pass -1.5 %, PCG -1 %, pose kernel unchanged
#pragma unroll
  for (int k = 0; k < 3; ++k) d.f[k] = tex_footprint<false>(c, pts[k].x, pts[k].y);
#pragma unroll
  for (int k = 0; k < 3; ++k) d.q[k] = quad_at(kf, c, d.f[k].i, d.f[k].j);
  asm volatile("" : "+v"(d.f[0].ua), "+v"(d.f[0].ub), "+v"(d.f[1].ua), "+v"(d.f[1].ub), "+v"(d.f[2].ua), "+v"(d.f[2].ub) : : "memory");
#pragma unroll
  for (int k = 0; k < 3; ++k) tex_weights(c, &d.f[k]);
  return d;
}
template <class SamplePoints>
__device__ __forceinline__ void descriptor_samples_finish(const KfDev& kf, const CamConsts& c, const DescSamples& d, float d1, float d2, SamplePoints&& sample_points,
                                                          float* r1, float* r2, float* gx1, float* gy1, float* gx2, float* gy2) {
  float val[3], gx[3], gy[3];   // byte units
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const LumaQuad t = unpack_quad_bytes(d.q[k]);
    val[k] = bilinear_bytes(t, d.f[k].a, d.f[k].b);
    bilinear_gradient_bytes(t, d.f[k].ua, d.f[k].ub, &gx[k], &gy[k]);
  }
  // some sample's 2x2 footprint touches the image border: a base texel outside [0, w - 2] x [0, h - 2] (-1 is the largest unsigned
  // value), decided on the maxima of the three base texels (two v_max3_u32 and two compares instead of six compares)
  const uint32_t max_i = max(max((uint32_t)d.f[0].i, (uint32_t)d.f[1].i), (uint32_t)d.f[2].i);
  const uint32_t max_j = max(max((uint32_t)d.f[0].j, (uint32_t)d.f[1].j), (uint32_t)d.f[2].j);
  if ((max_i >= (uint32_t)(c.color_width - 1)) | (max_j >= (uint32_t)(c.color_height - 1))) {
    f2 pts[3];
    sample_points(pts);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const GradFootprint g = grad_footprint(c, pts[k]);
      bilinear_gradient_bytes(unpack_quad_bytes(quad_at(kf, c, g.ix, g.iy)), g.tx, g.ty, &gx[k], &gy[k]);
    }
  }
  *r1 = __builtin_fmaf(kDescScale, val[1] - val[0], -d1);
  *r2 = __builtin_fmaf(kDescScale, val[2] - val[0], -d2);
  *gx1 = kDescScale * (gx[1] - gx[0]);
  *gy1 = kDescScale * (gy[1] - gy[0]);
  *gx2 = kDescScale * (gx[2] - gx[0]);
  *gy2 = kDescScale * (gy[2] - gy[0]);
}
__device__ __forceinline__ void descriptor_residual_and_jacobian(const KfDev& kf, const CamConsts& c, f2 cp, f2 t1, f2 t2, float d1, float d2,
                                                                 float* r1, float* r2, float* gx1, float* gy1, float* gx2, float* gy2) {
  const DescSamples d = descriptor_samples_issue(kf, c, cp, t1, t2);
  descriptor_samples_finish(kf, c, d, d1, d2, [&](f2 (&pts)[3]) { pts[0] = cp; pts[1] = t1; pts[2] = t2; }, r1, r2, gx1, gy1, gx2, gy2);
}

// BS/surfel_projection.cuh:196-207
__device__ __forceinline__ bool depth_to_color_pxy(const CamConsts& c, f2 pxy, f2* out) {
  out->x = c.d2c_fx * pxy.x + c.d2c_cx;
  out->y = c.d2c_fy * pxy.y + c.d2c_cy;
  return out->x >= 0 && out->y >= 0 && f2i(out->x) < c.color_width && f2i(out->y) < c.color_height;
}
// The same for a pixel position that has already passed the depth image's bounds test (project_to_pixel).  When both cameras
// share intrinsics and size (every RGB-D dataset registered to the depth frame: 1 * x + 0 is x, bit for bit, for x >= 0) the
// map is the identity and the test is already known to pass: a uniform branch instead of ten VALU instructions per pair.
__device__ __forceinline__ bool depth_to_color_pxy_in_bounds(const CamConsts& c, f2 pxy, f2* out) {
  if (c.d2c_identity) { *out = pxy; return true; }
  return depth_to_color_pxy(c, pxy, out);
}

// ---------------------------------------------------------------------------------------------
// Projection + association (BS/surfel_projection_nvcc_only.cuh:49-127, 302-332; BS/util.cuh:67-99)
// ---------------------------------------------------------------------------------------------
struct Proj {
  f3 local;        // surfel position in the keyframe
  f3 n_local;      // surfel normal rotated into the keyframe
  float depth;     // calibrated depth of the pixel
  int px, py;
  float nx, ny;    // normalised image coordinates of the pixel centre (nx_of / ny_of), formed once per pair
  f2 pxy;
  f3 pixel_normal;        // decoded image-space normal of the associated pixel
};

// One derived record per pixel: the calibrated depth depends only on the pixel (raw depth, cfactor cell, a) and the decoded
// normal only on its u16 code, so the two IEEE divisions + expf of RawToCalibratedDepth (BS/util.cuh:46-53) and the sqrt of
// U16ToImageSpaceNormal (BS/util.cuh:107-130) are paid once per pixel per call instead of once per (surfel, keyframe) pair
// (~40 VALU instructions), and everything arrives in ONE 16-byte gather.  Same arithmetic, hence bit-identical values; a pixel
// without measurement has depth 0, which the association test rejects like the reference's invalid bit (a valid raw depth
// of 0 calibrates to 0 as well and fails the depth comparison either way).
__global__ __launch_bounds__(256) void build_records_kernel(CamConsts c, const KfDev* __restrict__ kfs, PixelRecord* __restrict__ records) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int k = blockIdx.z;
  if (x >= c.width) return;
  const KfDev& kf = kfs[k];
  const uint32_t measured = *(const uint16_t*)(kf.depth + (size_t)y * kf.depth_pitch + 2 * (size_t)x);
  const uint32_t normal = *(const uint16_t*)(kf.normals + (size_t)y * kf.normals_pitch + 2 * (size_t)x);
  float depth = 0.f;
  if (!(measured & BSLAM_INVALID_DEPTH_BIT)) {
    const float cf = *(const float*)((const uint8_t*)c.cfactor + (size_t)(y / c.cell) * c.cfactor_pitch + 4 * (size_t)(x / c.cell));
    depth = raw_to_calibrated_depth(c.a, cf, c.raw_to_float_depth, measured);
  }
  const f3 n = u16_to_image_space_normal(normal);
  records[((size_t)k * c.height + y) * c.width + x] = PixelRecord{depth, n.x, n.y, n.z};
}

// Luma quads of every keyframe's colour image (see KfDev::quads); grid (ceil((w+1)/256), h+1, K).
__global__ __launch_bounds__(256) void build_quads_kernel(CamConsts c, const KfDev* __restrict__ kfs, uint32_t* __restrict__ quads) {
  const int qx = blockIdx.x * blockDim.x + threadIdx.x;   // = i + 1
  const int qy = blockIdx.y;                              // = j + 1
  const int k = blockIdx.z;
  const int w = c.color_width, h = c.color_height;
  if (qx > w) return;
  const KfDev& kf = kfs[k];
  const int i0 = max(0, qx - 1), i1 = min(qx, w - 1);
  const int j0 = max(0, qy - 1), j1 = min(qy, h - 1);
  const uint8_t* r0 = kf.color + (size_t)j0 * kf.color_pitch;
  const uint8_t* r1 = kf.color + (size_t)j1 * kf.color_pitch;
  const uint32_t tl = r0[4 * (size_t)i0 + 3], tr = r0[4 * (size_t)i1 + 3];
  const uint32_t bl = r1[4 * (size_t)i0 + 3], br = r1[4 * (size_t)i1 + 3];
  quads[((size_t)k * (h + 1) + qy) * (size_t)(w + 1) + qx] = tl | (tr << 8) | (bl << 16) | (br << 24);
}

// Projection + association in three stages, so that a kernel can put the record gathers of several surfels in flight before
// it consumes the first (project_and_associate below is their composition; same arithmetic either way).
// Stage 1: MultiplyIfResultZIsPositive BS/cuda_matrix.cuh:113-124 + ProjectSurfelToImage BS/util.cuh:86-99.  True when the
// surfel projects into the image; fills local, pxy, px, py.
__device__ __forceinline__ bool project_to_pixel(const CamConsts& c, const KfDev& kf, f3 gp, Proj* r) {
  const M34& T = kf.frame_T_global;
  r->local.z = tr_row(T.m[8], T.m[9], T.m[10], T.m[11], gp);
  r->local.x = tr_row(T.m[0], T.m[1], T.m[2], T.m[3], gp);
  r->local.y = tr_row(T.m[4], T.m[5], T.m[6], T.m[7], gp);
  r->pxy = project(c.fx, c.fy, c.cx, c.cy, r->local);
  r->px = f2i(r->pxy.x);
  r->py = f2i(r->pxy.y);
  // one exit (with the per-surfel work order a wave's lanes almost always agree, so early outs only cost exec-mask
  // bookkeeping): a point behind the camera projects to garbage (finite or not) that the z test discards
  return (r->local.z > 0.f) & !(r->pxy.x < 0 || r->pxy.y < 0 || r->px >= c.width || r->py >= c.height);
}
// Stage 2: the pixel's derived record.  The offset inside one keyframe's table fits 24 bits (v_mad_u32_u24, full rate) and is
// added to the uniform base as a 32-bit offset (global_load with an SGPR base): no 64-bit vector arithmetic per gather.
__device__ __forceinline__ PixelRecord load_record(const CamConsts& c, const KfDev& kf, const Proj& r) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = gload_at((const v4f*)kf.records, __umul24((uint32_t)r.py, 16u * (uint32_t)c.width) + ((uint32_t)r.px << 4));   // < 2^28
  return PixelRecord{v.x, v.y, v.z, v.w};
}
// raw u16 depth of the associated pixel (only the depth-intrinsics Jacobians need it)
__device__ __forceinline__ uint32_t raw_depth_of(const KfDev& kf, const Proj& r) {
  return gload((const uint16_t*)(kf.depth + (size_t)r.py * kf.depth_pitch + 2 * (size_t)r.px));
}
// Stage 3: IsAssociatedWithPixel<false, true> BS/surfel_projection_nvcc_only.cuh:49-127 on the loaded record.
__device__ __forceinline__ bool associate_with_record(const CamConsts& c, const KfDev& kf, f3 gn, PixelRecord rec, Proj* r) {
  const M34& T = kf.frame_T_global;
  // all four tests evaluated, one branch: with the per-surfel work order a wave's lanes almost always agree, and 99.6 % of the
  // in-bounds pairs pass, so the early outs only cost exec-mask bookkeeping
  r->depth = rec.depth;
  r->n_local = rot34(T, gn);
  r->nx = nx_of(c, (float)r->px);
  r->ny = ny_of(c, (float)r->py);
  r->pixel_normal = mk3(rec.nx, rec.ny, rec.nz);
  const float sd = depth_stddev(r->nx, r->ny, r->depth, r->n_local, c.inv_baseline_fx);
  bool ok = rec.depth != 0.f;
  ok &= !(fabsf(r->local.z - r->depth) > kDepthTukey * sd);
  // reference: (1.0f / Norm(local)) * Dot(local, n_local) > 0  (:107-111).  1 / |local| is a positive, finite, normal number
  // for every |local| in [2^-126, 2^126] (and |local| >= local.z > 0 here), so the product has the sign of the dot product
  // unless it underflows, which needs |dot| < 2^-23 * 2^-126 * |local|: the sign test is evaluated on the dot product directly
  // (saves a sqrt and a division per pair).
  ok &= !(dot(r->local, r->n_local) > 0);
  ok &= !(dot(r->n_local, r->pixel_normal) < kCosNormalCompat);
  return ok;
}
// gp: global position, gn: unit global normal (already decoded).  Returns true when the surfel is
// associated with the pixel it projects to.
__device__ __forceinline__ bool project_and_associate(const CamConsts& c, const KfDev& kf, f3 gp, f3 gn, Proj* r) {
  if (!project_to_pixel(c, kf, gp, r)) return false;
  return associate_with_record(c, kf, gn, load_record(c, kf, *r), r);
}

// ---------------------------------------------------------------------------------------------
// Residuals and Jacobians: ONE definition per formula, used by every kernel (pose, geometry, PCG, intrinsics,
// odometry) and checked point by point against the reference's own symbolic derivation
// (applications/badslam/scripts/jacobians_derivation.py -> tests/golden/jacobian_golden.npz) through
// bslam_debug_jacobians (tests/test_gpu_jacobians.py); the CPU checker used by the tests holds the same set.
// ---------------------------------------------------------------------------------------------
// Depth residual (BS/cost_function.cuh:56-68): n_local = surfel normal in the frame, lu = unprojected pixel,
// ls = surfel position in the frame.
__device__ __forceinline__ float depth_residual(float inv_stddev, f3 n_local, f3 lu, f3 ls) { return inv_stddev * dot(n_local, sub3(lu, ls)); }
// ... and its Jacobian wrt. the pose delta (BS/kernel_opt_pose.cu:45-94)
__device__ __forceinline__ void depth_pose_jacobian(float inv_stddev, f3 n_local, f3 lu, float* J) {
#pragma clang fp contract(fast)
  J[0] = inv_stddev * n_local.x;
  J[1] = inv_stddev * n_local.y;
  J[2] = inv_stddev * n_local.z;
  J[3] = inv_stddev * (-n_local.y * lu.z + n_local.z * lu.y);
  J[4] = inv_stddev * (n_local.x * lu.z - n_local.z * lu.x);
  J[5] = inv_stddev * (-n_local.x * lu.y + n_local.y * lu.x);
}
// The pose kernel's form of the two: past the association test nothing feeds an integer output any more, so residual and
// Jacobian may use fused multiply-adds (as the reference's nvcc build does by default); the predicate path stays unfused.
__device__ __forceinline__ void depth_residual_and_pose_jacobian_fused(float inv_stddev, f3 n_local, f3 lu, f3 ls, float* raw, float* J) {
#pragma clang fp contract(fast)
  const f3 dl = mk3(lu.x - ls.x, lu.y - ls.y, lu.z - ls.z);
  *raw = inv_stddev * (n_local.x * dl.x + n_local.y * dl.y + n_local.z * dl.z);
  J[0] = inv_stddev * n_local.x;
  J[1] = inv_stddev * n_local.y;
  J[2] = inv_stddev * n_local.z;
  J[3] = inv_stddev * (-n_local.y * lu.z + n_local.z * lu.y);
  J[4] = inv_stddev * (n_local.x * lu.z - n_local.z * lu.x);
  J[5] = inv_stddev * (-n_local.x * lu.y + n_local.y * lu.x);
}
__device__ __forceinline__ f3 pixel_unprojection(const Proj& r) {   // unproject(c, r.px, r.py, r.depth) from the cached nx, ny
#pragma clang fp contract(fast)
  return mk3(r.depth * r.nx, r.depth * r.ny, r.depth);
}
__device__ __forceinline__ void depth_residual_and_jacobian(const CamConsts& c, const Proj& r, float* raw, float* J) {
  const float inv_stddev = depth_inv_stddev(r.nx, r.ny, r.depth, r.n_local, c.baseline_fx);
  depth_residual_and_pose_jacobian_fused(inv_stddev, r.n_local, pixel_unprojection(r), r.local, raw, J);
}
// Depth residual wrt. a surfel move of t along its (unit) normal (BS/kernel_opt_geometry.cu:440, BS/kernel_pcg.cu:217).
__device__ __forceinline__ float depth_position_jacobian(float inv_stddev) { return -inv_stddev; }

// Descriptor residual wrt. the pose delta (BS/kernel_opt_pose.cu:122-141); gx, gy = image gradient of the residual times the
// colour camera's fx, fy; ls = surfel position in the frame.  kExact: correctly rounded reciprocal (odometry, as the oracle).
template <bool kExact = false>
__device__ __forceinline__ void descriptor_pose_jacobian(float gx, float gy, f3 ls, float* J) {
  if constexpr (kExact) {
    const float inv_ls_z = 1.f / ls.z;
    const float ls_z_sq = ls.z * ls.z;
    const float inv_ls_z_sq = inv_ls_z * inv_ls_z;
    J[0] = -gx * inv_ls_z;
    J[1] = -gy * inv_ls_z;
    J[2] = (ls.x * gx + ls.y * gy) * inv_ls_z_sq;
    const float ls_x_y = ls.x * ls.y;
    J[3] = ((ls.y * ls.y + ls_z_sq) * gy + ls_x_y * gx) * inv_ls_z_sq;
    J[4] = -((ls.x * ls.x + ls_z_sq) * gx + ls_x_y * gy) * inv_ls_z_sq;
    J[5] = -(ls.x * gy - ls.y * gx) * inv_ls_z;
  } else {
#pragma clang fp contract(fast)   // the BA kernels' form: sums of products fuse (6 instructions less per residual), as under nvcc's default
    const float inv_ls_z = rrcp(ls.z);
    const float ls_z_sq = ls.z * ls.z;
    const float inv_ls_z_sq = inv_ls_z * inv_ls_z;
    J[0] = -gx * inv_ls_z;
    J[1] = -gy * inv_ls_z;
    J[2] = (ls.x * gx + ls.y * gy) * inv_ls_z_sq;
    const float ls_x_y = ls.x * ls.y;
    J[3] = ((ls.y * ls.y + ls_z_sq) * gy + ls_x_y * gx) * inv_ls_z_sq;
    J[4] = -((ls.x * ls.x + ls_z_sq) * gx + ls_x_y * gy) * inv_ls_z_sq;
    J[5] = -(ls.x * gy - ls.y * gx) * inv_ls_z;
  }
}
// Descriptor residual wrt. a surfel move along its normal (BS/kernel_opt_geometry.cu:175-189, BS/kernel_pcg.cu:364-372):
// rn = normal in the frame, ls = position in the frame; the gradient (gx, gy) is multiplied by (fx, fy) here (the PCG kernels
// pass gradients that already carry the focal lengths and fx = fy = 1).
__device__ __forceinline__ float descriptor_position_jacobian(float gx, float gy, float fx, float fy, f3 rn, f3 ls) {
  // (not contracted: rn.x * ls.z - rn.z * ls.x cancels, and this single number is held to the CPU checker's at 2e-6)
  const float term1 = -fx * (rn.x * ls.z - rn.z * ls.x);
  const float term2 = -fy * (rn.y * ls.z - rn.z * ls.y);
  const float term3 = rrcp(ls.z * ls.z);
  return -(gx * term1 + gy * term2) * term3;
}
// Depth residual wrt. (fx_inv, fy_inv, cx_inv, cy_inv, a, cfactor of the pixel's cell), dj[0..5]
// (BS/kernel_opt_intrinsics.cu:82-118 = BS/kernel_pcg.cu:258-322).  m = frame_T_global, ln = normal in the frame.  Returns
// corrected_inv_depth, on which the callers base their validity tests.
__device__ __forceinline__ float depth_intrinsics_jacobian(float inv_stddev, float calibrated_depth, int px, int py, float nx, float ny, f3 n_global,
                                                          const float* m, f3 ln, float cfactor, float a, float raw_inv_depth, float* dj) {
  const float exp_inv_depth = det_expf(-a * raw_inv_depth);
  const float corrected_inv_depth = cfactor * exp_inv_depth + raw_inv_depth;
  const float dt = dot(mk3(nx, ny, 1), ln);
  const float jac_base = inv_stddev * dt * exp_inv_depth / (corrected_inv_depth * corrected_inv_depth);
  dj[2] = inv_stddev * calibrated_depth * dot(n_global, mk3(m[0], m[1], m[2]));
  dj[3] = inv_stddev * calibrated_depth * dot(n_global, mk3(m[4], m[5], m[6]));
  dj[0] = (float)px * dj[2];
  dj[1] = (float)py * dj[3];
  dj[4] = cfactor * raw_inv_depth * jac_base;
  dj[5] = -jac_base;
  return corrected_inv_depth;
}
// Descriptor residual wrt. the colour camera's (fx, fy, cx, cy) (BS/kernel_opt_intrinsics.cu:141-149, BS/kernel_pcg.cu:462-509):
// gx, gy = image gradient of the residual (no focal length), nx, ny = normalised image coordinates of the pixel.
__device__ __forceinline__ void color_intrinsics_jacobian(float gx, float gy, float nx, float ny, float* j) {
  j[0] = gx * nx;
  j[1] = gy * ny;
  j[2] = gx;
  j[3] = gy;
}

// An accumulator zeroed by its own opaque instruction.  Written as `x = 0.f` the optimiser knows all accumulators of a set to be one
// value: it folds the first fma(a, b, 0) into a multiply and then materialises the zeros a second time, as copies, for the
// lanes that skip that term (pose kernel: 62 v_mov per keyframe and thread instead of 27).
#ifndef BSLAM_INDEPENDENT_ZEROS
#define BSLAM_INDEPENDENT_ZEROS 1
#endif
#if BSLAM_INDEPENDENT_ZEROS
#define BSLAM_ZERO(x) asm volatile("v_mov_b32 %0, 0" : "=v"(x))
#else
#define BSLAM_ZERO(x) ((x) = 0.f)
#endif

// A wave-uniform value copied into a vector register by an opaque instruction.  A VOP3 instruction reads at most ONE scalar
// operand on gfx950, so fma(s_a, v, s_b) costs a v_mov of s_b in front of every such fma -- the compiler re-materialises the copy
// at each use rather than keep it live; the opaque copy is made once and stays.
#define BSLAM_TO_VGPR(x) do { float v_; asm volatile("v_mov_b32 %0, %1" : "=v"(v_) : "s"(x)); (x) = v_; } while (0)
#define BSLAM_HOIST_KF_TRANSLATION(kf) do { BSLAM_TO_VGPR((kf).frame_T_global.m[3]); BSLAM_TO_VGPR((kf).frame_T_global.m[7]); BSLAM_TO_VGPR((kf).frame_T_global.m[11]); } while (0)
#define BSLAM_HOIST_CAM_CENTRES(c) do { BSLAM_TO_VGPR((c).cx); BSLAM_TO_VGPR((c).cy); BSLAM_TO_VGPR((c).ccx); BSLAM_TO_VGPR((c).ccy); } while (0)
#define BSLAM_HOIST_UNPROJECTION_CENTRE(c) do { BSLAM_TO_VGPR((c).cx_inv); BSLAM_TO_VGPR((c).cy_inv); } while (0)   // nx_of / ny_of
#define BSLAM_HOIST_DEPTH_CAM_CENTRE(c) do { BSLAM_TO_VGPR((c).cx); BSLAM_TO_VGPR((c).cy); } while (0)   // kernels that never sample the colour image

// ---------------------------------------------------------------------------------------------
// Work schedule.  Surfels are handled in granules of 256 consecutive columns, in the order of a Morton curve (per surfel for
// long keyframe lists, per granule for short ones: badslam_hip.hip make_schedule), so that a workgroup's surfels are a compact
// blob; workgroups are dealt round-robin over the 8 XCDs (observed dispatch behaviour, used for speed only) and take the work
// slots in that order (slot_of_block).  Any permutation is correct; a stale one only costs locality.
// ---------------------------------------------------------------------------------------------
constexpr int kGranule = 256;

struct Schedule {
  const uint32_t* order;   // sorted position -> granule id (nullptr: identity)
  uint32_t granules;       // ceil(S / 256)
  uint32_t slots;          // ceil(granules / R): work items of R granules each
  uint32_t slots_per_xcd;  // ceil(slots / 8)
  const float4* bounds;    // per-granule bounding boxes of the sorted copy (block-level frustum culling), or nullptr: no culling
};

// Work slot of block index b (of a 1-D launch of 8 * slots_per_xcd blocks), or false.  Block b takes slot b: workgroups are
// dealt round-robin over the 8 XCDs, so consecutive slots -- neighbours in the Morton order -- go to different XCDs and every
// XCD gets an even share of every part of the scene.  (Rounds 1 - 3 gave each XCD one CONTIGUOUS eighth of the order, for the
// sake of its L2; the work of those eighths differs -- parts of the scene are seen by more keyframes than others, a short list of
// keyframes sees one or two of them only -- and a launch lasted as long as the busiest XCD: dense K = 300 photometric pose
// kernel 12.5 -> 10.2 ms, pcg_step1_kernel 13.8 -> 11.3 ms, survey-range stack 3.9 -> 2.75 ms per launch, same box.)
__device__ __forceinline__ bool slot_of_block(const Schedule& sc, uint32_t b, uint32_t* slot) {
  if ((b >> 3) >= sc.slots_per_xcd || b >= sc.slots) return false;
  *slot = b;
  return true;
}

// Surfel column of this thread for granule r of the slot (R granules per slot); >= size if none.
__device__ __forceinline__ uint32_t surfel_of_slot(const Schedule& sc, uint32_t slot, int r, int R) {
  const uint32_t pos = slot * (uint32_t)R + (uint32_t)r;
  if (pos >= sc.granules) return 0xffffffffu;
  const uint32_t g = sc.order ? sc.order[pos] : pos;
  return g * kGranule + threadIdx.x;
}

// ---------------------------------------------------------------------------------------------
// Block-level frustum culling.  With the per-surfel Morton order the surfels of a work slot (R granules of 256 consecutive
// positions of the sorted copy) are a compact blob; prepare_surfels stores the axis-aligned bounding box of every granule of
// that copy.  Before it walks its keyframes, a wave decides for 64 keyframes at once -- one keyframe per lane -- whether ANY
// point of the slot's box can pass project_to_pixel (z > 0 and the four image bounds); keyframes for which none can are not
// visited at all.  The reference spends a thread on every (surfel, keyframe) pair (BS/kernel_opt_pose.cu:263-275).
//
// The test must be exactly conservative: a skipped (slot, keyframe) may contain no pair that passes project_to_pixel as the
// kernels evaluate it in fp32.  Every inequality below is therefore relaxed by kCullSlack times the sum of the MAGNITUDES of
// the terms it is made of (not of their possibly cancelling sum): the kernels' own rounding of local = T * p and of the
// projection is bounded by a few 2^-24 of the same magnitudes, i.e. by < 1e-6 of them, and so is the rounding of the test
// itself; 1e-4 leaves two orders of magnitude and costs a band of < 1 px around the image.  NaN or infinite boxes (deleted
// surfels carry x = NaN) make every comparison false: not culled.
// ---------------------------------------------------------------------------------------------
constexpr float kCullSlack = 1e-4f;
struct SlotBox { f3 c, e; };   // centre and half extents in global coordinates

// Box of work slot `slot` (R granules) from the per-granule boxes {min.xyz, -}, {max.xyz, -}.
__device__ __forceinline__ SlotBox slot_box(const float4* __restrict__ bounds, uint32_t granules, uint32_t slot, int R) {
  const float inf = __uint_as_float(0x7f800000u);
  f3 lo = mk3(inf, inf, inf), hi = mk3(-inf, -inf, -inf);
  for (int r = 0; r < R; ++r) {
    const uint32_t pos = slot * (uint32_t)R + (uint32_t)r;
    if (pos >= granules) break;
    const float4 a = bounds[2 * (size_t)pos], b = bounds[2 * (size_t)pos + 1];
    lo = mk3(fminf(lo.x, a.x), fminf(lo.y, a.y), fminf(lo.z, a.z));
    hi = mk3(fmaxf(hi.x, b.x), fmaxf(hi.y, b.y), fmaxf(hi.z, b.z));
  }
  SlotBox o;
  o.c = mk3(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z));
  o.e = mk3(0.5f * (hi.x - lo.x), 0.5f * (hi.y - lo.y), 0.5f * (hi.z - lo.z));
  return o;
}

// True when no point of the box can pass project_to_pixel under frame_T_global = T (12 floats, row-major 3x4).
__device__ __forceinline__ bool box_outside_frustum(const CamConsts& c, const float* T, const SlotBox& b) {
  float cc[3], ee[3], mm[3];   // camera-space centre, half extents of the enclosing camera-aligned box, magnitude of the terms
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float a0 = T[4 * i], a1 = T[4 * i + 1], a2 = T[4 * i + 2], a3 = T[4 * i + 3];
    cc[i] = a0 * b.c.x + a1 * b.c.y + a2 * b.c.z + a3;
    ee[i] = fabsf(a0) * b.e.x + fabsf(a1) * b.e.y + fabsf(a2) * b.e.z;
    mm[i] = fabsf(a0) * fabsf(b.c.x) + fabsf(a1) * fabsf(b.c.y) + fabsf(a2) * fabsf(b.c.z) + fabsf(a3) + ee[i];
  }
  const float w = (float)c.width, h = (float)c.height;
  // local.z <= 0 for every point
  bool out = cc[2] + ee[2] < -kCullSlack * mm[2];
  // pxy.x < 0 for every point with z > 0:   fx x + cx z < 0
  out |= (c.fx * cc[0] + c.cx * cc[2]) + (c.fx * ee[0] + fabsf(c.cx) * ee[2]) < -kCullSlack * (c.fx * mm[0] + fabsf(c.cx) * mm[2]);
  // pxy.x >= width for every point with z > 0:   fx x + (cx - w) z >= 0
  out |= (c.fx * cc[0] + (c.cx - w) * cc[2]) - (c.fx * ee[0] + fabsf(c.cx - w) * ee[2]) > kCullSlack * (c.fx * mm[0] + fabsf(c.cx - w) * mm[2]);
  // the same for y
  out |= (c.fy * cc[1] + c.cy * cc[2]) + (c.fy * ee[1] + fabsf(c.cy) * ee[2]) < -kCullSlack * (c.fy * mm[1] + fabsf(c.cy) * mm[2]);
  out |= (c.fy * cc[1] + (c.cy - h) * cc[2]) - (c.fy * ee[1] + fabsf(c.cy - h) * ee[2]) > kCullSlack * (c.fy * mm[1] + fabsf(c.cy - h) * mm[2]);
  return out;
}

// Keyframes k0 + lane, lane = 0 .. 63, below k_end: bit `lane` of the result is set when the keyframe has to be visited
// (`wanted` is the caller's own per-lane condition -- not converged, not inactive; sc.bounds == nullptr means no culling).
// Uniform across the wave.  The slot's box is rebuilt from the granule boxes on every call (a few uniform loads per 64
// keyframes) rather than kept live across the caller's keyframe loop, where it would cost registers.
// lane_kf: the keyframe of this lane where the caller walks a LIST of keyframes (place k0 + lane of the list), else -1: keyframe
// k0 + lane itself.
__device__ __forceinline__ unsigned long long keyframes_to_visit(const CamConsts& c, const KfDev* __restrict__ kfs, int k0, int k_end, const Schedule& sc,
                                                                 uint32_t slot, int R, bool wanted, int lane_kf = -1) {
  const int place = k0 + (int)(threadIdx.x & 63u);
  const int k = lane_kf >= 0 ? lane_kf : place;
  bool visit = place < k_end && wanted;
  if (sc.bounds != nullptr) {
    uint32_t s_ = slot;
    asm volatile("" : "+s"(s_));   // keeps the box's loads inside the caller's loop over keyframe batches
    const SlotBox box = slot_box(sc.bounds, sc.granules, s_, R);
    if (visit) {
      float T[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) T[i] = kfs[k].frame_T_global.m[i];
      visit = !box_outside_frustum(c, T, box);
    }
  }
  return __ballot(visit);
}

// Walks the keyframes of [k_begin, k_end) that the slot's box reaches, in list order (64 keyframes are decided at a time, one
// per lane): inside the body, `K_` is the keyframe's index.  Needs c, kfs, sc and slot in scope; SLOT_R_ = granules per work slot.  WANTED_ is the caller's
// per-lane condition on keyframe k0_ + lane (evaluated only for lanes below k_end by the helpers below).
__device__ __forceinline__ bool keyframe_not_inactive(const KfDev* __restrict__ kfs, int k0, int k_end) {
  const int k = k0 + (int)(threadIdx.x & 63u);
  return k < k_end && kfs[k].activation != BSLAM_KF_INACTIVE;
}
__device__ __forceinline__ bool keyframe_active(const KfDev* __restrict__ kfs, int k0, int k_end) {
  const int k = k0 + (int)(threadIdx.x & 63u);
  return k < k_end && kfs[k].activation == BSLAM_KF_ACTIVE;
}
#define BSLAM_FOR_VISITED_KEYFRAMES_IF(K_, k_begin, k_end, SLOT_R_, WANTED_)                                                           \
  for (int k0_ = (k_begin); k0_ < (k_end); k0_ += 64)                                                                          \
    for (unsigned long long todo_ = keyframes_to_visit(c, kfs, k0_, (k_end), sc, slot, (SLOT_R_), (WANTED_));                  \
         todo_ != 0; todo_ &= todo_ - 1)                                                                                       \
      if (const int K_ = k0_ + __builtin_ctzll(todo_); true)
// ... those that are not INACTIVE (the geometry passes, BS/kernel_opt_geometry.cc:115-118)
#define BSLAM_FOR_VISITED_KEYFRAMES(K_, k_begin, k_end, SLOT_R_) BSLAM_FOR_VISITED_KEYFRAMES_IF(K_, k_begin, k_end, SLOT_R_, keyframe_not_inactive(kfs, k0_, (k_end)))

// ---------------------------------------------------------------------------------------------
// wave64 reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;   // valid in lane 0
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sums 32 per-lane values across the 64 lanes of a wave with 31 exchanges instead of 32 x 6:
// at every step a lane keeps one half of its values and trades the other half with a partner lane, so
// the value count halves while the number of lanes summed doubles.  Afterwards lane l holds the
// wave total of column (l >> 1) & 31 (both lanes of a pair hold it).  Fixed tree: deterministic.
// The exchanges across rows of 16 lanes go through ds_bpermute; those inside a row are DPP operands of
// the add (row_ror:8, row_half_mirror, quad_perm).  gfx950's v_permlane32_swap / v_permlane16_swap for
// the first two steps (85 instead of 190 instructions) measured the same kernel time, so the builtin-only
// form stays.
__device__ __forceinline__ float dpp_add_ror8(float keep, float send) {
  return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_add_half_mirror(float keep, float send) {
  return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x141, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_add_xor2(float keep, float send) {   // quad_perm:[2,3,0,1]
  return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x4e, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_add_xor1(float keep, float send) {   // quad_perm:[1,0,3,2]
  return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xb1, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_transpose_sum32(float (&v)[32]) {
  const uint32_t lane = threadIdx.x & 63u;
#define BSLAM_TR_STEP(N, MASK)                                           \
  {                                                                      \
    const bool up = (lane & MASK) != 0;                                  \
    _Pragma("unroll") for (int i = 0; i < N; ++i) {                      \
      const float send = up ? v[i] : v[i + N];                           \
      const float keep = up ? v[i + N] : v[i];                           \
      v[i] = keep + __shfl_xor(send, MASK, 64);                          \
    }                                                                    \
  }
  BSLAM_TR_STEP(16, 32)
  BSLAM_TR_STEP(8, 16)
#undef BSLAM_TR_STEP
  {
    const bool up = (lane & 8u) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = dpp_add_ror8(up ? v[i + 4] : v[i], up ? v[i] : v[i + 4]);
  }
  {
    const bool up = (lane & 4u) != 0;    // partner = lane ^ 7 inside each group of 8
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] = dpp_add_half_mirror(up ? v[i + 2] : v[i], up ? v[i] : v[i + 2]);
  }
  {
    const bool up = (lane & 2u) != 0;
    v[0] = dpp_add_xor2(up ? v[1] : v[0], up ? v[0] : v[1]);
  }
  return dpp_add_xor1(v[0], v[0]);
}

// The same sums through LDS, in two stages.  Stage 1, kCols columns per round: the wave writes them into a [kCols][64] tile
// (ds_write_b32 per column, conflict-free) and every lane reads back kCols neighbouring lanes' values of ONE column (ds_read_b128s:
// with L = 64 / kCols lanes per column, lane (g, i) = (lane / L, lane % L) reads entries kCols i .. kCols i + kCols - 1 of column
// g) and adds them.  Stage 2, once per FOUR rounds: the four partial sums of a lane go through the tile a second time -- one more
// ds_read_b128 and three adds -- and only then come the DPP adds, among the 16 / kCols lanes that are left per column.
// kCols = 4: 3 adds per round + (3 adds + 2 DPP adds) per four rounds = 31 VALU instructions for 27 columns (round 3's first
// form finished every round with log2(L) DPP adds and a select: 56; the transposing butterfly before it: ~190); kCols = 8:
// 7 per round + (3 + 1) = 32.  The wave's own LDS operations execute in order, so no barrier is needed and the tile is private to
// the wave.  Afterwards every lane returns the wave total of the column wave_column_sums_owner names (0 for a column >= kLive).  Fixed order:
// deterministic.
// Where the totals end up: lane -> (column it holds, whether it is the lane that stores that column).  With one batch of four
// rounds (<= 4 kCols columns) 16 / kCols lanes share a column; with two batches the first lane of a group holds the column of
// the first batch and the second lane that of the second.
template <int kLive, int kCols>
__device__ __forceinline__ void wave_column_sums_owner(int* col, bool* writer) {
  constexpr int kBatches = ((kLive + kCols - 1) / kCols + 3) / 4;
  constexpr uint32_t kShare = 16u / kCols;   // lanes per column after the second stage
  static_assert(kBatches == 1 || (kBatches == 2 && kShare >= 2), "two batches need two lanes per column");
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t sub = lane % kShare;
  *col = (int)(lane / kShare) + ((kBatches == 2 && sub == 1) ? 4 * kCols : 0);
  *writer = sub < (uint32_t)kBatches && *col < 32;
}
template <int kLive, int kCols, int N>
__device__ __forceinline__ float wave_column_sums_lds(const float (&v)[N], float* __restrict__ tile) {
  static_assert(kLive >= 1 && kLive <= N && N <= 32, "at most 32 columns");
  static_assert(kCols == 4 || kCols == 8, "4 or 8 columns per round");
  constexpr uint32_t L = 64 / kCols;
  constexpr int kRounds = (kLive + kCols - 1) / kCols;
  constexpr int kBatches = (kRounds + 3) / 4;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t g = lane / L, i = lane % L;
  typedef float v4f __attribute__((ext_vector_type(4)));
  float total[kBatches];
#pragma unroll
  for (int b = 0; b < kBatches; ++b) {
    float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = 4 * b + rr;
      if (r >= kRounds) break;
#pragma unroll
      for (int q = 0; q < kCols; ++q)
        if (kCols * r + q < kLive) tile[q * 64 + lane] = v[kCols * r + q];
      __builtin_amdgcn_wave_barrier();
      const v4f x = *reinterpret_cast<const v4f*>(tile + g * 64 + i * kCols);
      p[rr] = (x.x + x.y) + (x.z + x.w);
      if constexpr (kCols == 8) {
        const v4f y = *reinterpret_cast<const v4f*>(tile + g * 64 + i * kCols + 4);
        p[rr] = p[rr] + ((y.x + y.y) + (y.z + y.w));
      }
      __builtin_amdgcn_wave_barrier();
    }
    // stage 2: p[rr] of lane (g, i) is the partial sum of column kCols (4 b + rr) + g over the lanes kCols i .. kCols i + kCols - 1
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      if (4 * b + rr < kRounds) tile[rr * 64 + lane] = p[rr];
    __builtin_amdgcn_wave_barrier();
    // lane (rr2, g2, i2) = (lane / 16, (lane % 16) / (16 / kCols), lane % (16 / kCols)) adds four neighbouring partial sums of
    // column kCols (4 b + rr2) + g2 = 4 kCols b + lane / (16 / kCols); a row of the tile that this batch did not write (a last
    // batch of fewer than four rounds) only reaches columns >= kLive, which return 0 below
    const v4f z = *reinterpret_cast<const v4f*>(tile + (lane / 16u) * 64 + ((lane % 16u) / (16u / kCols)) * L + 4 * (lane % (16u / kCols)));
    __builtin_amdgcn_wave_barrier();
    float t = (z.x + z.y) + (z.z + z.w);
    t = dpp_add_xor1(t, t);
    if constexpr (kCols == 4) t = dpp_add_xor2(t, t);
    total[b] = t;
  }
  int col;
  bool writer;
  wave_column_sums_owner<kLive, kCols>(&col, &writer);
  float t = total[0];
  if constexpr (kBatches == 2) t = (lane % (16u / kCols) == 1) ? total[1] : total[0];
  return (col < kLive) ? t : 0.f;
}

// Generic form for N = 8 or 16 values: N - 1 exchanges down to one value per lane, then log2(64 / N)
// butterfly steps.  Afterwards every lane holds the wave total of column (lane / (64 / N)) % N.
template <int N>
__device__ __forceinline__ float wave_transpose_sum(float (&v)[N]) {
  static_assert(N == 8 || N == 16, "N must be 8 or 16");
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t mask = 32;
#pragma unroll
  for (int n = N / 2; n >= 1; n >>= 1, mask >>= 1) {
    const bool up = (lane & mask) != 0;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const float send = up ? v[i] : v[i + n];
      const float keep = up ? v[i + n] : v[i];
      v[i] = keep + __shfl_xor(send, mask, 64);
    }
  }
  float total = v[0];
#pragma unroll
  for (uint32_t m = 32 / N; m >= 1; m >>= 1) total += __shfl_xor(total, m, 64);
  return total;
}

}  // namespace bslam
