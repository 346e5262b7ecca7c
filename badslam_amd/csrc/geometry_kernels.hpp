// geometry_kernels.hpp -- surfel activation, normal / position / descriptor updates (gfx950).
//
// Replaces the kernels of BS/kernel_surfel_activation.cu and BS/kernel_opt_geometry.cu.
// The reference launches one kernel per keyframe and read-modify-writes the accumulator rows
// 8-16 of the surfel SoA on every launch (32 / 16 / 64 B of avoidable traffic per pair).  Here a
// thread owns one surfel, keeps it and its accumulators in registers and walks the keyframe
// table in list order -- the same order in which the reference's serialised launches add their
// terms, so the fp32 sums are formed in the reference's order -- and writes the surfel back
// once.  Scratch rows 8-16 are never touched.
#pragma once

#include "device_math.hpp"

namespace bslam {

// Rows the surfel kernels read and update.  With a per-surfel work order (`perm`, make_schedule) x .. d2 are the library's
// sorted copy of the caller's rows -- position j of the copy is the caller's column perm[j] -- and every update is written to
// both: the copy (later passes of the same call read it) and the caller's rows o*.  `active` is always the caller's.
struct SurfelRowsRW {
  float* x; float* y; float* z;
  uint32_t* normal;
  const float* radius_squared;
  float* d1; float* d2;
  uint8_t* active;
  uint32_t size;
  const uint32_t* perm;    // nullptr: x .. d2 ARE the caller's rows
  float* ox; float* oy; float* oz;
  uint32_t* onormal;
  float* od1; float* od2;
};
__device__ __forceinline__ uint32_t column_of(const SurfelRowsRW& s, uint32_t j) { return s.perm ? s.perm[j] : j; }
__device__ __forceinline__ void store_normal(const SurfelRowsRW& s, uint32_t j, uint32_t packed) {
  s.normal[j] = packed;
  if (s.perm) s.onormal[s.perm[j]] = packed;
}
__device__ __forceinline__ void store_position(const SurfelRowsRW& s, uint32_t j, f3 p) {
  s.x[j] = p.x; s.y[j] = p.y; s.z[j] = p.z;
  if (s.perm) { const uint32_t o = s.perm[j]; s.ox[o] = p.x; s.oy[o] = p.y; s.oz[o] = p.z; }
}
__device__ __forceinline__ void store_descriptor1(const SurfelRowsRW& s, uint32_t j, float v) { s.d1[j] = v; if (s.perm) s.od1[s.perm[j]] = v; }
__device__ __forceinline__ void store_descriptor2(const SurfelRowsRW& s, uint32_t j, float v) { s.d2[j] = v; if (s.perm) s.od2[s.perm[j]] = v; }

// Sorted copy of the seven persistent surfel rows the pair kernels read: out row r, position j = in row r, column perm[j].
// kRows = 4: position + normal only (what the geometry-only kernels read).  A block is one granule of the copy: it also
// stores the granule's axis-aligned bounding box (bounds[2 g] = min, bounds[2 g + 1] = max; NaN coordinates -- deleted
// surfels -- are ignored by fminf / fmaxf), the input of the block-level frustum culling (device_math.hpp).
template <int kRows>
__global__ __launch_bounds__(256) void permute_surfel_rows_kernel(const uint32_t* __restrict__ perm, uint32_t size, const float* __restrict__ in, size_t in_pitch_floats,
                                                                 float* __restrict__ out, size_t out_pitch_floats, float4* __restrict__ bounds) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const float inf = __uint_as_float(0x7f800000u);
  float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
  if (j < size) {
    const uint32_t i = perm[j];
    const int rows[7] = {BSLAM_SURFEL_X, BSLAM_SURFEL_Y, BSLAM_SURFEL_Z, BSLAM_SURFEL_NORMAL, BSLAM_SURFEL_RADIUS_SQUARED, BSLAM_SURFEL_DESCRIPTOR1, BSLAM_SURFEL_DESCRIPTOR2};
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const float v = in[(size_t)rows[r] * in_pitch_floats + i];
      out[(size_t)r * out_pitch_floats + j] = v;
      if (r < 3) { lo[r] = v; hi[r] = v; }
    }
  }
  if (bounds == nullptr) return;
  __shared__ float sm[4][6];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { sm[threadIdx.x >> 6][a] = lo[a]; sm[threadIdx.x >> 6][3 + a] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fminf(fminf(sm[0][a], sm[1][a]), fminf(sm[2][a], sm[3][a]));
      hi[a] = fmaxf(fmaxf(sm[0][3 + a], sm[1][3 + a]), fmaxf(sm[2][3 + a], sm[3][3 + a]));
    }
    bounds[2 * (size_t)blockIdx.x] = make_float4(lo[0], lo[1], lo[2], 0.f);
    bounds[2 * (size_t)blockIdx.x + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
  }
}

// SetSurfelInactiveKernel + K x DetermineActiveSurfelsKernel (BS/kernel_surfel_activation.cu:38-79)
// kCount (bslam_profile_enable): additionally adds the number of (surfel, keyframe) pairs actually visited -- the walk stops at
// the first associated active keyframe -- and the number of surfels set active to this block's two slots of `counters`
// (counters[2 * blockIdx.x + {0, 1}]; no atomics: a single contended address would serialise 1e4..1e5 waves): what bench.py
// credits this pass with.
template <bool kCount>
__global__ __launch_bounds__(256) void activation_kernel(CamConsts c, const KfDev* __restrict__ kfs, int kf_count, Schedule sc, SurfelRowsRW s,
                                                         unsigned long long* __restrict__ counters) {
  uint32_t slot;
  if (!slot_of_block(sc, blockIdx.x, &slot)) return;
  const uint32_t i = surfel_of_slot(sc, slot, 0, 1);
  uint32_t visited = 0, activated = 0;
  const bool valid = i < s.size;
  const uint32_t j = valid ? i : 0;
  const uint32_t col = column_of(s, j);
  uint8_t flag = valid ? (s.active[col] & (uint8_t)~BSLAM_SURFEL_ACTIVE_FLAG) : 0;
  const f3 gp = mk3(s.x[j], s.y[j], s.z[j]);
  const f3 gn = unpack_normal(s.normal[j]);
  bool searching = valid;   // until the surfel's first associated ACTIVE keyframe
  for (int k0 = 0; k0 < kf_count && __any(searching); k0 += 64) {   // uniform: 64 keyframes are decided at a time, one per lane
    for (unsigned long long todo = keyframes_to_visit(c, kfs, k0, kf_count, sc, slot, 1, keyframe_active(kfs, k0, kf_count)); todo != 0 && __any(searching); todo &= todo - 1) {
      if (!searching) continue;
      const int k = k0 + __builtin_ctzll(todo);
      if (kCount) visited += 1;
      Proj p;
      if (project_and_associate(c, kfs[k], gp, gn, &p)) { flag = BSLAM_SURFEL_ACTIVE_FLAG; activated = 1; searching = false; }
    }
  }
  if (valid) s.active[col] = flag;
  if (kCount) {
    __shared__ uint32_t sm[2][4];
    visited = wave_sum_u32(visited);
    activated = wave_sum_u32(activated);
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = visited; sm[1][threadIdx.x >> 6] = activated; }
    __syncthreads();
    if (threadIdx.x < 2) counters[2 * (size_t)blockIdx.x + threadIdx.x] += (unsigned long long)(sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3]);
  }
}

// Mean position of every granule of 256 surfels (NaN / deleted surfels ignored): input of the
// host-side Morton sort that builds the XCD-aware schedule.
__global__ __launch_bounds__(256) void granule_centroid_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
                                                               uint32_t size, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * kGranule + threadIdx.x;
  float vx = 0.f, vy = 0.f, vz = 0.f, n = 0.f;
  if (i < size) {
    const float a = x[i], b = y[i], cc = z[i];
    if (a == a && b == b && cc == cc && fabsf(a) < 1e18f && fabsf(b) < 1e18f && fabsf(cc) < 1e18f) { vx = a; vy = b; vz = cc; n = 1.f; }
  }
  __shared__ float sm[4][4];
  vx = wave_sum(vx); vy = wave_sum(vy); vz = wave_sum(vz); n = wave_sum(n);
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6][0] = vx; sm[threadIdx.x >> 6][1] = vy; sm[threadIdx.x >> 6][2] = vz; sm[threadIdx.x >> 6][3] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int w = 0; w < 4; ++w) { s0 += sm[w][0]; s1 += sm[w][1]; s2 += sm[w][2]; s3 += sm[w][3]; }
    const float inv = s3 > 0.f ? 1.f / s3 : 0.f;
    out[blockIdx.x] = make_float4(s0 * inv, s1 * inv, s2 * inv, s3);
  }
}

// 30-bit Morton key of every surfel position (10 bits per axis inside the box lo .. lo + 1 / inv_span; invalid positions
// last), with the identity as payload: input of the per-surfel order of make_schedule.
__device__ __forceinline__ uint32_t spread10(uint32_t v) {
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__global__ __launch_bounds__(256) void surfel_morton_key_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t size,
                                                                f3 lo, f3 inv_span, uint32_t* __restrict__ keys, uint32_t* __restrict__ ids) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= size) return;
  const float a = x[i], b = y[i], cc = z[i];
  uint32_t key = 0x3fffffffu;
  if (a == a && b == b && cc == cc && fabsf(a) < 1e18f && fabsf(b) < 1e18f && fabsf(cc) < 1e18f) {
    const uint32_t qx = (uint32_t)__builtin_amdgcn_fmed3f((a - lo.x) * inv_span.x * 1023.f, 0.f, 1023.f);
    const uint32_t qy = (uint32_t)__builtin_amdgcn_fmed3f((b - lo.y) * inv_span.y * 1023.f, 0.f, 1023.f);
    const uint32_t qz = (uint32_t)__builtin_amdgcn_fmed3f((cc - lo.z) * inv_span.z * 1023.f, 0.f, 1023.f);
    key = spread10(qx) | (spread10(qy) << 1) | (spread10(qz) << 2);
  }
  keys[i] = key;
  ids[i] = i;
}

// Association probe: out[i] = py * width + px or 0xffffffff.
__global__ __launch_bounds__(256) void association_kernel(CamConsts c, const KfDev* __restrict__ kfs, SurfelRowsRW s, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.size) return;
  const f3 gp = mk3(s.x[i], s.y[i], s.z[i]);
  const f3 gn = unpack_normal(s.normal[i]);
  Proj p;
  out[i] = project_and_associate(c, kfs[0], gp, gn, &p) ? (uint32_t)(p.py * c.width + p.px) : 0xffffffffu;
}

// ---------------------------------------------------------------------------------------------
// AssignColorsCUDA (BS/kernel_assign_colors.cu:42-125): mean bilinear colour over the associated pixels of all
// keyframes.  One launch, thread per surfel, sums in registers in keyframe order (the reference accumulates in
// rows 8..12 with one launch per keyframe: same order of additions).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t texel_rgba(const KfDev& kf, const CamConsts& c, int ix, int iy) {   // clamp addressing
  ix = min(max(ix, 0), c.color_width - 1);
  iy = min(max(iy, 0), c.color_height - 1);
  return gload((const uint32_t*)(kf.color + (size_t)iy * kf.color_pitch) + ix);
}
__device__ __forceinline__ uint8_t color_to_u8(float v) { return (uint8_t)min(255, max(0, f2i(v))); }   // cvt.rzi.u8.f32

__global__ __launch_bounds__(256) void assign_colors_kernel(CamConsts c, const KfDev* __restrict__ kfs, int kf_count, SurfelRowsRW s,
                                                            uint32_t* __restrict__ color_row) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.size) return;
  const f3 gp = mk3(s.x[i], s.y[i], s.z[i]);
  const f3 gn = unpack_normal(s.normal[i]);
  float count = 0.f, sum[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < kf_count; ++k) {
    const KfDev kf = kfs[k];
    Proj p;
    if (!project_and_associate(c, kf, gp, gn, &p)) continue;
    f2 cp;
    if (!depth_to_color_pxy(c, p.pxy, &cp)) continue;
    const TexFootprint f = tex_footprint(c, cp.x, cp.y);
    const uint32_t t00 = texel_rgba(kf, c, f.i, f.j), t10 = texel_rgba(kf, c, f.i + 1, f.j);
    const uint32_t t01 = texel_rgba(kf, c, f.i, f.j + 1), t11 = texel_rgba(kf, c, f.i + 1, f.j + 1);
    const float w00 = (1.0f - f.a) * (1.0f - f.b), w10 = f.a * (1.0f - f.b), w01 = (1.0f - f.a) * f.b, w11 = f.a * f.b;
    count += 1.f;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const float v00 = (float)((t00 >> (8 * ch)) & 0xffu) * (1.0f / 255.0f), v10 = (float)((t10 >> (8 * ch)) & 0xffu) * (1.0f / 255.0f);
      const float v01 = (float)((t01 >> (8 * ch)) & 0xffu) * (1.0f / 255.0f), v11 = (float)((t11 >> (8 * ch)) & 0xffu) * (1.0f / 255.0f);
      sum[ch] += ((w00 * v00 + w10 * v10) + w01 * v01) + w11 * v11;
    }
  }
  if (count > 0) {
    uint32_t packed = 0;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) packed |= (uint32_t)color_to_u8(255.f * sum[ch] / count + 0.5f) << (8 * ch);
    color_row[i] = packed;
  }
}

// bslam_debug_decode_normals: every u16 normal code through u16_to_image_space_normal
__global__ __launch_bounds__(256) void decode_normals_kernel(float* __restrict__ out) {
  const uint32_t code = blockIdx.x * 256u + threadIdx.x;
  const f3 n = u16_to_image_space_normal(code);
  out[3 * code + 0] = n.x;
  out[3 * code + 1] = n.y;
  out[3 * code + 2] = n.z;
}

// bslam_debug_jacobians: the residual / Jacobian functions of device_math.hpp at given points (layouts: include/badslam_hip.h)
__global__ __launch_bounds__(256) void jacobian_probe_kernel(int kind, int count, int in_width, int out_width, const float* __restrict__ in, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float* a = in + (size_t)i * in_width;
  float* o = out + (size_t)i * out_width;
  if (kind == 0 || kind == 6) {
    const f3 nl = mk3(a[1], a[2], a[3]), lu = mk3(a[4], a[5], a[6]), ls = mk3(a[7], a[8], a[9]);
    float raw, J[6];
    if (kind == 0) { raw = depth_residual(a[0], nl, lu, ls); depth_pose_jacobian(a[0], nl, lu, J); }
    else depth_residual_and_pose_jacobian_fused(a[0], nl, lu, ls, &raw, J);
    o[0] = raw;
    for (int k = 0; k < 6; ++k) o[1 + k] = J[k];
  } else if (kind == 1) {
    o[0] = depth_position_jacobian(a[0]);
  } else if (kind == 7) {
    o[0] = rcp_rn_midrange(a[0]);
    o[1] = 1.0f / a[0];
  } else if (kind == 2) {
    const float m[12] = {a[9], a[10], a[11], 0, a[12], a[13], a[14], 0, 0, 0, 0, 0};
    float dj[6];
    o[0] = depth_intrinsics_jacobian(a[0], a[1], (int)a[2], (int)a[3], a[4], a[5], mk3(a[6], a[7], a[8]), m, mk3(a[15], a[16], a[17]), a[18], a[19], a[20], dj);
    for (int k = 0; k < 6; ++k) o[1 + k] = dj[k];
  } else {
    const LumaQuad q{a[0], a[1], a[2], a[3]};
    float dx, dy;
    bilinear_gradient_bytes(q, a[4], a[5], &dx, &dy);   // the residual path's forms (scale-agnostic: any texel unit)
    if (kind == 3) {
      float J[6];
      o[0] = bilinear_bytes(q, a[4], a[5]);
      o[1] = dx * a[6];
      o[2] = dy * a[7];
      descriptor_pose_jacobian(o[1], o[2], mk3(a[8], a[9], a[10]), J);
      for (int k = 0; k < 6; ++k) o[3 + k] = J[k];
    } else if (kind == 4) {
      o[0] = descriptor_position_jacobian(dx, dy, a[6], a[7], mk3(a[8], a[9], a[10]), mk3(a[11], a[12], a[13]));
    } else {
      float j[4];
      color_intrinsics_jacobian(dx, dy, a[6], a[7], j);
      for (int k = 0; k < 4; ++k) o[k] = j[k];
    }
  }
}

// Census: pairs passing z > 0 and bounds, and associated pairs (roofline accounting).
__global__ __launch_bounds__(256) void count_pairs_kernel(CamConsts c, const KfDev* __restrict__ kfs, int kf_count, SurfelRowsRW s,
                                                          unsigned long long* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t inb = 0, assoc = 0;
  if (i < s.size) {
    const f3 gp = mk3(s.x[i], s.y[i], s.z[i]);
    const f3 gn = unpack_normal(s.normal[i]);
    for (int k = 0; k < kf_count; ++k) {
      const KfDev kf = kfs[k];
      const M34& T = kf.frame_T_global;
      f3 l;
      l.z = tr_row(T.m[8], T.m[9], T.m[10], T.m[11], gp);
      if (l.z <= 0.f) continue;
      l.x = tr_row(T.m[0], T.m[1], T.m[2], T.m[3], gp);
      l.y = tr_row(T.m[4], T.m[5], T.m[6], T.m[7], gp);
      const f2 pxy = project(c.fx, c.fy, c.cx, c.cy, l);
      if (pxy.x < 0 || pxy.y < 0 || f2i(pxy.x) >= c.width || f2i(pxy.y) >= c.height) continue;
      inb += 1;
      Proj p;
      if (project_and_associate(c, kf, gp, gn, &p)) assoc += 1;
    }
  }
  inb = wave_sum_u32(inb);
  assoc = wave_sum_u32(assoc);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[0], (unsigned long long)inb);
    atomicAdd(&out[1], (unsigned long long)assoc);
  }
}

// Per-surfel residual probe for one keyframe (parity aid): out[i*8 + ...] =
// [depth raw residual, depth weight, desc r1, desc w1, desc r2, desc w2, flags, 0].
template <bool kDepth, bool kDesc>
__global__ __launch_bounds__(256) void residual_probe_kernel(CamConsts c, const KfDev* __restrict__ kfs, SurfelRowsRW s, float* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.size) return;
  float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const f3 gp = mk3(s.x[i], s.y[i], s.z[i]);
  const f3 gn = unpack_normal(s.normal[i]);
  const KfDev& kf = kfs[0];
  Proj p;
  if (project_and_associate(c, kf, gp, gn, &p)) {
    uint32_t flags = 1;
    float J[6];
    if (kDepth) {
      float raw;
      depth_residual_and_jacobian(c, p, &raw, J);
      o[0] = raw; o[1] = depth_weight(raw);
    }
    if (kDesc) {
      f2 color_pxy;
      if (depth_to_color_pxy(c, p.pxy, &color_pxy)) {
        f2 t1, t2;
        tangent_projections(gp, gn, s.radius_squared[i], kf.frame_T_global, c, &t1, &t2);
        float r1, r2;
        raw_descriptor_residual(kf, c, color_pxy, t1, t2, s.d1[i], s.d2[i], &r1, &r2);
        o[2] = r1; o[3] = desc_weight(r1); o[4] = r2; o[5] = desc_weight(r2);
        flags |= 2;
      }
    }
    o[6] = (float)flags;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) out[(size_t)i * 8 + k] = o[k];
}

// Geometry-only iteration (normals, then position along the normal: UpdateSurfelNormalsCUDA + OptimizeGeometryIterationCUDA,
// BS/kernel_opt_geometry.cc:39-78, 137-169; kernels BS/kernel_opt_geometry.cu:417-459, 487-507, 527-597) with R surfels per thread,
// interleaved keyframe by keyframe: R independent gather chains per thread, per-surfel sums formed in keyframe order -- the order in
// which the reference's serialised per-keyframe launches add their terms.  One launch covers all surfels (first_i = 0); round 1 launched it
// one resident grid at a time to keep the workgroups in lockstep on the keyframe table, which the per-surfel work order made
// unnecessary (first_i = first slot of every XCD's range handled by a launch is kept for that use).
template <int R>
__global__ __launch_bounds__(256) void geometry_position_kernel(CamConsts c_in, const KfDev* __restrict__ kfs, int kf_count, Schedule sc, uint32_t first_i,
                                                               SurfelRowsRW s) {
  CamConsts c = c_in;
  uint32_t slot;
  if (!slot_of_block(sc, blockIdx.x + (first_i << 3), &slot)) return;
  uint32_t idx[R];
  bool on[R];
  f3 gp[R], gn[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    idx[r] = surfel_of_slot(sc, slot, r, R);
    on[r] = idx[r] < s.size;
    if (on[r]) on[r] = (s.active[column_of(s, idx[r])] & BSLAM_SURFEL_ACTIVE_FLAG) != 0;
    const uint32_t j = on[r] ? idx[r] : 0;
    gp[r] = mk3(s.x[j], s.y[j], s.z[j]);
    gn[r] = unpack_normal(s.normal[j]);
  }
  BSLAM_HOIST_DEPTH_CAM_CENTRE(c);
  BSLAM_HOIST_UNPROJECTION_CENTRE(c);
  {
    float sx[R], sy[R], sz[R], cnt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) sx[r] = sy[r] = sz[r] = cnt[r] = 0.f;
    BSLAM_FOR_VISITED_KEYFRAMES(k, 0, kf_count, R) {
      KfDev kf = kfs[k];
      BSLAM_HOIST_KF_TRANSLATION(kf);
      const float* Rm = kf.global_R_frame;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        Proj p;
        if (!on[r] || !project_and_associate(c, kf, gp[r], gn[r], &p)) continue;
        const f3 ln = p.pixel_normal;
        sx[r] += rot_row(Rm[0], Rm[1], Rm[2], ln);
        sy[r] += rot_row(Rm[3], Rm[4], Rm[5], ln);
        sz[r] += rot_row(Rm[6], Rm[7], Rm[8], ln);
        cnt[r] += 1.f;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (on[r] && cnt[r] >= 1) {
        const float inv = 1.f / cnt[r];
        const uint32_t packed = pack_normal(mk3(inv * sx[r], inv * sy[r], inv * sz[r]));
        store_normal(s, idx[r], packed);
        gn[r] = unpack_normal(packed);
      }
    }
  }
  float H[R], b[R];
#pragma unroll
  for (int r = 0; r < R; ++r) H[r] = b[r] = 0.f;
  BSLAM_FOR_VISITED_KEYFRAMES(k, 0, kf_count, R) {
    KfDev kf = kfs[k];
    BSLAM_HOIST_KF_TRANSLATION(kf);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      Proj p;
      if (!on[r] || !project_and_associate(c, kf, gp[r], gn[r], &p)) continue;
      const float inv_stddev = depth_inv_stddev(p.nx, p.ny, p.depth, p.n_local, c.baseline_fx);
      const float dj = depth_position_jacobian(inv_stddev);
      const f3 lu = mk3(p.depth * p.nx, p.depth * p.ny, p.depth);   // unproject(c, p.px, p.py, p.depth)
      const float raw = depth_residual(inv_stddev, p.n_local, lu, p.local);
      const float w = depth_weight(raw);
      const float wj = w * dj;
      H[r] += wj * dj;
      b[r] += wj * raw;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (on[r] && H[r] > 1e-6f) {
      const float t = -1.f * b[r] / H[r];
      const f3 np = add3(gp[r], scale3(t, gn[r]));
      store_position(s, idx[r], np);
    }
  }
}

// The same two passes for long keyframe lists, one launch per chunk of keyframes: the per-surfel sums travel between the
// launches in library scratch (acc: 4 floats per surfel).  Sums are still formed in keyframe order: same bits as the
// single-launch kernel.  (Selected by bslam_set_geometry_keyframe_chunk; the default since the per-surfel work order is one
// launch over the whole list.)  pass 0: normals (acc = sx, sy, sz, count), pass 1: position (acc = H, b).
template <int R, int kPass>
__global__ __launch_bounds__(256) void geometry_chunk_kernel(CamConsts c, const KfDev* __restrict__ kfs, int k_begin, int k_end, int first_chunk, int last_chunk,
                                                            Schedule sc, uint32_t first_i, SurfelRowsRW s, float* __restrict__ acc, uint32_t acc_pitch) {
  uint32_t slot;
  if (!slot_of_block(sc, blockIdx.x + (first_i << 3), &slot)) return;
  uint32_t idx[R];
  bool on[R];
  f3 gp[R], gn[R];
  float a0[R], a1[R], a2[R], a3[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    idx[r] = surfel_of_slot(sc, slot, r, R);
    on[r] = idx[r] < s.size;
    if (on[r]) on[r] = (s.active[column_of(s, idx[r])] & BSLAM_SURFEL_ACTIVE_FLAG) != 0;
    const uint32_t j = on[r] ? idx[r] : 0;
    gp[r] = mk3(s.x[j], s.y[j], s.z[j]);
    gn[r] = unpack_normal(s.normal[j]);
    a0[r] = a1[r] = a2[r] = a3[r] = 0.f;
    if (!first_chunk && on[r]) {
      a0[r] = acc[j]; a1[r] = acc[(size_t)acc_pitch + j];
      if (kPass == 0) { a2[r] = acc[(size_t)2 * acc_pitch + j]; a3[r] = acc[(size_t)3 * acc_pitch + j]; }
    }
  }
  BSLAM_FOR_VISITED_KEYFRAMES(k, k_begin, k_end, R) {
    const KfDev kf = kfs[k];
    const float* Rm = kf.global_R_frame;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      Proj p;
      if (!on[r] || !project_and_associate(c, kf, gp[r], gn[r], &p)) continue;
      if (kPass == 0) {
        const f3 ln = p.pixel_normal;
        a0[r] += rot_row(Rm[0], Rm[1], Rm[2], ln);
        a1[r] += rot_row(Rm[3], Rm[4], Rm[5], ln);
        a2[r] += rot_row(Rm[6], Rm[7], Rm[8], ln);
        a3[r] += 1.f;
      } else {
        const float inv_stddev = depth_inv_stddev(p.nx, p.ny, p.depth, p.n_local, c.baseline_fx);
        const float dj = depth_position_jacobian(inv_stddev);
        const f3 lu = mk3(p.depth * p.nx, p.depth * p.ny, p.depth);
        const float raw = depth_residual(inv_stddev, p.n_local, lu, p.local);
        const float w = depth_weight(raw);
        const float wj = w * dj;
        a0[r] += wj * dj;
        a1[r] += wj * raw;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!on[r]) continue;
    const uint32_t j = idx[r];
    if (!last_chunk) {
      acc[j] = a0[r]; acc[(size_t)acc_pitch + j] = a1[r];
      if (kPass == 0) { acc[(size_t)2 * acc_pitch + j] = a2[r]; acc[(size_t)3 * acc_pitch + j] = a3[r]; }
    } else if (kPass == 0) {
      if (a3[r] >= 1) {
        const float inv = 1.f / a3[r];
        store_normal(s, j, pack_normal(mk3(inv * a0[r], inv * a1[r], inv * a2[r])));
      }
    } else if (a0[r] > 1e-6f) {
      const float t = -1.f * a1[r] / a0[r];
      const f3 np = add3(gp[r], scale3(t, gn[r]));
      store_position(s, j, np);
    }
  }
}

// The joint position + descriptor iteration (kMode 2 of geometry_kernel) in the same launch shape as geometry_chunk_kernel: R surfels
// per thread walked keyframe by keyframe, the keyframe list optionally cut into chunks whose per-surfel sums
// travel in library scratch (acc: 4 floats per surfel for the normals pass, 8 for the joint pass).  Sums are formed in keyframe
// order: same bits as geometry_kernel<2, kDepth>.  kPass 0: normals (= geometry_chunk_kernel's pass 0, repeated here so that a
// photometric iteration needs no geometry-only instantiation of a different R); kPass 1: position + descriptors
// (BS/kernel_opt_geometry.cu:118-231, 273-361).
#ifndef BSLAM_GEOM_DESC_WAVES
#define BSLAM_GEOM_DESC_WAVES 4
#endif
template <int R, int kPass, bool kDepth>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BSLAM_GEOM_DESC_WAVES))) void geometry_desc_chunk_kernel(CamConsts c_in, const KfDev* __restrict__ kfs, int k_begin, int k_end, int first_chunk, int last_chunk,
                                                                 Schedule sc, uint32_t first_i, SurfelRowsRW s, float* __restrict__ acc, uint32_t acc_pitch) {
  CamConsts c = c_in;
  uint32_t slot;
  if (!slot_of_block(sc, blockIdx.x + (first_i << 3), &slot)) return;
  constexpr int kAcc = kPass == 0 ? 4 : 8;
  uint32_t idx[R];
  bool on[R];
  f3 gp[R], gn[R], tp1[R], tp2[R];
  float desc1[R], desc2[R];
  float a[R][kAcc];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    idx[r] = surfel_of_slot(sc, slot, r, R);
    on[r] = idx[r] < s.size;
    if (on[r]) on[r] = (s.active[column_of(s, idx[r])] & BSLAM_SURFEL_ACTIVE_FLAG) != 0;
    const uint32_t j = on[r] ? idx[r] : 0;
    gp[r] = mk3(s.x[j], s.y[j], s.z[j]);
    gn[r] = unpack_normal(s.normal[j]);
    if constexpr (kPass == 1) {
      desc1[r] = s.d1[j]; desc2[r] = s.d2[j];
      tangent_points(gp[r], gn[r], s.radius_squared[j], &tp1[r], &tp2[r]);
    }
#pragma unroll
    for (int q = 0; q < kAcc; ++q) a[r][q] = (!first_chunk && on[r]) ? acc[(size_t)q * acc_pitch + j] : 0.f;
  }
  if constexpr (kPass == 1) BSLAM_HOIST_CAM_CENTRES(c);
  if constexpr (kPass == 1) BSLAM_HOIST_UNPROJECTION_CENTRE(c);
  BSLAM_FOR_VISITED_KEYFRAMES(k, k_begin, k_end, R) {
    KfDev kf = kfs[k];
    if constexpr (kPass == 1) BSLAM_HOIST_KF_TRANSLATION(kf);
    const float* Rm = kf.global_R_frame;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      Proj p;
      DescSamples ds;
      f2 color_pxy, t1, t2;   // the three sample positions of the descriptor residual
      bool has_desc = false;
      if constexpr (kPass == 1) {
        // the three quad gathers of the descriptor samples do not depend on the pixel record: issued with the record gather,
        // before the association test (see pose_accumulate_kernel)
        if (!on[r] || !project_to_pixel(c, kf, gp[r], &p)) continue;
        const PixelRecord rec = load_record(c, kf, p);
        has_desc = depth_to_color_pxy_in_bounds(c, p.pxy, &color_pxy);
        project_tangent_points(tp1[r], tp2[r], kf.frame_T_global, c, &t1, &t2);
        ds = descriptor_samples_issue(kf, c, color_pxy, t1, t2);
        asm volatile("" ::: "memory");
        if (!associate_with_record(c, kf, gn[r], rec, &p)) continue;
      } else {
        if (!on[r] || !project_and_associate(c, kf, gp[r], gn[r], &p)) continue;
      }
      if constexpr (kPass == 0) {
        const f3 ln = p.pixel_normal;
        a[r][0] += rot_row(Rm[0], Rm[1], Rm[2], ln);
        a[r][1] += rot_row(Rm[3], Rm[4], Rm[5], ln);
        a[r][2] += rot_row(Rm[6], Rm[7], Rm[8], ln);
        a[r][3] += 1.f;
      } else {
        // accumulators: [0] H00, [1] H01, [2] H02, [3] H11, [4] H22, [5] b0, [6] b1, [7] b2   (H12 is never accumulated: quirk Q2)
#pragma clang fp contract(fast)   // the position / descriptor sums (compared at 1e-4); the normals pass above feeds packed normals and stays unfused
        const f3 rn = p.n_local;
        if (kDepth) {
          const float inv_stddev = depth_inv_stddev(p.nx, p.ny, p.depth, rn, c.baseline_fx);
          const float dj = depth_position_jacobian(inv_stddev);
          const f3 lu = mk3(p.depth * p.nx, p.depth * p.ny, p.depth);   // unproject(c, p.px, p.py, p.depth)
          const float raw = depth_residual(inv_stddev, rn, lu, p.local);
          const float w = depth_weight(raw);
          a[r][0] += w * dj * dj;
          a[r][5] += w * raw * dj;
        }
        if (has_desc) {
          float r1, rr2, gx1, gy1, gx2, gy2;
          descriptor_samples_finish(kf, c, ds, desc1[r], desc2[r], [&](f2 (&pts)[3]) { pts[0] = color_pxy; pts[1] = t1; pts[2] = t2; }, &r1, &rr2, &gx1, &gy1, &gx2, &gy2);
          const float jp1 = descriptor_position_jacobian(gx1, gy1, c.cfx, c.cfy, rn, p.local);
          const float jp2 = descriptor_position_jacobian(gx2, gy2, c.cfx, c.cfy, rn, p.local);
          const float jd = -1.f;
          const float w1 = desc_weight(r1);
          const float wr1 = w1 * r1;
          const float w2 = desc_weight(rr2);
          const float wr2 = w2 * rr2;
          a[r][0] += w1 * jp1 * jp1 + w2 * jp2 * jp2;
          a[r][1] += w1 * jp1 * jd;
          a[r][3] += w1 * jd * jd;
          a[r][5] += wr1 * jp1 + wr2 * jp2;
          a[r][6] += wr1 * jd;
          a[r][2] += w2 * jp2 * jd;
          a[r][4] += w2 * jd * jd;
          a[r][7] += wr2 * jd;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!on[r]) continue;
    const uint32_t j = idx[r];
    if (!last_chunk) {
#pragma unroll
      for (int q = 0; q < kAcc; ++q) acc[(size_t)q * acc_pitch + j] = a[r][q];
    } else if constexpr (kPass == 0) {
      if (a[r][3] >= 1) {
        const float inv = 1.f / a[r][3];
        store_normal(s, j, pack_normal(mk3(inv * a[r][0], inv * a[r][1], inv * a[r][2])));
      }
    } else {
      float H00 = a[r][0], H01 = a[r][1], H02 = a[r][2], H11 = a[r][3], H12 = 0.f, H22 = a[r][4];
      H00 += 1e-6f; H11 += 1e-6f; H22 += 1e-6f;
      H00 = sqrtf(H00);
      H01 = H01 / H00;
      H11 = sqrtf(H11 - H01 * H01);
      H02 = H02 / H00;
      H12 = (H12 - H02 * H01) / H11;
      H22 = sqrtf(H22 - H02 * H02 - H12 * H12);
      const float y0 = a[r][5] / H00;
      const float y1 = (a[r][6] - H01 * y0) / H11;
      const float y2 = (a[r][7] - H02 * y0 - H12 * y1) / H22;
      const float x2 = y2 / H22;
      const float x1 = (y1 - H12 * x2) / H11;
      const float x0 = (y0 - H02 * x2 - H01 * x1) / H00;
      if (x0 != 0) {
        const f3 np = sub3(gp[r], scale3(x0, gn[r]));
        store_position(s, j, np);
      }
      if (x1 != 0) store_descriptor1(s, j, fmaxf(-180.f, fminf(180.f, desc1[r] - x1)));
      if (x2 != 0) store_descriptor2(s, j, fmaxf(-180.f, fminf(180.f, desc2[r] - x2)));
    }
  }
}

}  // namespace bslam
