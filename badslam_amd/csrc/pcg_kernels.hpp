// pcg_kernels.hpp -- matrix-free preconditioned conjugate gradient kernels (gfx950).
//
// Replaces BS/kernel_pcg.cu.  The reference launches PCGInit / PCGStep1 once per keyframe over
// all surfels, with 13-29 block reductions + float atomics per residual and read-modify-write
// of the per-surfel entries of r, M, g on every launch.  Here one launch covers all keyframes:
//   * a thread keeps kPcgR surfels and their r/M (or p/g) entries in registers and walks the
//     keyframe table in list order, so per-surfel sums are formed in the reference's order and
//     written once;
//   * per-keyframe pose entries are reduced wave -> LDS -> one row per (tile, keyframe) and
//     summed over tiles by a second kernel in a fixed order (deterministic);
//   * the global intrinsics entries and alpha_d are accumulated per thread over the whole loop
//     and reduced once per block; only the per-cell cfactor entries use float atomics (as the
//     reference does, BS/kernel_pcg.cu:319).
#pragma once

#include "device_math.hpp"
#include "pose_kernels.hpp"

namespace bslam {

constexpr float kDiagEpsilon = 1e-8f;   // BS/kernel_pcg.cu:44
constexpr float kAPriorWeight = 10.f;   // BS/kernel_pcg.cu:48
constexpr uint32_t kInvalidUnknown = 0xffffffffu;

constexpr int kPcgThreads = 256;
#ifndef BSLAM_PCG_R
#define BSLAM_PCG_R 2
#endif
// ---------------------------------------------------------------------------------------------
// PCGStep1 for all keyframes (BS/kernel_pcg.cu:645-1025)
// ---------------------------------------------------------------------------------------------
#ifndef BSLAM_PCG_STEP1_WAVES_DESC
#define BSLAM_PCG_STEP1_WAVES_DESC 3
#endif
constexpr int kPcgR = BSLAM_PCG_R;
#ifndef BSLAM_PCG_R_GEO
#define BSLAM_PCG_R_GEO 3
#endif
// surfels per thread: the geometry-only kernels without intrinsics are light on registers and amortise the per-keyframe
// reduction (6 - 12 wave sums + a barrier) over twice as many pairs
constexpr int pcg_surfels_per_thread(bool desc, bool intr) { return (!desc && !intr) ? BSLAM_PCG_R_GEO : kPcgR; }
constexpr int kPcgPoseRow = 12;    // init: r[6], M[6];  step1: g[6] (+6 unused)
// Per-keyframe pose rows of a workgroup: every wave drops its wave sums into an LDS stash and moves on; after kPcgGroup visited
// keyframes ONE barrier lets the block add the four waves' entries -- ((w0 + w1) + w2) + w3, the order of the former
// per-keyframe reduction: same bits -- and store one row per (keyframe, work slot).  Two stashes alternate, so a wave that runs
// ahead into the next group never overwrites entries that are still being added (it cannot get two groups ahead: the barrier).
constexpr int kPcgGroup = 16;
struct PcgRowStash {
  float v[2][kPcgGroup][4][kPcgPoseRow];
  int kf[2][kPcgGroup];
  __attribute__((aligned(16))) float tile[4][4 * 64];   // wave-private tiles of the wave sums (wave_column_sums_lds, 4 columns per round)
};
// The wave sums of `pose` go into the stash entry of keyframe k, through the wave's LDS tile instead of six-step shuffle sums of
// each value (measured on one box: K = 300 photometric pcg_step1_kernel 16.7 -> 15.7 ms, pcg_init_kernel 18.6 -> 16.7 ms, BA
// iteration 340 -> 321 ms; K = 50 geometry-only 210 -> 187 us and 270 -> 195 us).
template <int kLive>
__device__ __forceinline__ void pcg_stash_wave_sums(PcgRowStash& st, int buf, int at, int k, const float (&pose)[kPcgPoseRow]) {
  const int wave = threadIdx.x >> 6;
  int col;
  bool writer;
  wave_column_sums_owner<kLive, 4>(&col, &writer);   // column lane / 4, stored by every fourth lane
  const float total = wave_column_sums_lds<kLive, 4>(pose, st.tile[wave]);
  if (writer && col < kPcgPoseRow) st.v[buf][at][wave][col] = total;
  if (threadIdx.x == 0) st.kf[buf][at] = k;
}
// Adds up and stores the `n` stashed keyframes of buffer `buf` (all threads call; contains the group's one barrier).
__device__ __forceinline__ void pcg_flush_rows(PcgRowStash& st, int buf, int n, float* __restrict__ partial_pose, uint32_t slots, int tile) {
  __syncthreads();
  const int j = threadIdx.x / kPcgPoseRow, col = threadIdx.x - j * kPcgPoseRow;
  if (j < n)
    partial_pose[((size_t)st.kf[buf][j] * slots + tile) * kPcgPoseRow + col] = ((st.v[buf][j][0][col] + st.v[buf][j][1][col]) + st.v[buf][j][2][col]) + st.v[buf][j][3][col];
}
static_assert(kPcgGroup * kPcgPoseRow <= kPcgThreads, "one thread per stashed value");
// Zero rows for the keyframes of the batch [k0, k0 + 64) that the workgroup does not visit (frustum culling): the row sums read
// every (keyframe, work slot) row.
__device__ __forceinline__ void pcg_zero_rows(unsigned long long visited, int k0, int k_end, float* __restrict__ partial_pose, uint32_t slots, int tile) {
  for (int i = threadIdx.x; i < 64 * kPcgPoseRow; i += kPcgThreads) {
    const int b = i / kPcgPoseRow, k = k0 + b;
    if (k < k_end && !((visited >> b) & 1ull)) partial_pose[((size_t)k * slots + tile) * kPcgPoseRow + (i - b * kPcgPoseRow)] = 0.f;
  }
}
constexpr int kPcgGlobRow = 20;    // init: depth intr r[5], M[5], colour r[4], M[4]; step1: alpha_d, g depth[5], g colour[4]

struct PcgParams {
  uint32_t unknown_count, surfel_start, depth_intr_start, a_index, color_intr_start;
  int gauge_kf;
  int optimize_poses, optimize_geometry, optimize_depth_intr, optimize_color_intr;
  int per_surfel;   // 1 (geometry only) or 3 (with descriptors)
  // unknowns [shard_begin, shard_end) belong to this rank's surfel shard; all others (poses, intrinsics,
  // cfactor cells) are shared between the ranks of a surfel-sharded run and are kept bit-identical on all
  uint32_t shard_begin, shard_end;
  float* r; float* M; float* delta; float* g; float* p;
  float* alpha_n; float* alpha_d; float* beta_n;
  // Sums over the pairs that fall into one cfactor cell (library scratch, zero on entry): formed with fp64 atomics
  // and rounded to fp32 once by pcg_cf_flush_kernel.  The reference adds floats atomically in arrival order
  // (BS/kernel_pcg.cu: atomicAddFloatOrDouble), which makes whole BA runs differ from one run to the next; in
  // fp64 the order only matters below 1e-16 relative, so the rounded sums are the same on every run.
  double* cf_acc0; double* cf_acc1;
  uint32_t cf_cells;
  // per-surfel work order (make_schedule): the surfel kernels read the library's sorted copy of the surfel rows; position j of
  // it is the caller's column perm[j], which is what the unknown layout counts in (nullptr: identity)
  const uint32_t* perm;
};

__device__ __forceinline__ uint32_t kf_pose_unknown_index(int gauge, int id) {   // BS/direct_ba_pcg.cc:329-337
  if (id == gauge) return kInvalidUnknown;
  return (uint32_t)(6 * (id < gauge ? id : id - 1));
}

// Block-wide sum of `n` per-thread values (n <= 20) into out[0..n) by thread 0..n-1; all threads call.
template <int N>
__device__ __forceinline__ void block_reduce_rows(float* vals, float (*red)[32], float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) vals[i] = wave_sum(vals[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) red[wave][i] = vals[i];
  }
  __syncthreads();
  if (threadIdx.x < N) out[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

struct DepthIntrinsicsTerms {
  bool valid;
  float d[5];           // fx_inv, fy_inv, cx_inv, cy_inv, a
  float cf_jac;         // Jacobian wrt. the cfactor cell
  uint32_t cf_index;    // unknown index of that cell
};

// BS/kernel_pcg.cu:258-322 / 721-772
__device__ __forceinline__ DepthIntrinsicsTerms depth_intrinsics_terms(const CamConsts& c, const KfDev& kf, const Proj& p, f3 gn, float inv_stddev, uint32_t d0) {
  DepthIntrinsicsTerms t;
  const int sparse_px = p.px / c.cell, sparse_py = p.py / c.cell;
  const float cfactor = *(const float*)((const uint8_t*)c.cfactor + (size_t)sparse_py * c.cfactor_pitch + 4 * (size_t)sparse_px);
  const uint32_t measured = raw_depth_of(kf, p);
  const float raw_inv_depth = 1.0f / (c.raw_to_float_depth * (float)measured);
  const float nx = nx_of(c, (float)p.px), ny = ny_of(c, (float)p.py);
  float dj[6];
  const float corrected_inv_depth = depth_intrinsics_jacobian(inv_stddev, p.depth, p.px, p.py, nx, ny, gn, kf.frame_T_global.m, p.n_local, cfactor, c.a,
                                                              raw_inv_depth, dj);
  t.valid = !(fabsf(corrected_inv_depth) < 1e-4f);
#pragma unroll
  for (int j = 0; j < 5; ++j) t.d[j] = dj[j];
  t.cf_jac = dj[5];
  t.cf_index = d0 + 5 + (uint32_t)sparse_px + (uint32_t)sparse_py * (uint32_t)c.cfactor_width;
  return t;
}

struct DescTerms {
  float r1, r2, w1, w2;
  float gx1, gy1, gx2, gy2;   // image gradients already multiplied by the colour focal lengths
};

// In two steps around the association test: the sample positions depend on the surfel and the pose only, so their three quad
// gathers are issued together with the record gather (see pose_accumulate_kernel); the filters run after the test.
__device__ __forceinline__ DescSamples descriptor_terms_issue(const CamConsts& c, const KfDev& kf, f3 tp1, f3 tp2, f2 color_pxy, f2* t1, f2* t2) {
  project_tangent_points(tp1, tp2, kf.frame_T_global, c, t1, t2);
  return descriptor_samples_issue(kf, c, color_pxy, *t1, *t2);
}

// Photometric variants: per-surfel constants that a pair only reads (normal, the two tangent sample points, descriptors, and
// in step 1 the surfel's entries of p) live in LDS -- [component][thread], conflict-free, private to the thread, no barrier --
// instead of VGPRs: registers for one more wave per SIMD, and the tangent points are formed once per surfel instead of once
// per pair.
constexpr int kPcgStateComps = 14;   // 0-2 normal, 3-5 / 6-8 tangent points, 9-10 descriptor, 11-13 p entries (step 1)
#define BSLAM_PCG_ST(r, comp) state[((r) * kPcgStateComps + (comp)) * kPcgThreads + threadIdx.x]
template <class SamplePoints>
__device__ __forceinline__ DescTerms descriptor_terms_finish(const CamConsts& c, const KfDev& kf, const DescSamples& ds, float d1, float d2, SamplePoints&& sample_points) {
  DescTerms t;
  descriptor_samples_finish(kf, c, ds, d1, d2, sample_points, &t.r1, &t.r2, &t.gx1, &t.gy1, &t.gx2, &t.gy2);
  t.gx1 *= c.cfx; t.gx2 *= c.cfx;
  t.gy1 *= c.cfy; t.gy2 *= c.cfy;
  t.w1 = desc_weight(t.r1);
  t.w2 = desc_weight(t.r2);
  return t;
}

// ---------------------------------------------------------------------------------------------
// PCGInit for all keyframes (BS/kernel_pcg.cu:179-513, BS/direct_ba_pcg.cc:339-365)
// ---------------------------------------------------------------------------------------------
template <bool kDepth, bool kDesc, bool kIntr>
__global__ __launch_bounds__(kPcgThreads) void pcg_init_kernel(
    CamConsts c_in, const KfDev* __restrict__ kfs, int kf_count, Schedule sc, SurfelRows s, PcgParams P,
    float* __restrict__ partial_pose, float* __restrict__ partial_glob) {
  CamConsts c = c_in;
  constexpr int R = pcg_surfels_per_thread(kDesc, kIntr);
  uint32_t slot;
  if (!slot_of_block(sc, blockIdx.x, &slot)) return;
  const int tile = (int)slot;
  __shared__ PcgRowStash stash;
  __shared__ float redg[4][32];

  f3 gp[R], gn[R];
  bool valid[R];
  uint32_t idx[R];
  __shared__ float state[kDesc ? kPcgStateComps * R * kPcgThreads : 1];
  float ar[R][3], aM[R][3];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = surfel_of_slot(sc, slot, r, R);
    valid[r] = i < s.size;
    idx[r] = valid[r] ? i : 0;
    gp[r] = mk3(s.x[idx[r]], s.y[idx[r]], s.z[idx[r]]);
    gn[r] = unpack_normal(s.normal[idx[r]]);
    if constexpr (kDesc) {
      f3 tp1, tp2;
      tangent_points(gp[r], gn[r], s.radius_squared[idx[r]], &tp1, &tp2);
      BSLAM_PCG_ST(r, 0) = gn[r].x; BSLAM_PCG_ST(r, 1) = gn[r].y; BSLAM_PCG_ST(r, 2) = gn[r].z;
      BSLAM_PCG_ST(r, 3) = tp1.x; BSLAM_PCG_ST(r, 4) = tp1.y; BSLAM_PCG_ST(r, 5) = tp1.z;
      BSLAM_PCG_ST(r, 6) = tp2.x; BSLAM_PCG_ST(r, 7) = tp2.y; BSLAM_PCG_ST(r, 8) = tp2.z;
      BSLAM_PCG_ST(r, 9) = s.d1[idx[r]]; BSLAM_PCG_ST(r, 10) = s.d2[idx[r]];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { ar[r][j] = 0.f; aM[r][j] = 0.f; }
  }
  float glob[kPcgGlobRow];
#pragma unroll
  for (int i = 0; i < kPcgGlobRow; ++i) glob[i] = 0.f;

  if constexpr (kDesc) BSLAM_HOIST_CAM_CENTRES(c);
  else BSLAM_HOIST_DEPTH_CAM_CENTRE(c);
  BSLAM_HOIST_UNPROJECTION_CENTRE(c);
  int stashed = 0;   // keyframes in the current stash (uniform)
  int buf = 0;
  for (int k0 = 0; k0 < kf_count; k0 += 64) {
  unsigned long long todo = keyframes_to_visit(c, kfs, k0, kf_count, sc, slot, R, true);
  if (P.optimize_poses && sc.bounds != nullptr) pcg_zero_rows(todo, k0, kf_count, partial_pose, sc.slots, tile);
  for (; todo != 0; todo &= todo - 1) {
    const int k = k0 + __builtin_ctzll(todo);
    KfDev kf = kfs[k];
    BSLAM_HOIST_KF_TRANSLATION(kf);
    const uint32_t kf_idx = kf_pose_unknown_index(P.gauge_kf, kf.id);
    const bool opt_pose = P.optimize_poses && kf_idx != kInvalidUnknown;
    float pose[kPcgPoseRow];
#pragma unroll
    for (int i = 0; i < kPcgPoseRow; ++i) BSLAM_ZERO(pose[i]);   // independent zeros: see pose_accumulate_kernel

#pragma unroll
    for (int r = 0; r < R; ++r) {
      Proj p;
      DescSamples ds;
      f2 color_pxy, t1, t2;   // the three sample positions of the descriptor residual
      bool has_desc = false;
      if constexpr (kDesc) {
        if (!valid[r] || !project_to_pixel(c, kf, gp[r], &p)) continue;
        const PixelRecord rec = load_record(c, kf, p);
        has_desc = depth_to_color_pxy_in_bounds(c, p.pxy, &color_pxy);
        ds = descriptor_terms_issue(c, kf, mk3(BSLAM_PCG_ST(r, 3), BSLAM_PCG_ST(r, 4), BSLAM_PCG_ST(r, 5)),
                                    mk3(BSLAM_PCG_ST(r, 6), BSLAM_PCG_ST(r, 7), BSLAM_PCG_ST(r, 8)), color_pxy, &t1, &t2);
        asm volatile("" ::: "memory");   // the gathers stay in front of the branches of the association test
        if (!associate_with_record(c, kf, mk3(BSLAM_PCG_ST(r, 0), BSLAM_PCG_ST(r, 1), BSLAM_PCG_ST(r, 2)), rec, &p)) continue;
      } else {
        if (!valid[r] || !project_and_associate(c, kf, gp[r], gn[r], &p)) continue;
      }
      bool visible = true;
      const f3 rn = p.n_local;
      if (kDepth) {
#pragma clang fp contract(fast)   // past the association test nothing feeds an integer output: products with p and the sums fuse (as nvcc's default does)
        const float inv_stddev = depth_inv_stddev(p.nx, p.ny, p.depth, rn, c.baseline_fx);
        const f3 lu = mk3(p.depth * p.nx, p.depth * p.ny, p.depth);   // unproject(c, p.px, p.py, p.depth)
        const float raw = depth_residual(inv_stddev, rn, lu, p.local);
        const float weight = depth_weight(raw);
        if (P.optimize_geometry) {                               // :217-221
          const float jp = depth_position_jacobian(inv_stddev);
          ar[r][0] -= jp * weight * raw;
          aM[r][0] += jp * weight * jp;
        }
        if (opt_pose) {                                          // :224-255
          float J[6];
          depth_pose_jacobian(inv_stddev, rn, lu, J);
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const float wj = weight * J[j];
            pose[j] += -1 * wj * raw;
            pose[6 + j] += J[j] * wj;
          }
        }
        if (kIntr && P.optimize_depth_intr) {                    // :258-322
          const DepthIntrinsicsTerms t = depth_intrinsics_terms(c, kf, p, kDesc ? mk3(BSLAM_PCG_ST(r, 0), BSLAM_PCG_ST(r, 1), BSLAM_PCG_ST(r, 2)) : gn[r], inv_stddev, P.depth_intr_start);
          if (t.valid) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
              const float wj = weight * t.d[j];
              glob[j] += -1 * wj * raw;
              glob[5 + j] += t.d[j] * wj;
            }
            const float wj = weight * t.cf_jac;
            const uint32_t cell = t.cf_index - (P.depth_intr_start + 5);
            atomicAdd(&P.cf_acc0[cell], (double)(-1 * wj * raw));
            atomicAdd(&P.cf_acc1[cell], (double)(t.cf_jac * wj));
          } else {
            visible = false;                                      // :272 (also disables the descriptor part)
          }
        }
      }
      if (kDesc) {                                               // :330-511
#pragma clang fp contract(fast)
        visible = visible && has_desc;
        if (!visible) continue;
        const DescTerms t = descriptor_terms_finish(c, kf, ds, BSLAM_PCG_ST(r, 9), BSLAM_PCG_ST(r, 10), [&](f2 (&pts)[3]) { pts[0] = color_pxy; pts[1] = t1; pts[2] = t2; });
        const f3 ls = p.local;
        if (P.optimize_geometry) {                               // :364-399
          const float jp1 = descriptor_position_jacobian(t.gx1, t.gy1, 1.f, 1.f, rn, ls);   // gx, gy already carry fx, fy
          const float jp2 = descriptor_position_jacobian(t.gx2, t.gy2, 1.f, 1.f, rn, ls);
          ar[r][0] -= jp1 * t.w1 * t.r1 + jp2 * t.w2 * t.r2;
          aM[r][0] += jp1 * t.w1 * jp1 + jp2 * t.w2 * jp2;
          const float j11 = -1, j12 = 0, j21 = 0, j22 = -1;
          ar[r][1] -= j11 * t.w1 * t.r1 + j12 * t.w2 * t.r2;
          aM[r][1] += j11 * t.w1 * j11 + j12 * t.w2 * j12;
          ar[r][2] -= j21 * t.w1 * t.r1 + j22 * t.w2 * t.r2;
          aM[r][2] += j21 * t.w1 * j21 + j22 * t.w2 * j22;
        }
        if (opt_pose) {                                          // :402-459
          float J1[6], J2[6];
          descriptor_pose_jacobian(t.gx1, t.gy1, ls, J1);
          descriptor_pose_jacobian(t.gx2, t.gy2, ls, J2);
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const float wj1 = t.w1 * J1[j], wj2 = t.w2 * J2[j];
            pose[j] += -1 * wj1 * t.r1 + -1 * wj2 * t.r2;
            pose[6 + j] += J1[j] * wj1 + J2[j] * wj2;
          }
        }
        if (kIntr && P.optimize_color_intr) {                    // :462-509
          const float gx_1 = t.gx1 / c.cfx, gy_1 = t.gy1 / c.cfy, gx_2 = t.gx2 / c.cfx, gy_2 = t.gy2 / c.cfy;
          const float nx = nx_of(c, (float)p.px), ny = ny_of(c, (float)p.py);
          const float Jc1[4] = {gx_1 * nx, gy_1 * ny, gx_1, gy_1};
          const float Jc2[4] = {gx_2 * nx, gy_2 * ny, gx_2, gy_2};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float wj1 = t.w1 * Jc1[j], wj2 = t.w2 * Jc2[j];
            glob[10 + j] += -1 * wj1 * t.r1 + -1 * wj2 * t.r2;
            glob[14 + j] += Jc1[j] * wj1 + Jc2[j] * wj2;
          }
        }
      }
    }

    if (opt_pose) {   // uniform
      pcg_stash_wave_sums<kPcgPoseRow>(stash, buf, stashed, k, pose);
      if (++stashed == kPcgGroup) { pcg_flush_rows(stash, buf, stashed, partial_pose, sc.slots, tile); stashed = 0; buf ^= 1; }
    }
  }
  }
  if (stashed) pcg_flush_rows(stash, buf, stashed, partial_pose, sc.slots, tile);

  if (P.optimize_geometry) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!valid[r]) continue;
      const uint32_t base = P.surfel_start + (uint32_t)P.per_surfel * (P.perm ? P.perm[idx[r]] : idx[r]);
      P.r[base] = ar[r][0];
      P.M[base] = aM[r][0];
      if (kDesc) {
        P.r[base + 1] = ar[r][1]; P.M[base + 1] = aM[r][1];
        P.r[base + 2] = ar[r][2]; P.M[base + 2] = aM[r][2];
      }
    }
  }
  if (kIntr) block_reduce_rows<kPcgGlobRow>(glob, redg, partial_glob + (size_t)tile * kPcgGlobRow);
}

// Sums the per-tile pose rows of one keyframe ([k][tile][12], contiguous per keyframe) and stores them at
// its unknown indices.  The block walks the keyframe's rows as one flat array with a stride that is a
// multiple of the row length, so a thread always sees the same column and loads are fully coalesced; the
// 21 per-thread sums of a column are then added in a fixed order (deterministic).
// mode 0: init (r, M get rows 0-5 / 6-11); mode 1: step1 (g gets rows 0-5).
constexpr int kPcgPoseReduceThreads = 21 * kPcgPoseRow;   // 252
__global__ __launch_bounds__(kPcgPoseReduceThreads) void pcg_pose_reduce_kernel(const float* __restrict__ partial_pose, int tiles, int kf_count,
                                                                               const KfDev* __restrict__ kfs, PcgParams P, int mode) {
  const int k = blockIdx.x;
  const uint32_t kf_idx = kf_pose_unknown_index(P.gauge_kf, kfs[k].id);
  if (kf_idx == kInvalidUnknown) return;
  __shared__ float sm[21][kPcgPoseRow];
  const int col = threadIdx.x % kPcgPoseRow, sub = threadIdx.x / kPcgPoseRow;
  const float* base = partial_pose + (size_t)k * tiles * kPcgPoseRow;
  const size_t n = (size_t)tiles * kPcgPoseRow;
  float v = 0.f;
  for (size_t e = threadIdx.x; e < n; e += kPcgPoseReduceThreads) v += base[e];
  sm[sub][col] = v;
  __syncthreads();
  if (threadIdx.x >= kPcgPoseRow) return;
  float total = 0.f;
  for (int i = 0; i < 21; ++i) total += sm[i][col];
  if (mode == 0) {
    if (col < 6) P.r[kf_idx + col] = total; else P.M[kf_idx + col - 6] = total;
  } else {
    if (col < 6) P.g[kf_idx + col] = total;
  }
}

// Rounds the fp64 cell sums into the unknown vectors (mode 0: r, M; mode 1: g) and leaves the scratch zero.
__global__ __launch_bounds__(256) void pcg_cf_flush_kernel(PcgParams P, int mode) {
  const uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= P.cf_cells) return;
  const uint32_t idx = P.depth_intr_start + 5 + cell;
  if (mode == 0) {
    P.r[idx] += (float)P.cf_acc0[cell];
    P.M[idx] += (float)P.cf_acc1[cell];
    P.cf_acc1[cell] = 0.0;
  } else {
    P.g[idx] += (float)P.cf_acc0[cell];
  }
  P.cf_acc0[cell] = 0.0;
}

// Sums the per-tile global rows.  mode 0: init -> r, M of the intrinsics; mode 1: step1 -> alpha_d and g.
// Adds to the destination (cfactor-independent entries were zeroed by the caller's memset).
constexpr int kPcgGlobReduceThreads = 50 * kPcgGlobRow;   // 1000
__global__ __launch_bounds__(kPcgGlobReduceThreads) void pcg_glob_reduce_kernel(const float* __restrict__ partial_glob, int tiles, PcgParams P, int mode,
                                                                               float* __restrict__ pairs_out) {
  // same flat, coalesced walk as pcg_pose_reduce_kernel (one block: the rows are ~100 KB)
  __shared__ float sm[50][kPcgGlobRow];
  const int col = threadIdx.x % kPcgGlobRow, sub = threadIdx.x / kPcgGlobRow;
  const size_t n = (size_t)tiles * kPcgGlobRow;
  float part = 0.f;
  for (size_t e = threadIdx.x; e < n; e += kPcgGlobReduceThreads) part += partial_glob[e];
  sm[sub][col] = part;
  __syncthreads();
  if (threadIdx.x >= kPcgGlobRow) return;
  float v = 0.f;
  for (int i = 0; i < 50; ++i) v += sm[i][col];
  if (mode == 0) {
    if (P.optimize_depth_intr) {
      if (col < 5) P.r[P.depth_intr_start + col] += v;
      else if (col < 10) P.M[P.depth_intr_start + col - 5] += v;
    }
    if (P.optimize_color_intr) {
      if (col >= 10 && col < 14) P.r[P.color_intr_start + col - 10] += v;
      else if (col >= 14 && col < 18) P.M[P.color_intr_start + col - 14] += v;
    }
  } else {
    if (col == 0) {
      // sum over this rank's pairs; pcg_alpha_d_kernel adds the epsilon terms (after the all-reduce, if any)
      *pairs_out = v;
    } else if (col < 6) {
      if (P.optimize_depth_intr) P.g[P.depth_intr_start + col - 1] += v;
    } else if (col < 10) {
      if (P.optimize_color_intr) P.g[P.color_intr_start + col - 6] += v;
    }
  }
}

template <bool kDepth, bool kDesc, bool kIntr>
__global__ __launch_bounds__(kPcgThreads) __attribute__((amdgpu_waves_per_eu(kDesc ? BSLAM_PCG_STEP1_WAVES_DESC : 4))) void pcg_step1_kernel(
    CamConsts c_in, const KfDev* __restrict__ kfs, int kf_count, Schedule sc, SurfelRows s, PcgParams P,
    float* __restrict__ partial_pose, float* __restrict__ partial_glob) {
  CamConsts c = c_in;
  constexpr int R = pcg_surfels_per_thread(kDesc, kIntr);
  uint32_t slot;
  if (!slot_of_block(sc, blockIdx.x, &slot)) return;
  const int tile = (int)slot;
  __shared__ PcgRowStash stash;
  __shared__ float redg[4][32];

  f3 gp[R], gn[R];
  bool valid[R];
  uint32_t idx[R];
  __shared__ float state[kDesc ? kPcgStateComps * R * kPcgThreads : 1];
  float ps[R][3], ag[R][3];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t i = surfel_of_slot(sc, slot, r, R);
    valid[r] = i < s.size;
    idx[r] = valid[r] ? i : 0;
    gp[r] = mk3(s.x[idx[r]], s.y[idx[r]], s.z[idx[r]]);
    gn[r] = unpack_normal(s.normal[idx[r]]);
#pragma unroll
    for (int j = 0; j < 3; ++j) { ps[r][j] = 0.f; ag[r][j] = 0.f; }
    if (P.optimize_geometry) {
      const uint32_t base = P.surfel_start + (uint32_t)P.per_surfel * (P.perm ? P.perm[idx[r]] : idx[r]);
      ps[r][0] = P.p[base];
      if (kDesc) { ps[r][1] = P.p[base + 1]; ps[r][2] = P.p[base + 2]; }
    }
    if constexpr (kDesc) {
      f3 tp1, tp2;
      tangent_points(gp[r], gn[r], s.radius_squared[idx[r]], &tp1, &tp2);
      BSLAM_PCG_ST(r, 0) = gn[r].x; BSLAM_PCG_ST(r, 1) = gn[r].y; BSLAM_PCG_ST(r, 2) = gn[r].z;
      BSLAM_PCG_ST(r, 3) = tp1.x; BSLAM_PCG_ST(r, 4) = tp1.y; BSLAM_PCG_ST(r, 5) = tp1.z;
      BSLAM_PCG_ST(r, 6) = tp2.x; BSLAM_PCG_ST(r, 7) = tp2.y; BSLAM_PCG_ST(r, 8) = tp2.z;
      BSLAM_PCG_ST(r, 9) = s.d1[idx[r]]; BSLAM_PCG_ST(r, 10) = s.d2[idx[r]];
      BSLAM_PCG_ST(r, 11) = ps[r][0]; BSLAM_PCG_ST(r, 12) = ps[r][1]; BSLAM_PCG_ST(r, 13) = ps[r][2];
    }
  }
  float glob[kPcgGlobRow];
#pragma unroll
  for (int i = 0; i < kPcgGlobRow; ++i) glob[i] = 0.f;
  float pdi[5] = {0, 0, 0, 0, 0}, pci[4] = {0, 0, 0, 0};
  if (kIntr && P.optimize_depth_intr) for (int j = 0; j < 5; ++j) pdi[j] = P.p[P.depth_intr_start + j];
  if (kIntr && P.optimize_color_intr) for (int j = 0; j < 4; ++j) pci[j] = P.p[P.color_intr_start + j];

  if constexpr (kDesc) BSLAM_HOIST_CAM_CENTRES(c);
  else BSLAM_HOIST_DEPTH_CAM_CENTRE(c);
  BSLAM_HOIST_UNPROJECTION_CENTRE(c);
  int stashed = 0;   // keyframes in the current stash (uniform)
  int buf = 0;
  for (int k0 = 0; k0 < kf_count; k0 += 64) {
  unsigned long long todo = keyframes_to_visit(c, kfs, k0, kf_count, sc, slot, R, true);
  if (P.optimize_poses && sc.bounds != nullptr) pcg_zero_rows(todo, k0, kf_count, partial_pose, sc.slots, tile);
  for (; todo != 0; todo &= todo - 1) {
    const int k = k0 + __builtin_ctzll(todo);
    KfDev kf = kfs[k];
    BSLAM_HOIST_KF_TRANSLATION(kf);
    const uint32_t kf_idx = kf_pose_unknown_index(P.gauge_kf, kf.id);
    const bool opt_pose = P.optimize_poses && kf_idx != kInvalidUnknown;
    float pp[6] = {0, 0, 0, 0, 0, 0};
    if (opt_pose) for (int j = 0; j < 6; ++j) pp[j] = P.p[kf_idx + j];
    float pose[kPcgPoseRow];   // 6 live columns
#pragma unroll
    for (int i = 0; i < kPcgPoseRow; ++i) {
      if (i < 6) BSLAM_ZERO(pose[i]);   // independent zeros: see pose_accumulate_kernel
      else pose[i] = 0.f;
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
      Proj p;
      DescSamples ds;
      f2 color_pxy, t1, t2;   // the three sample positions of the descriptor residual
      bool has_desc = false;
      if constexpr (kDesc) {
        if (!valid[r] || !project_to_pixel(c, kf, gp[r], &p)) continue;
        const PixelRecord rec = load_record(c, kf, p);
        has_desc = depth_to_color_pxy_in_bounds(c, p.pxy, &color_pxy);
        ds = descriptor_terms_issue(c, kf, mk3(BSLAM_PCG_ST(r, 3), BSLAM_PCG_ST(r, 4), BSLAM_PCG_ST(r, 5)),
                                    mk3(BSLAM_PCG_ST(r, 6), BSLAM_PCG_ST(r, 7), BSLAM_PCG_ST(r, 8)), color_pxy, &t1, &t2);
        asm volatile("" ::: "memory");   // the gathers stay in front of the branches of the association test
        if (!associate_with_record(c, kf, mk3(BSLAM_PCG_ST(r, 0), BSLAM_PCG_ST(r, 1), BSLAM_PCG_ST(r, 2)), rec, &p)) continue;
      } else {
        if (!valid[r] || !project_and_associate(c, kf, gp[r], gn[r], &p)) continue;
      }
      bool visible = true;
      const f3 rn = p.n_local;
      if (kDepth) {
#pragma clang fp contract(fast)   // past the association test nothing feeds an integer output: products with p and the sums fuse (as nvcc's default does)
        const float inv_stddev = depth_inv_stddev(p.nx, p.ny, p.depth, rn, c.baseline_fx);
        const f3 lu = mk3(p.depth * p.nx, p.depth * p.ny, p.depth);   // unproject(c, p.px, p.py, p.depth)
        const float raw = depth_residual(inv_stddev, rn, lu, p.local);
        const float weight = depth_weight(raw);
        float sum = 0;
        float gj = 0;
        float J[6] = {0, 0, 0, 0, 0, 0};
        if (P.optimize_geometry) { gj = depth_position_jacobian(inv_stddev); sum += gj * (kDesc ? BSLAM_PCG_ST(r, 11) : ps[r][0]); }
        if (opt_pose) {
          depth_pose_jacobian(inv_stddev, rn, lu, J);
#pragma unroll
          for (int j = 0; j < 6; ++j) sum += J[j] * pp[j];
        }
        DepthIntrinsicsTerms t;
        t.valid = false;
        if (kIntr && P.optimize_depth_intr) {
          t = depth_intrinsics_terms(c, kf, p, kDesc ? mk3(BSLAM_PCG_ST(r, 0), BSLAM_PCG_ST(r, 1), BSLAM_PCG_ST(r, 2)) : gn[r], inv_stddev, P.depth_intr_start);
          if (t.valid) {
            sum += t.d[2] * pdi[2];
            sum += t.d[3] * pdi[3];
            sum += t.d[0] * pdi[0];
            sum += t.d[1] * pdi[1];
            sum += t.d[4] * pdi[4];
            sum += t.cf_jac * P.p[t.cf_index];
          }
        }
        glob[0] += sum * weight * sum;
        sum *= weight;
        if (P.optimize_geometry) ag[r][0] += gj * sum;
        if (opt_pose) {
#pragma unroll
          for (int j = 0; j < 6; ++j) pose[j] += J[j] * sum;
        }
        if (kIntr && P.optimize_depth_intr && t.valid) {
#pragma unroll
          for (int j = 0; j < 5; ++j) glob[1 + j] += t.d[j] * sum;
          atomicAdd(&P.cf_acc0[t.cf_index - (P.depth_intr_start + 5)], (double)(t.cf_jac * sum));
        }
      }
      if (kDesc) {
#pragma clang fp contract(fast)
        visible = visible && has_desc;
        if (!visible) continue;
        const DescTerms t = descriptor_terms_finish(c, kf, ds, BSLAM_PCG_ST(r, 9), BSLAM_PCG_ST(r, 10), [&](f2 (&pts)[3]) { pts[0] = color_pxy; pts[1] = t1; pts[2] = t2; });
        const f3 ls = p.local;
        float sum_1 = 0, sum_2 = 0, gj1 = 0, gj2 = 0;
        float J1[6] = {0, 0, 0, 0, 0, 0}, J2[6] = {0, 0, 0, 0, 0, 0};
        float Jc1[4] = {0, 0, 0, 0}, Jc2[4] = {0, 0, 0, 0};
        if (P.optimize_geometry) {
          gj1 = descriptor_position_jacobian(t.gx1, t.gy1, 1.f, 1.f, rn, ls);
          gj2 = descriptor_position_jacobian(t.gx2, t.gy2, 1.f, 1.f, rn, ls);
          const float ps0 = BSLAM_PCG_ST(r, 11);
          sum_1 += gj1 * ps0;
          sum_2 += gj2 * ps0;
          sum_1 += -1.f * BSLAM_PCG_ST(r, 12);
          sum_2 += -1.f * BSLAM_PCG_ST(r, 13);
        }
        if (opt_pose) {
          descriptor_pose_jacobian(t.gx1, t.gy1, ls, J1);
          descriptor_pose_jacobian(t.gx2, t.gy2, ls, J2);
#pragma unroll
          for (int j = 0; j < 6; ++j) { sum_1 += J1[j] * pp[j]; sum_2 += J2[j] * pp[j]; }
        }
        if (kIntr && P.optimize_color_intr) {
          const float gx_1 = t.gx1 / c.cfx, gy_1 = t.gy1 / c.cfy, gx_2 = t.gx2 / c.cfx, gy_2 = t.gy2 / c.cfy;
          const float nx = nx_of(c, (float)p.px), ny = ny_of(c, (float)p.py);
          color_intrinsics_jacobian(gx_1, gy_1, nx, ny, Jc1);
          color_intrinsics_jacobian(gx_2, gy_2, nx, ny, Jc2);
#pragma unroll
          for (int j = 0; j < 4; ++j) { sum_1 += Jc1[j] * pci[j]; sum_2 += Jc2[j] * pci[j]; }
        }
        glob[0] += sum_1 * t.w1 * sum_1 + sum_2 * t.w2 * sum_2;
        sum_1 *= t.w1;
        sum_2 *= t.w2;
        if (P.optimize_geometry) {
          ag[r][0] += gj1 * sum_1 + gj2 * sum_2;
          ag[r][1] += -1.f * sum_1 + 0.f * sum_2;
          ag[r][2] += 0.f * sum_1 + -1.f * sum_2;
        }
        if (opt_pose) {
#pragma unroll
          for (int j = 0; j < 6; ++j) pose[j] += J1[j] * sum_1 + J2[j] * sum_2;
        }
        if (kIntr && P.optimize_color_intr) {
#pragma unroll
          for (int j = 0; j < 4; ++j) glob[6 + j] += Jc1[j] * sum_1 + Jc2[j] * sum_2;
        }
      }
    }

    if (opt_pose) {   // uniform
      pcg_stash_wave_sums<6>(stash, buf, stashed, k, pose);
      if (++stashed == kPcgGroup) { pcg_flush_rows(stash, buf, stashed, partial_pose, sc.slots, tile); stashed = 0; buf ^= 1; }
    }
  }
  }
  if (stashed) pcg_flush_rows(stash, buf, stashed, partial_pose, sc.slots, tile);

  if (P.optimize_geometry) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!valid[r]) continue;
      const uint32_t base = P.surfel_start + (uint32_t)P.per_surfel * (P.perm ? P.perm[idx[r]] : idx[r]);
      P.g[base] = ag[r][0];
      if (kDesc) { P.g[base + 1] = ag[r][1]; P.g[base + 2] = ag[r][2]; }
    }
  }
  block_reduce_rows<10>(glob, redg, partial_glob + (size_t)tile * kPcgGlobRow);
}

// ---------------------------------------------------------------------------------------------
// vector kernels over the unknowns with deterministic dot products
// ---------------------------------------------------------------------------------------------
constexpr int kVecThreads = 256;
constexpr int kVecPerThread = 4;
constexpr int kVecTile = kVecThreads * kVecPerThread;

__device__ __forceinline__ float diag_extra(uint32_t i, uint32_t a_index) {
  return kDiagEpsilon + ((i == a_index) ? (kAPriorWeight * kAPriorWeight) : 0);
}

// Dot products are formed as (shared part) + (this rank's sharded part).  The vector kernels run on a
// two-segment grid: blocks [0, sb) walk the SHARED unknowns in shared order (poses, then intrinsics / cfactor
// cells), blocks [sb, sb + lb) walk this rank's sharded unknowns.  The shared part is therefore summed in an
// order that does not depend on the shard size, so every rank of a surfel-sharded run forms bit-identical
// shared sums; only the sharded part goes through the all-reduce.
__host__ __device__ __forceinline__ uint32_t vec_shared_blocks(const PcgParams& P) {
  return (P.unknown_count - (P.shard_end - P.shard_begin) + kVecTile - 1) / kVecTile;
}
__host__ __device__ __forceinline__ uint32_t vec_local_blocks(const PcgParams& P) {
  return (P.shard_end - P.shard_begin + kVecTile - 1) / kVecTile;
}
// Unknown index of element j (0..kVecPerThread) of this thread, or false.
__device__ __forceinline__ bool vec_element(const PcgParams& P, int j, uint32_t* i) {
  const uint32_t shard_len = P.shard_end - P.shard_begin, shared = P.unknown_count - shard_len;
  const uint32_t sb = vec_shared_blocks(P);
  if (blockIdx.x < sb) {
    const uint32_t u = blockIdx.x * kVecTile + j * kVecThreads + threadIdx.x;
    if (u >= shared) return false;
    *i = (u < P.shard_begin) ? u : u + shard_len;
  } else {
    const uint32_t u = (blockIdx.x - sb) * kVecTile + j * kVecThreads + threadIdx.x;
    if (u >= shard_len) return false;
    *i = P.shard_begin + u;
  }
  return true;
}
__device__ __forceinline__ void block_sum_to(float v, float* out) {
  __shared__ float sm[kVecThreads / 64];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) *out = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

// PCGInit2CUDAKernel BS/kernel_pcg.cu:564-606: partial[b] = sum r*p
__global__ __launch_bounds__(kVecThreads) void pcg_init2_kernel(PcgParams P, float a, float* __restrict__ partial) {
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < kVecPerThread; ++j) {
    uint32_t i;
    if (vec_element(P, j, &i)) {
      P.g[i] = 0;
      const float r_value = P.r[i] + ((i == P.a_index) ? (-kAPriorWeight * kAPriorWeight * a) : 0);
      const float p_value = r_value / (P.M[i] + kDiagEpsilon + ((i == P.a_index) ? (kAPriorWeight * kAPriorWeight) : 0));
      P.p[i] = p_value;
      P.delta[i] = 0;
      acc += r_value * p_value;
    }
  }
  block_sum_to(acc, &partial[blockIdx.x]);
}

// AddAlphaDEpsilonTermsCUDAKernel BS/kernel_pcg.cu:1027-1048: partial[b] = sum (eps [+100]) p^2
__global__ __launch_bounds__(kVecThreads) void pcg_eps_kernel(PcgParams P, float* __restrict__ partial) {
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < kVecPerThread; ++j) {
    uint32_t i;
    if (vec_element(P, j, &i)) { const float p = P.p[i]; acc += diag_extra(i, P.a_index) * p * p; }
  }
  block_sum_to(acc, &partial[blockIdx.x]);
}

// PCGStep2CUDAKernel BS/kernel_pcg.cu:1116-1170: partial[b] = sum z*r
__global__ __launch_bounds__(kVecThreads) void pcg_step2_kernel(PcgParams P, float* __restrict__ partial) {
  const float ad = *P.alpha_d;
  const float alpha = (ad >= 1e-35f) ? (*P.alpha_n / ad) : 0;
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < kVecPerThread; ++j) {
    uint32_t i;
    if (vec_element(P, j, &i)) {
      const float p_value = P.p[i];
      P.delta[i] += alpha * p_value;
      float r_value = P.r[i];
      r_value -= alpha * (P.g[i] + diag_extra(i, P.a_index) * p_value);
      P.r[i] = r_value;
      const float z_value = r_value / (P.M[i] + kDiagEpsilon + ((i == P.a_index) ? (kAPriorWeight * kAPriorWeight) : 0));
      P.g[i] = z_value;
      acc += z_value * r_value;
    }
  }
  block_sum_to(acc, &partial[blockIdx.x]);
}

// PCGStep3CUDAKernel BS/kernel_pcg.cu:1211-1230
__global__ __launch_bounds__(kVecThreads) void pcg_step3_kernel(PcgParams P) {
  const float an = *P.alpha_n;
  const float beta = (an >= 1e-35f) ? (*P.beta_n / an) : 0;
#pragma unroll
  for (int j = 0; j < kVecPerThread; ++j) {
    uint32_t i;
    if (vec_element(P, j, &i)) P.p[i] = P.g[i] + beta * P.p[i];
  }
}

// Sums the block partials of the shared segment [0, sb) and of the sharded segment [sb, sb + lb) in a fixed
// order, one block.  combine != 0: out[0] = shared + sharded.  combine == 0: out[0] = shared, out[1] = sharded
// (the caller all-reduces out[1] across ranks and adds).
__global__ __launch_bounds__(256) void pcg_final_sum_kernel(const float* __restrict__ partial, int sb, int lb, float* __restrict__ out, int combine) {
  __shared__ float sm[2][256];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < sb; i += 256) a += partial[i];
  for (int i = threadIdx.x; i < lb; i += 256) b += partial[sb + i];
  sm[0][threadIdx.x] = a;
  sm[1][threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sm[0][threadIdx.x] += sm[0][threadIdx.x + s]; sm[1][threadIdx.x] += sm[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (combine) out[0] = sm[0][0] + sm[1][0];
    else { out[0] = sm[0][0]; out[1] = sm[1][0]; }
  }
}

// Multi-rank exchange helpers: the shared unknowns of up to two vectors plus a few scalars travel in one
// staging buffer [vec0 shared | vec1 shared | scalars].
__global__ __launch_bounds__(256) void pcg_pack_shared_kernel(const float* __restrict__ v0, const float* __restrict__ v1, uint32_t n,
                                                              uint32_t shard_begin, uint32_t shard_end, float* __restrict__ stage) {
  const uint32_t shared = n - (shard_end - shard_begin);
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= shared) return;
  const uint32_t i = (j < shard_begin) ? j : j + (shard_end - shard_begin);
  stage[j] = v0[i];
  if (v1) stage[shared + j] = v1[i];
}
__global__ __launch_bounds__(256) void pcg_unpack_shared_kernel(float* __restrict__ v0, float* __restrict__ v1, uint32_t n,
                                                                uint32_t shard_begin, uint32_t shard_end, const float* __restrict__ stage) {
  const uint32_t shared = n - (shard_end - shard_begin);
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= shared) return;
  const uint32_t i = (j < shard_begin) ? j : j + (shard_end - shard_begin);
  v0[i] = stage[j];
  if (v1) v1[i] = stage[shared + j];
}
// out = s[0] + s[1]
__global__ void pcg_add2_kernel(const float* __restrict__ s, float* __restrict__ out) { *out = s[0] + s[1]; }
// alpha_d = pairs + kf_count x (eps_shared + eps_sharded)   (quirk Q7: the epsilon term is added once per keyframe)
__global__ void pcg_alpha_d_kernel(const float* __restrict__ pairs, const float* __restrict__ eps2, int kf_count, float* __restrict__ alpha_d) {
  float a = *pairs;
  const float e = eps2[0] + eps2[1];
  for (int k = 0; k < kf_count; ++k) a += e;
  *alpha_d = a;
}

// UpdateSurfelsFromPCGDeltaCUDAKernel BS/kernel_pcg.cu:1305-1333
template <bool kDesc>
__global__ __launch_bounds__(256) void pcg_update_surfels_kernel(float* x, float* y, float* z, const uint32_t* normal, float* d1, float* d2,
                                                                 uint32_t size, uint32_t s0, const float* __restrict__ delta) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= size) return;
  const float t = delta[s0 + (kDesc ? 3 : 1) * i];
  if (t != 0) {
    const f3 n = unpack_normal(normal[i]);
    x[i] = x[i] + t * n.x;
    y[i] = y[i] + t * n.y;
    z[i] = z[i] + t * n.z;
  }
  if (kDesc) {
    float a = d1[i];
    a += delta[s0 + 3 * i + 1];
    d1[i] = fmaxf(-180.f, fminf(180.f, a));
    float b = d2[i];
    b += delta[s0 + 3 * i + 2];
    d2[i] = fmaxf(-180.f, fminf(180.f, b));
  }
}

// UpdateCFactorsFromPCGDeltaCUDAKernel BS/kernel_pcg.cu:1361-1372
__global__ __launch_bounds__(256) void pcg_update_cfactors_kernel(uint8_t* cf, uint32_t pitch, int w, int h, uint32_t start, const float* __restrict__ delta) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint32_t)(w * h)) return;
  const uint32_t yy = i / (uint32_t)w, xx = i - yy * (uint32_t)w;
  float* p = (float*)(cf + (size_t)yy * pitch) + xx;
  *p += delta[start + i];
}

}  // namespace bslam
