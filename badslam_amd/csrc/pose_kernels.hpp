// pose_kernels.hpp -- pose Gauss-Newton kernels (gfx950).
//
// Replaces AccumulatePoseEstimationCoeffsCUDAKernel + AccumulateGaussNewtonHAndB
// (BS/kernel_opt_pose.cu:251-383, BS/gauss_newton.cuh:47-95) and the host part of
// DirectBA::EstimateFramePose (BS/direct_ba_alternating.cc:130-233).
//
// MI355X design (not the reference's one-launch-per-keyframe, 27-block-reductions-per-launch
// shape):
//   * one launch covers (surfel tiles) x (keyframe chunks); a thread keeps kR surfels (position +
//     decoded normal in registers; the photometric variant's 14 per-surfel constants in LDS) and
//     walks the keyframes of its chunk, so surfel bytes are read once per chunk instead of once
//     per keyframe; the surfels come in the library's per-surfel Morton order (sorted copy of the
//     rows, badslam_hip.hip: prepare_surfels), so a wave's 64 surfels are a compact blob;
//   * the 21 + 6 (+ cost) coefficients are accumulated per thread over its kR surfels (fused
//     multiply-adds: these sums are compared at 1e-4, only the association predicates need
//     bit-exact arithmetic), reduced across the wave through a wave-private LDS tile (8 columns
//     per round: ds_write / ds_read_b128 / three DPP adds, no barrier) and the four waves' rows
//     meet in an LDS stash, so that ONE 32-float row per (work slot, keyframe) leaves the block,
//     with one barrier per four visited keyframes; a second kernel sums the rows in a fixed
//     order: deterministic, no float atomics;
//   * a block decides up front which keyframes of its chunk its surfels can be seen from at all
//     (block-level frustum culling, device_math.hpp) and visits only those;
//   * the 6x6 solve, SE3 update and convergence test run on the device (one thread per
//     keyframe), all keyframes advance in lock-step.
#pragma once

#include "device_math.hpp"

namespace bslam {

constexpr int kPoseThreads = 256;
// surfels per thread (granules per work slot): geometric-only and photometric variants are tuned
// separately, the photometric one is register-heavy (tools/variants.py measures the choices)
#ifndef BSLAM_POSE_R_GEO
#define BSLAM_POSE_R_GEO 4
#endif
#ifndef BSLAM_POSE_R_DESC
#define BSLAM_POSE_R_DESC 2
#endif
constexpr int kPoseRGeo = BSLAM_POSE_R_GEO;
constexpr int kPoseRDesc = BSLAM_POSE_R_DESC;
// Large geometry-only problems amortise the per-keyframe wave reduction over 6 surfels per thread at 5 waves per SIMD (K = 200,
// S = 3.84 M: 1256 -> 1200 us per launch); small ones need the blocks (K = 50, S = 0.96 M: 97.5 us with 4, 98.3 with 6).
#ifndef BSLAM_POSE_R_GEO_LARGE
#define BSLAM_POSE_R_GEO_LARGE 6
#endif
constexpr int kPoseRGeoLarge = BSLAM_POSE_R_GEO_LARGE;
constexpr uint32_t kPoseLargeSurfels = 2000000u;
inline int pose_surfels_per_thread(bool use_desc, uint32_t surfels_size) {
  return use_desc ? kPoseRDesc : (surfels_size >= kPoseLargeSurfels ? kPoseRGeoLarge : kPoseRGeo);
}
constexpr int kRow = 32;                       // floats per partial row: 21 H, 6 b, cost, count bits, pad
constexpr int kRowCost = 27;
constexpr int kRowCount = 28;

struct SurfelRows {
  const float* x; const float* y; const float* z;
  const uint32_t* normal;
  const float* radius_squared;
  const float* d1; const float* d2;
  uint32_t size;
};

struct PoseState {
  float q[4];
  float t[3];
  int converged;     // 1 = no further iterations
  int iterations;
  int pad[3];
};

__device__ __forceinline__ void accumulate_h_b(float raw, float w, const float* J, float* acc) {
  // BS/gauss_newton.cuh:60-91: H[idx] += (w * J[row]) * J[col]; b[i] += (w * raw) * J[i]
  int idx = 0;
#pragma unroll
  for (int row = 0; row < 6; ++row) {
    const float wj = w * J[row];
#pragma unroll
    for (int col = row; col < 6; ++col) {
      acc[idx] = __builtin_fmaf(wj, J[col], acc[idx]);
      ++idx;
    }
  }
  const float wr = w * raw;
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[21 + i] = __builtin_fmaf(wr, J[i], acc[21 + i]);
}

// Occupancy target handed to the register allocator: without it the 32 accumulators' zero initialisation lands in a
// second 32-register tuple (104 VGPRs, 4 waves per SIMD); with it the geometric kernel fits 76 VGPRs (6 waves).
// The photometric variants fit 95 VGPRs (5 waves) with their per-surfel constants in LDS.
#ifndef BSLAM_POSE_WAVES_GEO
#define BSLAM_POSE_WAVES_GEO 6
#endif
#ifndef BSLAM_POSE_WAVES_DESC
#define BSLAM_POSE_WAVES_DESC 5
#endif
#define BSLAM_POSE_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(kDesc ? BSLAM_POSE_WAVES_DESC : ((kPoseR > 4 || kCost) ? 5 : BSLAM_POSE_WAVES_GEO))))
// Wave reduction of a row through LDS, kRedCols columns per round (wave_column_sums_lds; tile: kRedCols x 64 floats per wave).
// Measured against the transposing butterfly (wave_transpose_sum32, still used by the image-pair kernels) on one box: photometric
// K = 300 pose kernel 14.90 -> 14.37 ms with 8 columns per round (4: no gain; 16: the tile costs the photometric kernel a
// block per CU), geometry-only K = 50 / 200: 94.6 -> 90.8 us / 1206 -> 1160 us.
#ifndef BSLAM_POSE_REDUCE_COLS_DESC
#define BSLAM_POSE_REDUCE_COLS_DESC 4
#endif
#ifndef BSLAM_POSE_STASH_GROUP_DESC
#define BSLAM_POSE_STASH_GROUP_DESC 2
#endif
// columns per round / stashed keyframes per barrier: 8 / 4 for the geometry-only kernels; the photometric ones, whose 24 KB of
// per-surfel constants already sit in LDS, take 4 / 2 so that tiles (4 KB) and stash (2 KB) leave FIVE workgroups per CU
// (30 KB; 32 KB already means four): K = 300 dense 13.47 -> 13.27 ms, trajectory stack 1252 -> 1203 us, survey 4.40 -> 4.28 ms
constexpr int kRedColsGeo = 8, kRedColsDesc = BSLAM_POSE_REDUCE_COLS_DESC;
// One partial row per (work slot, keyframe): the four waves' rows meet in an LDS stash, one barrier per kPoseStashGroup visited
// keyframes (two stashes alternate, so a wave that runs ahead never overwrites rows that are still being added; it cannot get
// two groups ahead: the barrier).  A quarter of the row traffic of one row per wave: the row sums of a batched Gauss-Newton
// iteration at K = 300 go from 428 to about 110 us.
constexpr int kPoseRowsPerSlot = 1;
typedef unsigned long long VisWord;   // visit word of a (chunk, work slot): one bit per keyframe of the chunk (<= 64)
constexpr int kPoseStashGroupGeo = 4, kPoseStashGroupDesc = BSLAM_POSE_STASH_GROUP_DESC;
// kCost: the robust cost (column kRowCost) is wanted -- only the per-keyframe debug entry point returns it; the batched
// Gauss-Newton loop never does, and leaves the tukey / huber residual evaluations out.
template <bool kDepth, bool kDesc, int kPoseR, bool kCost>
__global__ __launch_bounds__(kPoseThreads) BSLAM_POSE_WAVES_ATTR void pose_accumulate_kernel(
    CamConsts c_in, const KfDev* __restrict__ kfs, int kf_count, int kfs_per_block, Schedule sc,
    SurfelRows s, float* __restrict__ partials, int rows_per_kf, const PoseState* __restrict__ states, VisWord* __restrict__ vis,
    const int* __restrict__ kf_list) {
  CamConsts c = c_in;
  // kf_list (batched Gauss-Newton loop on long keyframe lists): {n, list[n]} = the keyframes that are still unconverged, in
  // ascending order (pose_active_list_kernel).  The chunks of a launch then cut that LIST, not the keyframe table, so a late
  // iteration with a handful of stragglers runs one chunk's worth of workgroups instead of every chunk's, each of which would only
  // start, find its keyframes converged and leave.  nullptr: the chunks cut the table itself.
  const int kf_places = kf_list ? kf_list[0] : kf_count;
  // 1-D grid of 8 * slots_per_xcd * chunks blocks: block b -> XCD lane x = b % 8; within an XCD the
  // blocks run chunk-major over that XCD's range of surfel slots.
  const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
  const uint32_t chunk = j / sc.slots_per_xcd, local = j - chunk * sc.slots_per_xcd;
  uint32_t slot;
  if (!slot_of_block(sc, (local << 3) | xcd, &slot)) return;
  const int tile = (int)slot;
  const int kf_begin = (int)chunk * kfs_per_block;
  if (kf_begin >= kf_places) return;   // the host sized the grid by an older (larger) count of unconverged keyframes
  const int kf_end = min(kf_places, kf_begin + kfs_per_block);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // Which keyframes of the chunk (<= 64) this block visits, decided for all of them at once, one keyframe per lane: not
  // converged (late Gauss-Newton iterations: most chunks have nothing left to do) and -- block-level frustum culling -- the
  // bounding box of the slot's surfels reaches into the keyframe's image.  The word goes to vis[chunk][slot]: the row sums
  // (pose_reduce_*_kernel) only read the partial rows of visited (slot, keyframe) pairs, the others are never written.
  unsigned long long todo;
  // decided by the first wave and handed to the others through LDS: for the workgroups that leave here (nine in ten on a
  // trajectory) the three other waves' copies of the test were the larger part of their work (survey-range stack -1.5 %)
  __shared__ unsigned long long todo_shared;
  if (wave == 0) {
    const int place = kf_begin + lane;
    const int k = (kf_list && place < kf_end) ? kf_list[1 + place] : place;
    const bool wanted = states == nullptr || (place < kf_end && !states[k].converged);
    todo = keyframes_to_visit(c, kfs, kf_begin, kf_end, sc, slot, kPoseR, wanted, kf_list ? k : -1);
    if (lane == 0) { todo_shared = todo; vis[(size_t)chunk * sc.slots + slot] = todo; }
  }
  __syncthreads();
  todo = todo_shared;
  if (todo == 0) return;   // leaves before touching the surfels

  // surfels of this thread: tile * kPoseTile + r * kPoseThreads + threadIdx.x (coalesced per r)
  f3 gp[kPoseR], gn[kPoseR];
  bool valid[kPoseR];
  // Photometric variant: 12 per-surfel constants (position, normal, the two tangent sample points of the descriptor
  // residual) live in LDS -- [component][thread]: conflict-free, private to the thread, no barrier -- instead of 24 VGPRs, and
  // are read back where a pair needs them (round 2: 123 -> 95 VGPRs, 518 -> 495 us at K = 50).
  constexpr int kState = 12;
  __shared__ float state[kDesc ? kState * kPoseR * kPoseThreads : 1];
  float desc1[kDesc ? kPoseR : 1], desc2[kDesc ? kPoseR : 1];   // in registers: 24 + 8 + 4 KB of LDS per block leave four blocks per CU
  constexpr int kRedCols = kDesc ? kRedColsDesc : kRedColsGeo;
  constexpr int kPoseStashGroup = kDesc ? kPoseStashGroupDesc : kPoseStashGroupGeo;
  __shared__ float row_stash[2][kPoseStashGroup][kPoseThreads / 64][kRow];
  __shared__ int stash_kf[2][kPoseStashGroup];
  int stashed = 0, stash_buf = 0;   // uniform
  // adds up the four waves' rows of the `n` stashed keyframes -- ((w0 + w1) + w2) + w3 -- and stores one row per keyframe
  auto flush_rows = [&](int n) {
    __syncthreads();
    const int j = threadIdx.x / kRow, col = threadIdx.x % kRow;
    if (j < n) {
      const float (*w)[kRow] = row_stash[stash_buf][j];
      partials[((size_t)stash_kf[stash_buf][j] * rows_per_kf + (size_t)tile) * kRow + col] = ((w[0][col] + w[1][col]) + w[2][col]) + w[3][col];
    }
  };
  static_assert(kPoseStashGroup * kRow <= kPoseThreads && kPoseThreads / 64 == 4, "one thread per stashed column; four waves");
  __shared__ __attribute__((aligned(16))) float red_tile[kPoseThreads / 64][kRedCols * 64];   // wave-private tiles of the row reduction
  auto st = [&](int r, int comp) -> float& { return state[(r * kState + comp) * kPoseThreads + threadIdx.x]; };
#pragma unroll
  for (int r = 0; r < kPoseR; ++r) {
    const uint32_t i = surfel_of_slot(sc, slot, r, kPoseR);
    valid[r] = i < s.size;
    const uint32_t j = valid[r] ? i : 0;
    gp[r] = mk3(s.x[j], s.y[j], s.z[j]);
    gn[r] = unpack_normal(s.normal[j]);
    if constexpr (kDesc) {
      f3 tp1, tp2;
      tangent_points(gp[r], gn[r], s.radius_squared[j], &tp1, &tp2);
      st(r, 0) = gp[r].x; st(r, 1) = gp[r].y; st(r, 2) = gp[r].z;
      st(r, 3) = gn[r].x; st(r, 4) = gn[r].y; st(r, 5) = gn[r].z;
      st(r, 6) = tp1.x; st(r, 7) = tp1.y; st(r, 8) = tp1.z;
      st(r, 9) = tp2.x; st(r, 10) = tp2.y; st(r, 11) = tp2.z;
      desc1[r] = s.d1[j]; desc2[r] = s.d2[j];
    }
  }

  if constexpr (kDesc) BSLAM_HOIST_CAM_CENTRES(c);
  else if constexpr (kPoseR > 4) BSLAM_HOIST_DEPTH_CAM_CENTRE(c);
  // the list entry of the next keyframe to visit is fetched (a scalar load) one visit ahead
  int k_next = kf_begin + __builtin_ctzll(todo);
  if (kf_list) k_next = kf_list[1 + k_next];
  while (todo != 0) {   // uniform
    const int k = k_next;
    todo &= todo - 1;
    if (todo != 0) {
      k_next = kf_begin + __builtin_ctzll(todo);
      if (kf_list) k_next = kf_list[1 + k_next];
    }
    KfDev kf = kfs[k];   // by value: the uniform fields are fetched once per keyframe, ahead of the per-surfel branches
    if constexpr (kDesc || kPoseR > 4) BSLAM_HOIST_KF_TRANSLATION(kf);
    float acc[kRow];
    // Every live accumulator is zeroed by its own opaque instruction.  Written as acc[i] = 0.f the optimiser knows all of
    // them to be one value: it folds the first surfel's fma(wj, J, 0) into a multiply and then has to materialise the zeros a
    // second time, as copies, for the lanes that skip that surfel (62 v_mov per keyframe and thread instead of 27).
#pragma unroll
    for (int i = 0; i < kRow; ++i) {
      if (i < (kCost ? kRowCost + 1 : kRowCost)) BSLAM_ZERO(acc[i]);
      else acc[i] = 0.f;
    }
    // residual count of the wave: formed from ballots at the points where the lanes have reconverged (s_bcnt1 on the mask: no
    // VALU, and one column less in the reduction below); uniform
    uint32_t count = 0;

#pragma unroll
    for (int r = 0; r < kPoseR; ++r) {
      bool got_depth = false, got_desc = false;
      do {
        Proj p;
        DescSamples ds;
        f2 color_pxy, t1, t2;   // the three sample positions of the descriptor residual
        bool has_desc = false;
        if (!valid[r]) break;
        if constexpr (!kDesc) {
          if (!project_and_associate(c, kf, gp[r], gn[r], &p)) break;
        } else {
          // The descriptor samples depend on the surfel and the pose only, not on the pixel record: their three quad gathers
          // are issued together with the record gather, BEFORE the association test (99.6 % of the in-bounds pairs pass it),
          // so a pair waits for one L2 round trip instead of two and the depth residual is evaluated while the quads are in
          // flight (551 -> 517 us at K = 50).  Issued unconditionally -- the quad table's clamp addressing makes every address
          // valid -- so that no control-flow join sits in front of the record's wait (s_waitcnt vmcnt(3), not vmcnt(0)).
          if (!project_to_pixel(c, kf, mk3(st(r, 0), st(r, 1), st(r, 2)), &p)) break;
          const PixelRecord rec = load_record(c, kf, p);
          has_desc = depth_to_color_pxy_in_bounds(c, p.pxy, &color_pxy);
          project_tangent_points(mk3(st(r, 6), st(r, 7), st(r, 8)), mk3(st(r, 9), st(r, 10), st(r, 11)), kf.frame_T_global, c, &t1, &t2);
          ds = descriptor_samples_issue(kf, c, color_pxy, t1, t2);
          asm volatile("" ::: "memory");   // keeps the compiler from sinking the gathers below the branches that follow
          if (!associate_with_record(c, kf, mk3(st(r, 3), st(r, 4), st(r, 5)), rec, &p)) break;
        }
        float J[6];
        float raw;
        if (kDepth) {                                           // BS/kernel_opt_pose.cu:283-317
          depth_residual_and_jacobian(c, p, &raw, J);
          accumulate_h_b(raw, depth_weight(raw), J, acc);
          if constexpr (kCost) acc[kRowCost] += weighted_depth_residual(raw);
          got_depth = true;
        }
        if (kDesc) {                                            // BS/kernel_opt_pose.cu:320-382
          if (has_desc) {
            float r1, rr2, gx1, gy1, gx2, gy2;
            descriptor_samples_finish(kf, c, ds, desc1[r], desc2[r], [&](f2 (&pts)[3]) { pts[0] = color_pxy; pts[1] = t1; pts[2] = t2;
            }, &r1, &rr2, &gx1, &gy1, &gx2, &gy2);
            gx1 *= c.cfx; gx2 *= c.cfx;
            gy1 *= c.cfy; gy2 *= c.cfy;
            descriptor_pose_jacobian(gx1, gy1, p.local, J);
            accumulate_h_b(r1, desc_weight(r1), J, acc);
            descriptor_pose_jacobian(gx2, gy2, p.local, J);
            accumulate_h_b(rr2, desc_weight(rr2), J, acc);
            if constexpr (kCost) acc[kRowCost] += weighted_desc_residual(r1);        // quirk Q1: only the first residual is counted
            got_desc = true;
          }
        }
      } while (false);
      count += (uint32_t)__builtin_popcountll(__ballot(got_depth)) + (uint32_t)__builtin_popcountll(__ballot(got_desc));
    }

    // wave reduction (skipped when the whole wave saw nothing for this keyframe)
    const bool any = count != 0;   // uniform
    float total = 0.f;
    constexpr int kLive = kCost ? kRowCost + 1 : kRowCost;   // 21 H, 6 b (, cost); the count column is filled in below
    int my_col;    // the column this lane ends up with, and whether it is the lane that stores it
    bool writer;
    wave_column_sums_owner<kLive, kRedCols>(&my_col, &writer);
    if (any) total = wave_column_sums_lds<kLive, kRedCols>(acc, red_tile[wave]);
    if (my_col == kRowCount) total = (float)count;   // <= 64 * kPoseR: exact
    if (writer) row_stash[stash_buf][stashed][wave][my_col] = total;
    if (threadIdx.x == 0) stash_kf[stash_buf][stashed] = k;
    if (++stashed == kPoseStashGroup) { flush_rows(stashed); stashed = 0; stash_buf ^= 1; }
  }
  if (stashed) flush_rows(stashed);
}

// bslam_debug_wave_column_sums: wave_column_sums_lds on given values -- one wave, lane l holds in[l][0 .. 32), out[c] = the total
// the lane that owns column c ends up with (columns nobody owns in this configuration stay as the caller set them).
template <int kLive, int kCols>
__global__ __launch_bounds__(64) void wave_column_sums_probe_kernel(const float* __restrict__ in, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float tile[kCols * 64];
  float v[kRow];
#pragma unroll
  for (int i = 0; i < kRow; ++i) v[i] = in[threadIdx.x * kRow + i];
  int col;
  bool writer;
  wave_column_sums_owner<kLive, kCols>(&col, &writer);
  const float total = wave_column_sums_lds<kLive, kCols>(v, tile);
  if (writer) out[col] = total;
}

// Sums the partial rows [k][row][32] of one keyframe in a fixed order, in two stages so that the
// sum is spread over K x kReduceParts workgroups.  Stage A: block (k, part) sums its contiguous
// range of rows (thread t owns column t % 32 and row residue t / 32); stage B adds the parts.
constexpr int kReduceParts = 8;

__global__ __launch_bounds__(256) void pose_reduce_kernel(const float* __restrict__ partials, int rows_per_kf, int kf_count,
                                                           float* __restrict__ parts, const PoseState* __restrict__ states) {
  const int k = blockIdx.x, part = blockIdx.y;
  if (states != nullptr && states[k].converged) return;
  const int col = threadIdx.x & 31;
  const int sub = threadIdx.x >> 5;   // 0..7
  const int per_part = (rows_per_kf + kReduceParts - 1) / kReduceParts;
  const int row_begin = part * per_part, row_end = min(rows_per_kf, row_begin + per_part);
  __shared__ float sm[8][kRow];
  const float* base = partials + (size_t)k * rows_per_kf * kRow;
  float v = 0.f;
  for (int t = row_begin + sub; t < row_end; t += 8) v += base[(size_t)t * kRow + col];
  sm[sub][col] = v;
  __syncthreads();
  if (threadIdx.x < kRow) {
    float total = 0.f;
    for (int i = 0; i < 8; ++i) total += sm[i][col];
    parts[((size_t)k * kReduceParts + part) * kRow + col] = total;
  }
}

// Stage B: coeffs[k][col] = sum over parts; the residual count (per-row counts are small exact
// floats, their sum is formed in integers) leaves as two exactly representable floats (low / high
// 16 bits) so that the row can go through a float all-reduce across GPUs unchanged in meaning.
__global__ __launch_bounds__(64) void pose_reduce_final_kernel(const float* __restrict__ parts, int kf_count, float* __restrict__ coeffs,
                                                              const PoseState* __restrict__ states) {
  const int k = blockIdx.x;
  if (states != nullptr && states[k].converged) return;
  const int col = threadIdx.x;
  if (col >= kRow) return;
  if (col == kRowCount) {
    uint32_t total = 0;
    for (int p = 0; p < kReduceParts; ++p) total += (uint32_t)parts[((size_t)k * kReduceParts + p) * kRow + col];
    coeffs[(size_t)k * kRow + kRowCount] = (float)(total & 0xffffu);
    coeffs[(size_t)k * kRow + kRowCount + 1] = (float)(total >> 16);
  } else if (col != kRowCount + 1) {
    float total = 0.f;
    for (int p = 0; p < kReduceParts; ++p) total += parts[((size_t)k * kReduceParts + p) * kRow + col];
    coeffs[(size_t)k * kRow + col] = total;
  }
}

// Row sums of one keyframe by a 1024-thread block.  Only the rows of work slots that visited the keyframe exist (vis: one word
// per slot, bit = the keyframe's place in its chunk; a slot owns kPoseThreads / 64 consecutive rows).  The block first turns
// the keyframe's bit of every slot's word into a bitmap in LDS (one bit per slot); whole groups of 64 slots (256 rows) that
// did not visit are then skipped with one uniform test -- in Morton order the visiting slots of a keyframe form a few runs.
// Thread (sub, col) adds column `col` of the rows sub, sub + 32, sub + 64, ...: row r always goes to thread r % 32 and to its
// partial sum (r / 32) % 8, whatever was visited, and a row that was not visited would be a row of zeros -- so the sums are the
// same bits with and without culling.  Eight independent partial sums keep eight loads in flight; a row that is skipped
// loads a zero from a fixed address instead of branching around the load.
__device__ const float kZeroFloat = 0.f;
constexpr int kRowsPerSlot = kPoseRowsPerSlot;
constexpr int kSlotsPerPass = 256 / kRowsPerSlot;        // a pass of the 32 x 8 partial sums covers 256 rows
constexpr int kWordsPerPass = (kSlotsPerPass + 63) / 64;   // 1 (four rows per slot) or 4 (one)
__host__ __device__ __forceinline__ size_t visit_map_bytes(int slots) {
  return (size_t)(((slots + kSlotsPerPass - 1) / kSlotsPerPass) * kWordsPerPass) * sizeof(unsigned long long);
}
// Returns the number of visiting slots (valid in every thread after the trailing barrier).
__device__ __forceinline__ uint32_t build_visit_map(const VisWord* __restrict__ vis, uint32_t bit, int slots, unsigned long long* __restrict__ vmap) {
  __shared__ uint32_t visiting;
  if (threadIdx.x == 0) visiting = 0;
  __syncthreads();
  const int padded = (int)(visit_map_bytes(slots) / sizeof(unsigned long long)) * 64;
  uint32_t mine = 0;
  for (int s0 = 0; s0 < padded; s0 += (int)blockDim.x) {
    const int sl = s0 + (int)threadIdx.x;
    const bool v = sl < slots && ((vis[sl] >> bit) & 1ull);
    const unsigned long long w = __ballot(v);
    if ((threadIdx.x & 63u) == 0 && sl < padded) { vmap[sl >> 6] = w; mine += (uint32_t)__builtin_popcountll(w); }
  }
  if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(&visiting, mine);
  __syncthreads();
  return visiting;
}
__device__ __forceinline__ float column_share_of_rows(const float* __restrict__ base, int sub, int rows, const unsigned long long* __restrict__ vmap) {
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* zero = &kZeroFloat;
  for (int pass = 0; pass * 256 < rows; ++pass) {
    const unsigned long long* w = vmap + pass * kWordsPerPass;   // the slots whose rows this pass holds: uniform
    unsigned long long any = w[0];
#pragma unroll
    for (int q = 1; q < kWordsPerPass; ++q) any |= w[q];
    if (any == 0) continue;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int in_pass = sub + 32 * u;
      const int r = pass * 256 + in_pass;
      const int sl = in_pass / kRowsPerSlot;
      const bool take = r < rows && ((w[sl >> 6] >> (sl & 63)) & 1ull);
      const float* p = take ? base + (size_t)r * kRow : zero;
      v[u] += *p;
    }
  }
  return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

// Both stages in one launch for the batched loop with an all-reduce hook: block k (1024 threads) sums keyframe k's
// rows coalesced in the same fixed order as pose_reduce_solve_kernel and writes the coefficient row that goes through
// the exchange (count as two exact 16-bit halves, as above).  Rows of converged keyframes are written as zeros, so
// that repeated in-place all-reduces never grow stale values.
__global__ __launch_bounds__(1024) void pose_reduce_rows_kernel(const float* __restrict__ partials, int rows_per_kf, int kf_count,
                                                               float* __restrict__ coeffs, const PoseState* __restrict__ states,
                                                               const VisWord* __restrict__ vis, int kfs_per_block, unsigned long long* __restrict__ stats,
                                                               const int* __restrict__ kf_pos) {
  const int k = blockIdx.x;
  if (states != nullptr && states[k].converged) {
    if (threadIdx.x < kRow) coeffs[(size_t)k * kRow + threadIdx.x] = 0.f;
    return;
  }
  __shared__ float sm[32][kRow];
  extern __shared__ unsigned long long vmap[];   // visit_map_bytes(slots)
  const int col = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int slots = rows_per_kf / kRowsPerSlot;
  const int place = kf_pos ? kf_pos[k] : k;   // the keyframe's place in the list the accumulation walked (its chunk and bit)
  const uint32_t visiting = build_visit_map(vis + (size_t)(place / kfs_per_block) * slots, (uint32_t)(place % kfs_per_block), slots, vmap);
  if (stats != nullptr && threadIdx.x == 0) { atomicAdd(&stats[0], (unsigned long long)slots); atomicAdd(&stats[1], (unsigned long long)visiting); }
  sm[sub][col] = column_share_of_rows(partials + (size_t)k * rows_per_kf * kRow + col, sub, rows_per_kf, vmap);   // the count column: <= 256 per row, exact in fp32 up to 65k rows
  __syncthreads();
  if (threadIdx.x >= kRow) return;
  if (col == kRowCount) {
    uint32_t total = 0;
    for (int i = 0; i < 32; ++i) total += (uint32_t)sm[i][col];
    coeffs[(size_t)k * kRow + kRowCount] = (float)(total & 0xffffu);
    coeffs[(size_t)k * kRow + kRowCount + 1] = (float)(total >> 16);
  } else if (col != kRowCount + 1) {
    float total = 0.f;
    for (int i = 0; i < 32; ++i) total += sm[i][col];
    coeffs[(size_t)k * kRow + col] = total;
  }
}

// ---------------------------------------------------------------------------------------------
// SE3 (Sophus::SE3f semantics: unit quaternion + translation, fp32) and the 6x6 solve
// ---------------------------------------------------------------------------------------------
struct Quat { float x, y, z, w; };

__device__ __forceinline__ Quat qmul(Quat a, Quat b) {   // Eigen quat_product
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
__device__ __forceinline__ f3 qrot(Quat q, f3 v) {       // Eigen QuaternionBase::_transformVector
  f3 qv = mk3(q.x, q.y, q.z);
  f3 uv = mk3(qv.y * v.z - qv.z * v.y, qv.z * v.x - qv.x * v.z, qv.x * v.y - qv.y * v.x);
  uv = add3(uv, uv);
  f3 cx = mk3(qv.y * uv.z - qv.z * uv.y, qv.z * uv.x - qv.x * uv.z, qv.x * uv.y - qv.y * uv.x);
  return mk3(v.x + q.w * uv.x + cx.x, v.y + q.w * uv.y + cx.y, v.z + q.w * uv.z + cx.z);
}
__device__ __forceinline__ void qmat(Quat q, float* R) { // Eigen QuaternionBase::toRotationMatrix
  const float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
  R[3] = txy + twz;          R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = 1.0f - (txx + tyy);
}

// SE3::exp (sophus/se3.hpp:293-313, so3.hpp:282-318)
__device__ inline void se3_exp(const float* a, Quat* q_out, f3* t_out) {
  const float kEps = 1e-5f;
  const f3 omega = mk3(a[3], a[4], a[5]);
  const float theta_sq = sqlen(omega);
  const float theta = sqrtf(theta_sq);
  const float half_theta = 0.5f * theta;
  float imag_factor, real_factor;
  if (theta < kEps) {
    const float theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * theta_po4;
    real_factor = 1.f - 0.5f * theta_sq + (float)(1.0 / 384.0) * theta_po4;
  } else {
    imag_factor = sinf(half_theta) / theta;
    real_factor = cosf(half_theta);
  }
  Quat q{imag_factor * omega.x, imag_factor * omega.y, imag_factor * omega.z, real_factor};
  const float O[9] = {0.f, -omega.z, omega.y, omega.z, 0.f, -omega.x, -omega.y, omega.x, 0.f};
  float O2[9];
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) O2[3 * r + cc] = O[3 * r + 0] * O[0 + cc] + O[3 * r + 1] * O[3 + cc] + O[3 * r + 2] * O[6 + cc];
  float V[9];
  if (theta < kEps) {
    qmat(q, V);
  } else {
    const float c1 = (1.f - cosf(theta)) / (theta_sq);
    const float c2 = (theta - sinf(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; ++i) {
      const float id = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
      V[i] = (id + c1 * O[i]) + c2 * O2[i];
    }
  }
  *q_out = q;
  *t_out = mk3(V[0] * a[0] + V[1] * a[1] + V[2] * a[2], V[3] * a[0] + V[4] * a[1] + V[5] * a[2], V[6] * a[0] + V[7] * a[1] + V[8] * a[2]);
}

// T <- T * D (SE3Base::operator*=, SO3Base::operator*= with the first-order renormalisation)
__device__ inline void se3_mul_inplace(Quat* q, f3* t, Quat dq, f3 dt) {
  const f3 rt = qrot(*q, dt);
  *t = mk3(t->x + rt.x, t->y + rt.y, t->z + rt.z);
  Quat r = qmul(*q, dq);
  const float sn = r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
  if (sn != 1.0f) {
    const float f = 2.0f / (1.0f + sn);
    r.x *= f; r.y *= f; r.z *= f; r.w *= f;
  }
  *q = r;
}

// frame_T_global = global_T_frame.inverse().matrix3x4()
__device__ inline void se3_inverse_matrix(Quat q, f3 t, float* m /*12*/) {
  const Quat qi{-q.x, -q.y, -q.z, q.w};
  const f3 ti = qrot(qi, mk3(t.x * -1.f, t.y * -1.f, t.z * -1.f));
  float R[9];
  qmat(qi, R);
  for (int r = 0; r < 3; ++r) { m[4 * r] = R[3 * r]; m[4 * r + 1] = R[3 * r + 1]; m[4 * r + 2] = R[3 * r + 2]; }
  m[3] = ti.x; m[7] = ti.y; m[11] = ti.z;
}

// 6x6 symmetric solve in double: LDL^T with diagonal pivoting (what Eigen's
// selfadjointView<Upper>().ldlt().solve() does at BS/direct_ba_alternating.cc:206).
// Every loop below is unrolled to constant indices, so A, y and the transpositions live in registers; the dynamic pivot
// index only selects which (constant-index) swap runs.  One thread per keyframe executes this as a serial fp64 chain:
// with the workspace in LDS every dependent access cost ~100 cycles (17 us per Gauss-Newton iteration at K = 50),
// in registers ~8.
__device__ __forceinline__ void swap_d(double& a, double& b) { const double t = a; a = b; b = t; }

__device__ inline void solve_ldlt6(const float* H_upper, const float* b, float* x) {
  constexpr int n = 6;
  double A[n][n];
  {
    int idx = 0;
#pragma unroll
    for (int r = 0; r < n; ++r)
#pragma unroll
      for (int cc = r; cc < n; ++cc) { A[r][cc] = (double)H_upper[idx]; A[cc][r] = (double)H_upper[idx]; ++idx; }
  }
  int perm[n];
#pragma unroll
  for (int k = 0; k < n; ++k) {
    int p = k;
    double biggest = fabs(A[k][k]);
#pragma unroll
    for (int i = k + 1; i < n; ++i) { const double v = fabs(A[i][i]); if (v > biggest) { biggest = v; p = i; } }
    perm[k] = p;
#pragma unroll
    for (int i = k + 1; i < n; ++i) {
      if (p == i) {
#pragma unroll
        for (int cc = 0; cc < n; ++cc) swap_d(A[k][cc], A[i][cc]);
#pragma unroll
        for (int r = 0; r < n; ++r) swap_d(A[r][k], A[r][i]);
      }
    }
    double temp[n];
#pragma unroll
    for (int j = 0; j < k; ++j) temp[j] = A[j][j] * A[k][j];
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < k; ++j) acc += A[k][j] * temp[j];
    const double dk = A[k][k] - acc;
    A[k][k] = dk;
#pragma unroll
    for (int i = k + 1; i < n; ++i) {
      double sacc = 0.0;
#pragma unroll
      for (int j = 0; j < k; ++j) sacc += A[i][j] * temp[j];
      const double v = A[i][k] - sacc;
      A[i][k] = (fabs(dk) > 0.0) ? v / dk : 0.0;
    }
  }
  double y[n];
#pragma unroll
  for (int i = 0; i < n; ++i) y[i] = (double)b[i];
#pragma unroll
  for (int k = 0; k < n; ++k) {
#pragma unroll
    for (int i = k + 1; i < n; ++i)
      if (perm[k] == i) swap_d(y[k], y[i]);
  }
#pragma unroll
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int j = 0; j < i; ++j) y[i] -= A[i][j] * y[j];
#pragma unroll
  for (int i = 0; i < n; ++i) { const double d = A[i][i]; y[i] = (fabs(d) > 2.2250738585072014e-308) ? y[i] / d : 0.0; }
#pragma unroll
  for (int i = n - 1; i >= 0; --i)
#pragma unroll
    for (int j = i + 1; j < n; ++j) y[i] -= A[j][i] * y[j];
#pragma unroll
  for (int k = n - 1; k >= 0; --k) {
#pragma unroll
    for (int i = k + 1; i < n; ++i)
      if (perm[k] == i) swap_d(y[k], y[i]);
  }
#pragma unroll
  for (int i = 0; i < n; ++i) x[i] = (float)y[i];
}

// frame_T_global_estimate = global_T_frame_estimate.inverse() for every keyframe's start pose
// (BS/direct_ba_alternating.cc:140).
__global__ void pose_init_kernel(int kf_count, const PoseState* __restrict__ states, KfDev* __restrict__ kfs, int* __restrict__ active_counters) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < 4) active_counters[k] = 0;   // the loop's four rotating "still unconverged" counters
  if (k >= kf_count) return;
  const PoseState st = states[k];
  se3_inverse_matrix(Quat{st.q[0], st.q[1], st.q[2], st.q[3]}, mk3(st.t[0], st.t[1], st.t[2]), kfs[k].frame_T_global.m);
}

// The keyframes that are still unconverged, in ascending order: out = {n, list[n] (entries up to kf_count), pos[kf_count]} with
// pos[k] = place of keyframe k in the list, or -1.  One block; runs between the solve of one Gauss-Newton iteration and the
// accumulation of the next (and once behind pose_init_kernel).
constexpr int kActiveListThreads = 1024;
__global__ __launch_bounds__(kActiveListThreads) void pose_active_list_kernel(int kf_count, const PoseState* __restrict__ states, int* __restrict__ out) {
  __shared__ int wave_total[kActiveListThreads / 64];
  __shared__ int base;
  int* list = out + 1;
  int* pos = out + 1 + kf_count;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int k0 = 0; k0 < kf_count; k0 += kActiveListThreads) {
    const int k = k0 + (int)threadIdx.x;
    const bool active = k < kf_count && !states[k].converged;
    const unsigned long long mask = __ballot(active);
    if (lane == 0) wave_total[wave] = __builtin_popcountll(mask);
    __syncthreads();
    int before = base;
    for (int w = 0; w < wave; ++w) before += wave_total[w];
    if (active) {
      const int place = before + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
      list[place] = k;
      pos[k] = place;
    } else if (k < kf_count) {
      pos[k] = -1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = base;
      for (int w = 0; w < kActiveListThreads / 64; ++w) t += wave_total[w];
      base = t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = base;
}

// One Gauss-Newton update per keyframe (BS/direct_ba_alternating.cc:206-233).
__global__ void pose_solve_kernel(const float* __restrict__ coeffs, int kf_count, PoseState* __restrict__ states,
                                  KfDev* __restrict__ kfs, int* __restrict__ active_count, int* __restrict__ next_active_count) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0 && next_active_count != nullptr) *next_active_count = 0;   // last read four iterations ago
  if (k >= kf_count) return;
  PoseState st = states[k];
  if (st.converged) return;
  float H[21], b[6], x[6];
  for (int i = 0; i < 21; ++i) H[i] = coeffs[(size_t)k * kRow + i];
  for (int i = 0; i < 6; ++i) b[i] = coeffs[(size_t)k * kRow + 21 + i];
  solve_ldlt6(H, b, x);
  float neg[6];
  for (int i = 0; i < 6; ++i) neg[i] = -1.f * x[i];
  Quat dq; f3 dt;
  se3_exp(neg, &dq, &dt);
  Quat q{st.q[0], st.q[1], st.q[2], st.q[3]};
  f3 t = mk3(st.t[0], st.t[1], st.t[2]);
  se3_mul_inplace(&q, &t, dq, dt);
  // IsScale1PoseEstimationConverged BS/convergence_analysis.h:45-52
  const float sc = 1e-06f / 1e-07f;
  float nrm = 0.f;
  for (int i = 0; i < 6; ++i) { const float v = (i < 3) ? x[i] : x[i] * sc; nrm += v * v; }
  st.q[0] = q.x; st.q[1] = q.y; st.q[2] = q.z; st.q[3] = q.w;
  st.t[0] = t.x; st.t[1] = t.y; st.t[2] = t.z;
  st.iterations += 1;
  st.converged = (nrm < 1e-06f) ? 1 : 0;
  states[k] = st;
  se3_inverse_matrix(q, t, kfs[k].frame_T_global.m);
  if (!st.converged) atomicAdd(active_count, 1);
}

// pose_reduce_kernel + pose_reduce_final_kernel + pose_solve_kernel in one launch (single-GPU path: no exchange in
// between).  Block k sums the per-wave rows [rows][32] of keyframe k -- thread (sub, col) owns the rows r = sub mod 32 of
// column col and keeps four independent partial sums for memory-level parallelism; the 32 per-thread sums of a column
// are then added in a fixed order (deterministic) -- and thread 0 does the fp64 pivoted LDL^T, the SE3 update and the
// convergence test (LDL^T unrolled in registers).
constexpr int kReduceSolveThreads = 1024;
__global__ __launch_bounds__(kReduceSolveThreads) void pose_reduce_solve_kernel(const float* __restrict__ partials, int rows_per_kf, int kf_count,
                                                                               PoseState* __restrict__ states, KfDev* __restrict__ kfs,
                                                                               int* __restrict__ active_count, int* __restrict__ next_active_count,
                                                                               const VisWord* __restrict__ vis, int kfs_per_block, unsigned long long* __restrict__ stats,
                                                                               const int* __restrict__ kf_pos) {
  const int k = blockIdx.x;
  if (k == 0 && threadIdx.x == 0) *next_active_count = 0;   // the next iteration's counter (last read four iterations ago)
  if (states[k].converged) return;   // uniform
  __shared__ float sm[32][kRow];
  __shared__ float row[kRow];
  extern __shared__ unsigned long long vmap[];   // visit_map_bytes(slots)
  const int col = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int slots = rows_per_kf / kRowsPerSlot;
  const int place = kf_pos ? kf_pos[k] : k;   // the keyframe's place in the list the accumulation walked (its chunk and bit)
  const uint32_t visiting = build_visit_map(vis + (size_t)(place / kfs_per_block) * slots, (uint32_t)(place % kfs_per_block), slots, vmap);
  if (stats != nullptr && threadIdx.x == 0) { atomicAdd(&stats[0], (unsigned long long)slots); atomicAdd(&stats[1], (unsigned long long)visiting); }
  sm[sub][col] = column_share_of_rows(partials + (size_t)k * rows_per_kf * kRow + col, sub, rows_per_kf, vmap);
  __syncthreads();
  if (threadIdx.x < 27) {
    float total = 0.f;
    for (int i = 0; i < 32; ++i) total += sm[i][threadIdx.x];
    row[threadIdx.x] = total;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  PoseState st = states[k];
  float H[21], b[6], x[6];
  for (int i = 0; i < 21; ++i) H[i] = row[i];
  for (int i = 0; i < 6; ++i) b[i] = row[21 + i];
  solve_ldlt6(H, b, x);
  float neg[6];
  for (int i = 0; i < 6; ++i) neg[i] = -1.f * x[i];
  Quat dq; f3 dt;
  se3_exp(neg, &dq, &dt);
  Quat q{st.q[0], st.q[1], st.q[2], st.q[3]};
  f3 t = mk3(st.t[0], st.t[1], st.t[2]);
  se3_mul_inplace(&q, &t, dq, dt);
  const float sc = 1e-06f / 1e-07f;   // IsScale1PoseEstimationConverged BS/convergence_analysis.h:45-52
  float nrm = 0.f;
  for (int i = 0; i < 6; ++i) { const float v = (i < 3) ? x[i] : x[i] * sc; nrm += v * v; }
  st.q[0] = q.x; st.q[1] = q.y; st.q[2] = q.z; st.q[3] = q.w;
  st.t[0] = t.x; st.t[1] = t.y; st.t[2] = t.z;
  st.iterations += 1;
  st.converged = (nrm < 1e-06f) ? 1 : 0;
  states[k] = st;
  se3_inverse_matrix(q, t, kfs[k].frame_T_global.m);
  if (!st.converged) atomicAdd(active_count, 1);
}

}  // namespace bslam
