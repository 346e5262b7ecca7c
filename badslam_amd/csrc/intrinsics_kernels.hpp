// intrinsics_kernels.hpp -- alternating-scheme intrinsics / depth-deformation step (gfx950).
//
// Replaces the three kernels of BS/kernel_opt_intrinsics.cu.  The reference accumulates one
// keyframe per launch with 20 + 28 block reductions and 8 contended float atomics per thread.
// Here one launch walks all keyframes; the 5x5 (+5) depth system and the 4x4 (+4) colour system
// are accumulated per thread over the whole loop and reduced once per block (fixed-order sums);
// only the per-cell Schur blocks B, D, b2 and the observation counts use atomics, as they are
// scattered by pixel cell (BS/kernel_opt_intrinsics.cu:176-196).
#pragma once

#include "device_math.hpp"
#include "pose_kernels.hpp"

namespace bslam {

constexpr int kIntrThreads = 256;
constexpr int kIntrR = 2;
constexpr int kIntrRow = 40;   // A[15], b1[5], colour H[10], colour b[4], pad

struct IntrinsicsCells {
  float* B;          // [5][cells]
  float* D;          // [cells]
  float* b2;         // [cells]
  uint32_t* obs;     // [cells]
  double* acc;       // [7][cells]: B, D, b2 summed with fp64 atomics, rounded into the float arrays by intrinsics_cells_round_kernel
  int cells;
};

// The reference adds floats atomically in arrival order (BS/kernel_opt_intrinsics.cu:176-196); the fp64 sums make the
// rounded blocks, and with them whole BA runs, the same on every run.
__global__ __launch_bounds__(256) void intrinsics_cells_round_kernel(IntrinsicsCells cells) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= cells.cells) return;
  for (int q = 0; q < 5; ++q) cells.B[(size_t)q * cells.cells + cell] = (float)cells.acc[(size_t)q * cells.cells + cell];
  cells.D[cell] = (float)cells.acc[(size_t)5 * cells.cells + cell];
  cells.b2[cell] = (float)cells.acc[(size_t)6 * cells.cells + cell];
}

template <bool kDepthIntr, bool kColorIntr>
__global__ __launch_bounds__(kIntrThreads) void intrinsics_accumulate_kernel(
    CamConsts c, const KfDev* __restrict__ kfs, int kf_count, Schedule sc, SurfelRows s, IntrinsicsCells cells,
    float* __restrict__ partial) {
  uint32_t slot;
  if (!slot_of_block(sc, blockIdx.x, &slot)) return;
  f3 gp[kIntrR], gn[kIntrR];
  bool valid[kIntrR];
  float r2[kIntrR], d1[kIntrR], d2[kIntrR];
#pragma unroll
  for (int r = 0; r < kIntrR; ++r) {
    const uint32_t i = surfel_of_slot(sc, slot, r, kIntrR);
    valid[r] = i < s.size;
    const uint32_t j = valid[r] ? i : 0;
    gp[r] = mk3(s.x[j], s.y[j], s.z[j]);
    gn[r] = unpack_normal(s.normal[j]);
    if (kColorIntr) { r2[r] = s.radius_squared[j]; d1[r] = s.d1[j]; d2[r] = s.d2[j]; }
  }
  float acc[kIntrRow];
#pragma unroll
  for (int i = 0; i < kIntrRow; ++i) acc[i] = 0.f;

  BSLAM_FOR_VISITED_KEYFRAMES_IF(k, 0, kf_count, kIntrR, true) {
    const KfDev kf = kfs[k];
#pragma unroll
    for (int r = 0; r < kIntrR; ++r) {
      Proj p;
      DescSamples ds;
      f2 color_pxy, t1, t2;   // the three sample positions of the descriptor residual
      bool has_desc = false;
      if constexpr (kColorIntr) {
        // the quad gathers of the descriptor samples are issued with the record gather, before the association test
        // (see pose_accumulate_kernel)
        if (!valid[r] || !project_to_pixel(c, kf, gp[r], &p)) continue;
        const PixelRecord rec = load_record(c, kf, p);
        has_desc = depth_to_color_pxy_in_bounds(c, p.pxy, &color_pxy);
        tangent_projections(gp[r], gn[r], r2[r], kf.frame_T_global, c, &t1, &t2);
        ds = descriptor_samples_issue(kf, c, color_pxy, t1, t2);
        asm volatile("" ::: "memory");
        if (!associate_with_record(c, kf, gn[r], rec, &p)) continue;
      } else {
        if (!valid[r] || !project_and_associate(c, kf, gp[r], gn[r], &p)) continue;
      }
      const float nx = nx_of(c, (float)p.px), ny = ny_of(c, (float)p.py);
      if (kDepthIntr) {                                           // BS/kernel_opt_intrinsics.cu:82-118, 170-196
        const int sparse_px = p.px / c.cell, sparse_py = p.py / c.cell;
        const float cfactor = *(const float*)((const uint8_t*)c.cfactor + (size_t)sparse_py * c.cfactor_pitch + 4 * (size_t)sparse_px);
        const float raw_inv_depth = 1.0f / (c.raw_to_float_depth * (float)raw_depth_of(kf, p));
        const f3 ln = p.n_local;
        const float inv_stddev = depth_inv_stddev(nx, ny, p.depth, ln, c.baseline_fx);
        float dj[6];
        const float corrected_inv_depth = depth_intrinsics_jacobian(inv_stddev, p.depth, p.px, p.py, nx, ny, gn[r], kf.frame_T_global.m, ln, cfactor, c.a,
                                                                    raw_inv_depth, dj);
        if (fabsf(corrected_inv_depth) > 1e-4f) {
          const f3 lu = mk3(p.depth * nx, p.depth * ny, p.depth);
          const float raw = depth_residual(inv_stddev, ln, lu, p.local);
          const float w = depth_weight(raw);
          int idx = 0;
#pragma unroll
          for (int row = 0; row < 5; ++row) {
#pragma unroll
            for (int col = row; col < 5; ++col) { acc[idx] += w * dj[row] * dj[col]; ++idx; }
          }
          const float wr = w * raw;
#pragma unroll
          for (int i = 0; i < 5; ++i) acc[15 + i] += wr * dj[i];
          const int cell = sparse_px + sparse_py * c.cfactor_width;
#pragma unroll
          for (int q = 0; q < 5; ++q) atomicAdd(&cells.acc[(size_t)q * cells.cells + cell], (double)(w * dj[q] * dj[5]));
          atomicAdd(&cells.acc[(size_t)5 * cells.cells + cell], (double)(w * dj[5] * dj[5]));
          atomicAdd(&cells.acc[(size_t)6 * cells.cells + cell], (double)(w * raw * dj[5]));
          atomicAdd(&cells.obs[cell], 1u);
        }
      }
      if (kColorIntr) {                                           // :120-158, 198-216
        if (has_desc) {
          float r1, rr2, gx1, gy1, gx2, gy2;
          descriptor_samples_finish(kf, c, ds, d1[r], d2[r], [&](f2 (&pts)[3]) { pts[0] = color_pxy; pts[1] = t1; pts[2] = t2; }, &r1, &rr2, &gx1, &gy1, &gx2, &gy2);
          float j1[4], j2[4];
          color_intrinsics_jacobian(gx1, gy1, nx, ny, j1);
          color_intrinsics_jacobian(gx2, gy2, nx, ny, j2);
          if (r1 != 0) {   // the reference uses "residual != 0" as the validity flag (:200, :208)
            const float w = desc_weight(r1);
            int idx = 20;
#pragma unroll
            for (int row = 0; row < 4; ++row) {
#pragma unroll
              for (int col = row; col < 4; ++col) { acc[idx] += w * j1[row] * j1[col]; ++idx; }
            }
            const float wr = w * r1;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[30 + i] += wr * j1[i];
          }
          if (rr2 != 0) {
            const float w = desc_weight(rr2);
            int idx = 20;
#pragma unroll
            for (int row = 0; row < 4; ++row) {
#pragma unroll
              for (int col = row; col < 4; ++col) { acc[idx] += w * j2[row] * j2[col]; ++idx; }
            }
            const float wr = w * rr2;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[30 + i] += wr * j2[i];
          }
        }
      }
    }
  }
  // one row of 40 sums per block, fixed order
  __shared__ float red[4][kIntrRow];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < kIntrRow; ++i) acc[i] = wave_sum(acc[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < kIntrRow; ++i) red[wave][i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < kIntrRow) partial[(size_t)slot * kIntrRow + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// Sums `rows` rows of kIntrRow floats in index order: out[col] (+)= sum.  One block.
__global__ __launch_bounds__(64) void intrinsics_sum_rows_kernel(const float* __restrict__ rows_in, int rows, float* __restrict__ out, float scale, int accumulate) {
  const int col = threadIdx.x;
  if (col >= kIntrRow) return;
  float v = 0.f;
  for (int t = 0; t < rows; ++t) v += rows_in[(size_t)t * kIntrRow + col];
  out[col] = accumulate ? out[col] + scale * v : scale * v;
}

// ComputeIntrinsicsIntermediateMatricesCUDAKernel (BS/kernel_opt_intrinsics.cu:265-340): Schur
// complement terms per cell; the 15 + 5 sums over cells go to one row per block.
__global__ __launch_bounds__(256) void intrinsics_intermediate_kernel(IntrinsicsCells cells, float* __restrict__ partial) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  float acc[kIntrRow];
#pragma unroll
  for (int i = 0; i < kIntrRow; ++i) acc[i] = 0.f;
  if (p < cells.cells) {
    const float D_inverse = 1.0f / cells.D[p];
    if (!(D_inverse < 1e12f)) {
      cells.D[p] = __uint_as_float(0x7fffffffu);
    } else {
      const float D_inv_b2 = D_inverse * cells.b2[p];
      cells.D[p] = D_inv_b2;
      float Bv[5];
#pragma unroll
      for (int row = 0; row < 5; ++row) Bv[row] = cells.B[(size_t)row * cells.cells + p];
      int idx = 0;
#pragma unroll
      for (int row = 0; row < 5; ++row) {
#pragma unroll
        for (int col = row; col < 5; ++col) { acc[idx] = Bv[row] * D_inverse * Bv[col]; ++idx; }
      }
#pragma unroll
      for (int row = 0; row < 5; ++row) acc[15 + row] = Bv[row] * D_inv_b2;
#pragma unroll
      for (int row = 0; row < 5; ++row) cells.B[(size_t)row * cells.cells + p] = D_inverse * Bv[row];
    }
  }
  __shared__ float red[4][kIntrRow];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 20; ++i) acc[i] = wave_sum(acc[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < kIntrRow; ++i) red[wave][i] = (i < 20) ? acc[i] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < kIntrRow) partial[(size_t)blockIdx.x * kIntrRow + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// SolveForPixelIntrinsicsUpdateCUDAKernel (BS/kernel_opt_intrinsics.cu:374-420)
__global__ __launch_bounds__(256) void intrinsics_pixel_update_kernel(IntrinsicsCells cells, const float* __restrict__ x1, uint8_t* cfactor, uint32_t cfactor_pitch, int cfactor_width) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= cells.cells) return;
  float offset = cells.D[p];
  if (offset != offset) {
    offset = 0;
  } else {
#pragma unroll
    for (int row = 0; row < 5; ++row) offset -= cells.B[(size_t)row * cells.cells + p] * x1[row];
  }
  const int y = p / cfactor_width, x = p - y * cfactor_width;
  float* cf = (float*)(cfactor + (size_t)y * cfactor_pitch) + x;
  float v = *cf - offset;
  if (cells.obs[p] == 0) v = 0;
  *cf = v;
}

// Observation counts <-> exactly representable floats, in place (multi-rank exchange of the cell sums).
__global__ __launch_bounds__(256) void intrinsics_obs_convert_kernel(uint32_t* obs, int cells, int to_integer) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= cells) return;
  if (to_integer) obs[p] = (uint32_t)__uint_as_float(obs[p]);
  else obs[p] = __float_as_uint((float)obs[p]);
}

}  // namespace bslam
