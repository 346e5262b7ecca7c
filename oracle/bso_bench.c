/*
 * bso_bench.c -- ORACLE (test infrastructure only): multi-threaded driver used ONLY by
 * bench.py's cpu_baseline leg.  Runs the serial restatement of the batched pose pass
 * (bso_accumulate_pose_estimation_coeffs) with OpenMP over (keyframe, surfel chunk) tasks -- enough
 * tasks to occupy every host thread -- and adds the chunks' coefficient rows per keyframe.  The arithmetic
 * per (surfel, keyframe) pair is exactly the oracle's.  Returns the number of OpenMP threads used.
 */
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "bslam_oracle.h"

int bso_bench_pose_pass(
    int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels, int tex_mode,
    float* Hb /* [K][27] */, uint32_t* counts /* [K] */, int num_threads) {
  if (num_threads > 0) omp_set_num_threads(num_threads);
  int used = 1;
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
  }
  int chunks = (4 * used + keyframe_count - 1) / (keyframe_count > 0 ? keyframe_count : 1);
  if (chunks < 1) chunks = 1;
  if ((uint32_t)chunks > surfels_size / 1024 + 1) chunks = (int)(surfels_size / 1024 + 1);
  float* part = (float*)calloc((size_t)keyframe_count * chunks * 27, sizeof(float));
  uint32_t* pcount = (uint32_t*)calloc((size_t)keyframe_count * chunks, sizeof(uint32_t));
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int k = 0; k < keyframe_count; ++k) {
    for (int ch = 0; ch < chunks; ++ch) {
      const uint32_t lo = (uint32_t)(((uint64_t)surfels_size * ch) / chunks), hi = (uint32_t)(((uint64_t)surfels_size * (ch + 1)) / chunks);
      bslam_buffer2d view = *surfels;                       /* SoA rows are pitch-strided: a column range is a view */
      view.address = (char*)surfels->address + sizeof(float) * (size_t)lo;
      view.width = (int)(hi - lo);
      uint32_t count = 0;
      float cost = 0.f;
      float* row = part + ((size_t)k * chunks + ch) * 27;
      bso_accumulate_pose_estimation_coeffs(
          use_depth_residuals, use_descriptor_residuals, color_camera, depth_camera, dp,
          &keyframes[k].depth, &keyframes[k].normals, &keyframes[k].color, &keyframes[k].frame_T_global,
          hi - lo, &view, tex_mode, &count, &cost, row, row + 21, 0, 0, 0);
      pcount[(size_t)k * chunks + ch] = count;
    }
  }
  for (int k = 0; k < keyframe_count; ++k) {
    float* out = Hb + 27 * (size_t)k;
    memset(out, 0, 27 * sizeof(float));
    uint32_t c = 0;
    for (int ch = 0; ch < chunks; ++ch) {
      for (int i = 0; i < 27; ++i) out[i] += part[((size_t)k * chunks + ch) * 27 + i];
      c += pcount[(size_t)k * chunks + ch];
    }
    if (counts) counts[k] = c;
  }
  free(part);
  free(pcount);
  return used;
}

/* One alternating BA iteration's surfel x keyframe passes (BS/direct_ba_alternating.cc:430-600) on the given surfels: surfel
 * activation, the geometry iteration (normals, then position [+ descriptors]) and ONE pose-coefficient pass over every
 * keyframe -- what a step of bench.py does on the GPU, with one Gauss-Newton pass instead of the converged loop.  Surfel
 * columns are independent in all three, so the work is split into contiguous column ranges, one OpenMP task each; the
 * arithmetic per pair is exactly the oracle's.  The surfel buffer is MODIFIED (normals / positions / descriptors and the
 * scratch rows 8-16).  pairs3: [0] pairs the activation pass visited, [1] geometry pairs = 2 passes x keyframes x active
 * surfels, [2] pose pairs = keyframes x surfels.  Hb: [K][27] coefficient sums (so that the work cannot be optimised away).
 * Returns the number of OpenMP threads used. */
int bso_bench_ba_iteration(
    int use_depth_residuals, int use_descriptor_residuals, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp, int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels, const bslam_buffer2d* active_surfels, int tex_mode, int num_threads,
    uint64_t* pairs3, float* Hb) {
  if (num_threads > 0) omp_set_num_threads(num_threads);
  int used = 1;
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
  }
  int chunks = 8 * used;
  if ((uint32_t)chunks > surfels_size / 256 + 1) chunks = (int)(surfels_size / 256 + 1);
  float* part = (float*)calloc((size_t)chunks * keyframe_count * 27, sizeof(float));
  uint64_t* visited = (uint64_t*)calloc((size_t)chunks, sizeof(uint64_t));
  uint64_t* active = (uint64_t*)calloc((size_t)chunks, sizeof(uint64_t));
#pragma omp parallel for schedule(dynamic, 1)
  for (int ch = 0; ch < chunks; ++ch) {
    const uint32_t lo = (uint32_t)(((uint64_t)surfels_size * ch) / chunks), hi = (uint32_t)(((uint64_t)surfels_size * (ch + 1)) / chunks);
    bslam_buffer2d view = *surfels, aview = *active_surfels;
    view.address = (char*)surfels->address + sizeof(float) * (size_t)lo;
    view.width = (int)(hi - lo);
    aview.address = (char*)active_surfels->address + (size_t)lo;
    aview.width = (int)(hi - lo);
    bso_update_surfel_activation_counted(depth_camera, dp, keyframe_count, keyframes, hi - lo, &view, &aview, &visited[ch]);
    for (uint32_t i = 0; i < hi - lo; ++i) active[ch] += (((const uint8_t*)aview.address)[i] & BSLAM_SURFEL_ACTIVE_FLAG) ? 1 : 0;
    bso_optimize_geometry_iteration(use_depth_residuals, use_descriptor_residuals, color_camera, depth_camera, dp, keyframe_count, keyframes,
                                    hi - lo, &view, &aview, tex_mode);
    for (int k = 0; k < keyframe_count; ++k) {
      uint32_t count = 0;
      float cost = 0.f;
      float* row = part + ((size_t)ch * keyframe_count + k) * 27;
      bso_accumulate_pose_estimation_coeffs(use_depth_residuals, use_descriptor_residuals, color_camera, depth_camera, dp, &keyframes[k].depth,
                                            &keyframes[k].normals, &keyframes[k].color, &keyframes[k].frame_T_global, hi - lo, &view, tex_mode,
                                            &count, &cost, row, row + 21, 0, 0, 0);
    }
  }
  pairs3[0] = pairs3[1] = 0;
  for (int ch = 0; ch < chunks; ++ch) { pairs3[0] += visited[ch]; pairs3[1] += 2ull * (uint64_t)keyframe_count * active[ch]; }
  pairs3[2] = (uint64_t)keyframe_count * surfels_size;
  if (Hb) {
    memset(Hb, 0, (size_t)keyframe_count * 27 * sizeof(float));
    for (int ch = 0; ch < chunks; ++ch)
      for (int i = 0; i < keyframe_count * 27; ++i) Hb[i] += part[(size_t)ch * keyframe_count * 27 + i];
  }
  free(part); free(visited); free(active);
  return used;
}
