/*
 * bso_bench.c -- ORACLE (test infrastructure only): multi-threaded driver used ONLY by
 * bench.py's cpu_baseline leg.  Runs the serial restatement of the batched pose pass
 * (bso_accumulate_pose_estimation_coeffs per keyframe) over keyframes in parallel with
 * OpenMP, one keyframe per task, so the arithmetic per (surfel, keyframe) pair is exactly
 * the oracle's.  Returns the number of OpenMP threads used.
 */
#include <omp.h>

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
#pragma omp for schedule(dynamic, 1)
    for (int k = 0; k < keyframe_count; ++k) {
      uint32_t count = 0;
      float cost = 0.f;
      bso_accumulate_pose_estimation_coeffs(
          use_depth_residuals, use_descriptor_residuals, color_camera, depth_camera, dp,
          &keyframes[k].depth, &keyframes[k].normals, &keyframes[k].color, &keyframes[k].frame_T_global,
          surfels_size, surfels, tex_mode, &count, &cost, Hb + 27 * (size_t)k, Hb + 27 * (size_t)k + 21, 0, 0, 0);
      if (counts) counts[k] = count;
    }
  }
  return used;
}
