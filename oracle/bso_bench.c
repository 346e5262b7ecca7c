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
