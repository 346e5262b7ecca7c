/*
 * bslam_oracle.h -- ORACLE (test infrastructure only).
 *
 * CPU restatement of the reference's bundle-adjustment hot path
 * (/root/reference/applications/badslam/src/badslam, "BS/").  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / reported baseline.  The product (badslam_amd/) never
 * links, imports or calls it.
 *
 * Pinning: the reference ships no golden vectors for this path (BS/test/ holds only
 * GPU integration tests with closed-form answers).  The oracle is pinned by
 * re-stating those known-answer scenarios (tests/test_oracle_known_answers.py):
 * pose recovery < 1.1e-6 (BS/test/test_pose_optimization_geometric_residual.cc:168),
 * < 8e-5 photometric (BS/test/test_pose_optimization_photometric_residual.cc:175),
 * surfel depth < 1e-4 (BS/test/test_geometry_optimization_geometric_residual.cc:190-205).
 * A reference build (oracle/_ref) is NOT possible in this image: the path's sources
 * need <cuda_runtime.h>, <cub/cub.cuh>, Eigen and Sophus/Eigen headers that the
 * image lacks, and stand-ins for missing headers are not allowed.
 *
 * All pointers are HOST pointers; the POD types are those of include/badslam_hip.h.
 */
#ifndef BSLAM_ORACLE_H_
#define BSLAM_ORACLE_H_

#include <stdint.h>

#include "../include/badslam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- SE3 helpers (Sophus::SE3f restated: libvis/third_party/sophus/sophus/{so3,se3}.hpp) */
void bso_se3_identity(bslam_se3f* T);
void bso_se3_exp(const float x[6], bslam_se3f* out);             /* se3.hpp:293-313, so3.hpp:282-318 */
void bso_se3_log(const bslam_se3f* T, float out[6]);             /* se3.hpp:435-466, so3.hpp:421-466 */
void bso_se3_mul(const bslam_se3f* a, const bslam_se3f* b, bslam_se3f* out);  /* se3.hpp operator*=, so3.hpp:215-232 */
void bso_se3_inverse(const bslam_se3f* T, bslam_se3f* out);      /* se3.hpp inverse() */
void bso_se3_matrix3x4(const bslam_se3f* T, bslam_mat3x4* out);  /* se3.hpp matrix3x4(), Eigen Quaternion::toRotationMatrix */
void bso_se3_rotation(const bslam_se3f* T, bslam_mat3x3* out);
/* Fills kf->frame_T_global and kf->global_R_frame from global_T_frame (BS/keyframe.h:160-173). */
void bso_keyframe_set_pose(bslam_keyframe_view* kf, const bslam_se3f* global_T_frame);

/* BS/convergence_analysis.h:45-52 */
int bso_is_scale1_pose_estimation_converged(const float x[6]);

/* H (21, upper triangle row-major) x = b in double via pivoted LDL^T, restating
 * Eigen 3.3 LDLT (un-vendored dependency, "3.3.7 known to work", REF/README.md:79)
 * as used at BS/direct_ba_alternating.cc:206.  n <= 6. */
void bso_solve_ldlt_upper(int n, const float* H_upper, const float* b, float* x);

/* ---- per-keyframe kernels ---- */

/* out_pixel[i] = py*width+px or 0xffffffff (SurfelProjectsToAssociatedPixel,
 * BS/surfel_projection_nvcc_only.cuh:302-332). */
void bso_association(const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
                     const bslam_keyframe_view* kf, uint32_t surfels_size, const bslam_buffer2d* surfels,
                     uint32_t* out_pixel);

/* AccumulatePoseEstimationCoeffsCUDA + kernel (BS/kernel_opt_pose.cc:39-97,
 * BS/kernel_opt_pose.cu:251-383).  H[21], b[6] are fp32 sums in surfel-index order;
 * H64/b64 (may be NULL) the same sums in double (per-term values still fp32).
 * per_surfel (may be NULL): 8 floats per surfel
 *   [depth raw residual, depth weight, desc r1, desc w1, desc r2, desc w2, flags, 0]
 *   flags bit0 = depth-associated, bit1 = descriptor residuals valid. */
void bso_accumulate_pose_estimation_coeffs(
    int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp,
    const bslam_buffer2d* depth_buffer, const bslam_buffer2d* normals_buffer, const bslam_buffer2d* color_buffer,
    const bslam_mat3x4* frame_T_global, uint32_t surfels_size, const bslam_buffer2d* surfels,
    int tex_mode, uint32_t* residual_count, float* residual_sum,
    float* H, float* b, double* H64, double* b64, float* per_surfel);

/* DirectBA::EstimateFramePose (BS/direct_ba_alternating.cc:42-283). */
void bso_estimate_frame_pose(
    int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp,
    const bslam_buffer2d* depth_buffer, const bslam_buffer2d* normals_buffer, const bslam_buffer2d* color_buffer,
    const bslam_se3f* global_T_frame_initial, uint32_t surfels_size, const bslam_buffer2d* surfels,
    int tex_mode, int max_iterations, bslam_se3f* out_global_T_frame, int* iterations_done, int* converged);

/* UpdateSurfelActivationCUDA (BS/kernel_surfel_activation.cc:39-67). */
/* AssignColorsCUDA (BS/kernels.h:301-308).  Writes the scratch rows 8..12 like the reference. */
void bso_assign_colors(
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    int keyframe_count, const bslam_keyframe_view* keyframes, int tex_mode,
    uint32_t surfels_size, const bslam_buffer2d* surfels);

void bso_update_surfel_activation(
    const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels, const bslam_buffer2d* active_surfels);

/* UpdateSurfelNormalsCUDA (BS/kernel_opt_geometry.cc:39-78). */
void bso_update_surfel_normals(
    const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels, const bslam_buffer2d* active_surfels);

/* OptimizeGeometryIterationCUDA (BS/kernel_opt_geometry.cc:80-201). */
void bso_optimize_geometry_iteration(
    int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels, const bslam_buffer2d* active_surfels,
    int tex_mode);

/* ---- PCG (BS/kernel_pcg.cu, host pointers in bslam_pcg_vectors) ---- */
void bso_pcg_init(const bslam_pcg_layout* layout,
                  const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
                  const bslam_depth_params* dp,
                  int keyframe_count, const bslam_keyframe_view* keyframes,
                  uint32_t surfels_size, const bslam_buffer2d* surfels,
                  const bslam_pcg_vectors* v, int tex_mode);
void bso_pcg_init2(const bslam_pcg_layout* layout, float a, const bslam_pcg_vectors* v);
void bso_pcg_step1(const bslam_pcg_layout* layout,
                   const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
                   const bslam_depth_params* dp,
                   int keyframe_count, const bslam_keyframe_view* keyframes,
                   uint32_t surfels_size, const bslam_buffer2d* surfels,
                   const bslam_pcg_vectors* v, int clear_g, int tex_mode);
double bso_pcg_last_alpha_d64(void);  /* float64 sum of the alpha_d terms of the last bso_pcg_step1 */
void bso_pcg_step2(const bslam_pcg_layout* layout, const bslam_pcg_vectors* v, float* beta_n_host);
void bso_pcg_step3(const bslam_pcg_layout* layout, const bslam_pcg_vectors* v);
void bso_update_surfels_from_pcg_delta(uint32_t surfels_size, const bslam_buffer2d* surfels,
                                       int use_descriptor_residuals, uint32_t surfel_unknown_start_index,
                                       const float* pcg_delta);
void bso_update_cfactors_from_pcg_delta(const bslam_buffer2d* cfactor_buffer,
                                        uint32_t cfactor_unknown_start_index, const float* pcg_delta);

/* OptimizeIntrinsicsCUDA (BS/kernel_opt_intrinsics.cc:38-283): one Gauss-Newton step on the depth
 * intrinsics (1/fx, 1/fy, -cx/fx, -cy/fy, a; then the cfactor cells through the Schur complement)
 * and/or the colour intrinsics.  Updates dp->cfactor_buffer in place, *a, and the out cameras. */
void bso_optimize_intrinsics(
    int optimize_depth_intrinsics, int optimize_color_intrinsics,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    bslam_camera4f* out_color_camera, bslam_camera4f* out_depth_camera, float* a, int tex_mode);

/* ---- scene construction (restates the producers either side of the path so the
 * reference's known-answer scenes can be rebuilt; "next" rows of SURVEY.md 8f) ---- */

/* ComputeBrightnessCUDA (BS/cuda_image_processing.cu:165-194): rgb (3 bytes/pixel,
 * tightly packed) -> uchar4 with .w = luma. */
void bso_compute_brightness(int width, int height, const uint8_t* rgb, const bslam_buffer2d* out_color);

/* ComputeNormalsCUDA (BS/cuda_depth_processing.cu:134-255) followed by
 * ComputePointRadiiAndRemoveIsolatedPixelsCUDA (:286-357), as the Keyframe
 * constructor chains them (BS/keyframe.cc:116-138): in_depth -> kf depth / normals /
 * radius (all u16, caller-allocated).  Returns min/max metric depth of valid pixels
 * in *min_depth / *max_depth (BS/cuda_depth_processing.cu:391-465). */
/* keyframe preprocessing producers, BS/cuda_depth_processing.cu */
void bso_bilateral_filter_and_depth_cutoff(float sigma_xy, float sigma_value, float radius_factor, uint16_t max_depth,
                                           float raw_to_float_depth, const bslam_buffer2d* in_depth, const bslam_buffer2d* out_depth);
void bso_compute_normals(const bslam_camera4f* depth_camera, const bslam_depth_params* dp, const bslam_buffer2d* in_depth,
                         const bslam_buffer2d* out_depth, const bslam_buffer2d* out_normals);
void bso_compute_point_radii_and_remove_isolated_pixels(const bslam_camera4f* depth_camera, float raw_to_float_depth,
                                                        const bslam_buffer2d* in_depth, const bslam_buffer2d* out_radius,
                                                        const bslam_buffer2d* out_depth);
void bso_compute_min_max_depth(const bslam_buffer2d* depth, float raw_to_float_depth, float* min_depth, float* max_depth);
void bso_preprocess_depth(const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
                          const bslam_buffer2d* in_depth, const bslam_buffer2d* out_depth,
                          const bslam_buffer2d* out_normals, const bslam_buffer2d* out_radius,
                          float* min_depth, float* max_depth);

float bso_half_to_float(uint16_t h);   /* IEEE binary16 -> float (__half2float) */

/* ---- surfel lifecycle ------------------------------------------------------
 * The reference decides cell ownership with atomicCAS races (BS/kernel_supporting_surfels.cu:58,
 * BS/kernel_create_surfels.cu:62); any interleaving is a valid outcome.  The oracle (and the HIP
 * kernels) use the interleaving "lowest surfel index / raster order first". */

/* DetermineSupportingSurfelsCUDA / DetermineSupportingSurfelsAndMergeSurfelsCUDA
 * (BS/kernel_supporting_surfels.{cu,cc}).  supporting[3] are cell images (cells_h x cells_w u32, row-major,
 * kInvalidIndex = 0xffffffff).  With merge != 0 merged surfels get x = NaN (0x7fffffff) and *surfel_count
 * is decreased. */
void bso_determine_supporting_surfels(
    int merge, float merge_dist_factor, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    const bslam_keyframe_view* kf, uint32_t surfels_size, const bslam_buffer2d* surfels,
    uint32_t* supporting0, uint32_t* supporting1, uint32_t* supporting2, uint32_t* surfel_count);

/* DirectBA::CreateSurfelsForKeyframe (BS/direct_ba.cc:340-405, BS/kernel_create_surfels.{cu,cc}).
 * global_T_frame: the keyframe's pose matrix; covis_*: the co-visible keyframes and
 * covis_frame_T_global * global_T_frame per entry (only used when filter_new_surfels).
 * Appends at *surfels_size and returns the number of surfels created; creates none if they would not fit
 * (the reference logs an error and returns, BS/kernel_create_surfels.cc:162-165). */
uint32_t bso_create_surfels_for_keyframe_ex(
    int filter_new_surfels, int min_observation_count,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp, const bslam_keyframe_view* kf, const bslam_mat3x4* global_T_frame,
    int covis_count, const bslam_keyframe_view* covis_keyframes, const bslam_mat3x4* covis_T_frame,
    uint32_t* surfels_size, uint32_t max_surfels, const bslam_buffer2d* surfels, int tex_mode);

/* filter_new_surfels = false, pose given as SE3 (scene builder of the known-answer tests) */
uint32_t bso_create_surfels_for_keyframe(
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp, const bslam_keyframe_view* kf, const bslam_se3f* global_T_frame,
    uint32_t* surfels_size, uint32_t max_surfels, const bslam_buffer2d* surfels, int tex_mode);

/* DeleteSurfelsAndUpdateRadiiCUDA (BS/kernel_delete_surfels.{cu,cc}) */
void bso_delete_surfels_and_update_radii(
    int min_observation_count, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    int keyframe_count, const bslam_keyframe_view* keyframes, uint32_t* surfel_count, uint32_t surfels_size,
    const bslam_buffer2d* surfels);

/* CompactSurfelsCUDA (BS/kernel_compact_surfels.cu:111-281); active_surfels may be NULL */
void bso_compact_surfels(uint32_t surfel_count, uint32_t* surfels_size, const bslam_buffer2d* surfels,
                         const bslam_buffer2d* active_surfels);

/* ---- pairwise frame tracking / odometry, see bso_odometry.c ------------------ */
void bso_brightness_from_color(const bslam_buffer2d* color_uchar4, const bslam_buffer2d* out_u8);
void bso_calibrate_depth(const bslam_depth_params* dp, const bslam_buffer2d* depth_u16, const bslam_buffer2d* out_depth);
void bso_downsample_images(const bslam_buffer2d* depth, const bslam_buffer2d* normals, const bslam_buffer2d* color_u8, int tex_mode,
                           const bslam_buffer2d* out_depth, const bslam_buffer2d* out_normals, const bslam_buffer2d* out_color);
/* out: [num_scales][6] images {base depth f32, base normals u16, base colour u8, tracked depth, normals, colour};
 * free with bso_free_tracking_pyramids */
void bso_compute_sobel_gradient_magnitude(const bslam_buffer2d* color_uchar4, const bslam_buffer2d* out_u8);
void bso_calibrate_and_downsample_images(int downsample_color, const bslam_depth_params* dp, const bslam_buffer2d* depth_u16, const bslam_buffer2d* normals,
                                         const bslam_buffer2d* color_u8, int tex_mode, const bslam_buffer2d* out_depth, const bslam_buffer2d* out_normals,
                                         const bslam_buffer2d* out_color);
/* tracker variant used by the image-pair functions below and by bso_track_frame_pairwise (default: 0, 1) */
void bso_set_tracking_variant(int use_gradmag, int use_pyramid_level_0);
void bso_build_tracking_pyramids(
    int num_scales, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    const bslam_buffer2d* tracked_depth_u16, const bslam_buffer2d* tracked_normals, const bslam_buffer2d* tracked_color_uchar4,
    const bslam_buffer2d* base_depth_u16, const bslam_buffer2d* base_normals, const bslam_buffer2d* base_color_uchar4, int tex_mode,
    bslam_buffer2d* out);
void bso_free_tracking_pyramids(int num_scales, bslam_buffer2d* pyramids);
void bso_accumulate_pose_coeffs_from_images(
    int use_depth, int use_desc, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, float baseline_fx, float threshold_factor,
    const bslam_buffer2d* frame_depth, const bslam_buffer2d* frame_normals, const bslam_buffer2d* frame_color,
    const bslam_mat3x4* estimate_frame_T_surfel_frame,
    const bslam_buffer2d* surfel_depth, const bslam_buffer2d* surfel_normals, const bslam_buffer2d* surfel_color, int tex_mode,
    double* H, double* b, uint32_t* visible_count);
void bso_compute_cost_and_residual_count_from_images(
    int use_depth, int use_desc, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, float baseline_fx, float threshold_factor,
    const bslam_buffer2d* frame_depth, const bslam_buffer2d* frame_normals, const bslam_buffer2d* frame_color,
    const bslam_mat3x4* estimate_frame_T_surfel_frame,
    const bslam_buffer2d* surfel_depth, const bslam_buffer2d* surfel_normals, const bslam_buffer2d* surfel_color, int tex_mode,
    uint32_t* residual_count, double* cost);
int bso_is_scale_n_pose_estimation_converged(const float x[6], float scaling_factor);
void bso_track_frame_pairwise(
    int num_scales, int use_depth, int use_desc, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
    const bslam_buffer2d* tracked_depth_u16, const bslam_buffer2d* tracked_normals, const bslam_buffer2d* tracked_color_uchar4,
    const bslam_buffer2d* base_depth_u16, const bslam_buffer2d* base_normals, const bslam_buffer2d* base_color_uchar4, int tex_mode,
    int test_different_initial_estimates, const bslam_se3f* init1, const bslam_se3f* init2, bslam_se3f* out_base_T_frame, int* iterations_per_scale);

void bso_update_surfel_activation_counted(
    const bslam_camera4f* depth_camera, const bslam_depth_params* dp, int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels, const bslam_buffer2d* active_surfels, uint64_t* visited);
/* bench.py's cpu_baseline leg (bso_bench.c) */
int bso_bench_ba_iteration(
    int use_depth_residuals, int use_descriptor_residuals, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* dp, int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels, const bslam_buffer2d* active_surfels, int tex_mode, int num_threads,
    uint64_t* pairs3, float* Hb);
/* point-wise Jacobian formulas (layouts in bslam_oracle.c) */
int bso_jacobian_probe(int kind, int count, const float* in, float* out);
/* 1: the global blocks of bso_optimize_intrinsics are float64 sums of the fp32 terms (default 0: serial fp32) */
void bso_set_intrinsics_sum64(int enable);
/* evaluation shape of bso_math.h: 0 = twin of the HIP kernels (default), 1 = literal transcription of the reference */
void bso_set_literal_mode(int mode);
int bso_get_literal_mode(void);
void bso_association_margins(const bslam_camera4f* depth_camera, const bslam_depth_params* dp, const bslam_keyframe_view* kf,
                             uint32_t surfels_size, const bslam_buffer2d* surfels, double* out);

#ifdef __cplusplus
}
#endif
#endif /* BSLAM_ORACLE_H_ */
