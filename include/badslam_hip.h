/*
 * badslam_hip.h -- C ABI of the MI355X-native BAD SLAM bundle-adjustment hot path.
 *
 * Every entry point here replaces one free function of the reference's operator
 * boundary, applications/badslam/src/badslam/kernels.h (cited per function as
 * "BS/kernels.h:<line>"), or one host loop of DirectBA
 * (BS/direct_ba_alternating.cc, BS/direct_ba_pcg.cc).  Signatures use only plain
 * pointers, sizes and the POD mirrors below, so the reference's C++ host code (or
 * any FFI) can bind them without seeing HIP or torch types.
 *
 * Conventions (mirroring the reference):
 *   - first argument after the context is the HIP stream (`void*` == hipStream_t),
 *     BS/kernels.h passes cudaStream_t first everywhere;
 *   - all image / surfel memory is owned by the caller and lives in device memory;
 *   - small results written through host pointers are valid on return (the function
 *     synchronises the stream, as BS/kernel_opt_pose.cc:96 does);
 *   - return value: 0 on success, negative bslam_status on error.  The reference
 *     aborts via CHECK()/LOG(FATAL) (BS/kernel_opt_pose.cc:58-61); we return the
 *     code and keep the message in bslam_last_error().
 */
#ifndef BADSLAM_HIP_H_
#define BADSLAM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* Constants (BS/kernels.cuh:38-93)                                           */
/* ------------------------------------------------------------------------- */

enum {
  BSLAM_INVALID_DEPTH_BIT = 1 << 15,   /* kInvalidDepthBit   BS/kernels.cuh:38 */
  BSLAM_UNKNOWN_DEPTH = 65535,         /* kUnknownDepth      BS/kernels.cuh:41 */
  BSLAM_SURFEL_ACTIVE_FLAG = 1,        /* kSurfelActiveFlag  BS/kernels.cuh:44 */

  /* surfel SoA rows (BS/kernels.cuh:69-93) */
  BSLAM_SURFEL_X = 0,
  BSLAM_SURFEL_Y = 1,
  BSLAM_SURFEL_Z = 2,
  BSLAM_SURFEL_NORMAL = 3,          /* u32: 3 x signed 10 bit */
  BSLAM_SURFEL_RADIUS_SQUARED = 4,
  BSLAM_SURFEL_COLOR = 5,           /* uchar4 rgb0 */
  BSLAM_SURFEL_DESCRIPTOR1 = 6,
  BSLAM_SURFEL_DESCRIPTOR2 = 7,
  BSLAM_SURFEL_ACCUM0 = 8,          /* rows 8..16: scratch, no cross-call guarantees */
  BSLAM_SURFEL_DATA_ATTRIBUTE_COUNT = 8,
  BSLAM_SURFEL_ATTRIBUTE_COUNT = 17
};

/* Keyframe::Activation (BS/keyframe.h:54-67) */
enum {
  BSLAM_KF_ACTIVE = 0,
  BSLAM_KF_COVISIBLE_ACTIVE = 1,
  BSLAM_KF_INACTIVE = 2
};

typedef enum bslam_status {
  BSLAM_OK = 0,
  BSLAM_ERR_INVALID_ARGUMENT = -1,  /* a reference CHECK() would have fired */
  BSLAM_ERR_HIP = -2,               /* a HIP runtime call failed */
  BSLAM_ERR_NO_DEVICE = -3,         /* no gfx950 device visible */
  BSLAM_ERR_OUT_OF_MEMORY = -4
} bslam_status;

/* ------------------------------------------------------------------------- */
/* POD mirrors of the reference's kernel argument types                       */
/* ------------------------------------------------------------------------- */

/* = vis::CUDABuffer_<T> (libvis/src/libvis/cuda/cuda_buffer.cuh:115-118):
 * element (y, x) lives at (T*)((char*)address + y * pitch) + x. */
typedef struct bslam_buffer2d {
  void* address;
  int32_t height;
  int32_t width;
  size_t pitch; /* bytes */
} bslam_buffer2d;

/* = vis::CUDAMatrix3x4 (BS/cuda_matrix.cuh:140-142): three float4 rows. */
typedef struct bslam_mat3x4 {
  float m[12]; /* row-major: m[4*r + c] */
} bslam_mat3x4;

/* = vis::CUDAMatrix3x3 (BS/cuda_matrix.cuh:76-78): three float3 rows. */
typedef struct bslam_mat3x3 {
  float m[9]; /* row-major */
} bslam_mat3x3;

/* = vis::PinholeCamera4f parameters (libvis/src/libvis/camera.h:1740-1743):
 * fx, fy, cx, cy in the pixel-CORNER convention, plus the image size. */
typedef struct bslam_camera4f {
  float fx, fy, cx, cy;
  int32_t width, height;
} bslam_camera4f;

/* = vis::DepthParameters (BS/surfel_projection.cuh:129-149). */
typedef struct bslam_depth_params {
  bslam_buffer2d cfactor_buffer; /* float image, ceil(h/cell) x ceil(w/cell) */
  float a;
  float raw_to_float_depth;
  float baseline_fx;
  int32_t sparse_surfel_cell_size;
} bslam_depth_params;

/* The per-keyframe buffers DirectBA hands to the kernels
 * (BS/keyframe.h:160-173,227-231).  `color` is the uchar4 buffer whose .w holds
 * the luma (BS/cuda_image_processing.cu:173-174); MI355X has no texture unit, so
 * the buffer replaces the reference's cudaTextureObject_t and the bilinear /
 * clamp semantics of BS/keyframe.cc:67-73 are computed in the kernel. */
typedef struct bslam_keyframe_view {
  bslam_buffer2d depth;    /* u16 */
  bslam_buffer2d normals;  /* u16 */
  bslam_buffer2d radius;   /* u16 (IEEE half bits) */
  bslam_buffer2d color;    /* uchar4 */
  bslam_mat3x4 frame_T_global;
  bslam_mat3x3 global_R_frame;
  int32_t activation;      /* BSLAM_KF_* */
  int32_t id;
} bslam_keyframe_view;

/* Pose of one keyframe as Sophus::SE3f stores it (unit quaternion xyzw +
 * translation; libvis/third_party/sophus/sophus/se3.hpp). */
typedef struct bslam_se3f {
  float q[4]; /* x, y, z, w */
  float t[3];
} bslam_se3f;

/* How the colour image is sampled where the reference uses tex2D() with
 * cudaFilterModeLinear (BS/cost_function.cuh:140-156). */
enum {
  BSLAM_TEX_FIXED_POINT_1_8 = 0, /* NVIDIA texture unit semantics: 8 fractional weight bits */
  BSLAM_TEX_EXACT_FLOAT = 1      /* exact fp32 bilinear weights */
};

/* Opaque context: owns the scratch the reference keeps in
 * PoseEstimationHelperBuffers / IntrinsicsOptimizationHelperBuffers
 * (BS/kernels.h:46-89) plus the per-launch partial-sum slabs. */
typedef struct bslam_context bslam_context;

/* ------------------------------------------------------------------------- */
/* Library / context                                                          */
/* ------------------------------------------------------------------------- */

/* Version of this ABI; bump on any signature change. */
int bslam_abi_version(void);

/* Last error message of the calling thread ("" if none). */
const char* bslam_last_error(void);

/* Creates a context on HIP device `device`.  Fails with BSLAM_ERR_NO_DEVICE when
 * no GPU is visible: there is no CPU fallback. */
int bslam_create(int device, bslam_context** out_ctx);
int bslam_destroy(bslam_context* ctx);

/* Texture filtering mode (BSLAM_TEX_*), default BSLAM_TEX_FIXED_POINT_1_8. */
int bslam_set_texture_mode(bslam_context* ctx, int mode);

/* The library derives one record per keyframe pixel (calibrated depth, pixel normal, raw depth)
 * from the caller's depth / normal / cfactor images.  By default the records are rebuilt on every
 * call, because the caller owns those images and may rewrite them in place between calls (the
 * reference's tests do, BS/test/test_geometry_optimization_geometric_residual.cc:124-139).
 * enable = 1: the caller promises to call bslam_invalidate_keyframe_cache() after changing the
 * CONTENT of any keyframe depth / normal image or of the cfactor image in place; records are then
 * re-used while buffer addresses, pitches, a, raw_to_float_depth and the cell size are unchanged. */
int bslam_set_keyframe_cache(bslam_context* ctx, int enable);
int bslam_invalidate_keyframe_cache(bslam_context* ctx);

/* Work order of the surfel kernels (default on): surfels are visited along a Morton curve, the work
 * slots of that order dealt round-robin over the XCDs (every XCD gets an even share of every part of
 * the scene; up to round 3 each XCD took one contiguous eighth, and a launch lasted as long as the
 * busiest eighth).  Calls with at least 4 keyframes order the individual surfels by the Morton code of
 * their positions (device radix sort, cached per surfel buffer) and read a sorted copy of the surfel
 * rows, so that a wave's 64 surfels project onto a few cache lines in every keyframe; shorter
 * keyframe lists (the per-keyframe entry points) order 256-column granules by their centroids.
 * Results do not depend on it except for the summation order of the per-keyframe sums; 0 restores
 * index order (A/B measurements, bit-for-bit comparisons between entry points). */
int bslam_set_xcd_schedule(bslam_context* ctx, int enable);

/* Block-level frustum culling (default on).  With the per-surfel work order the surfels of a workgroup are a compact blob; a
 * workgroup skips every keyframe into whose image no point of its surfels' bounding box can project (decided once per
 * workgroup and keyframe, exactly conservatively -- no pair that passes the reference's projection test
 * (ProjectSurfelToImage, BS/util.cuh:86-99) is ever skipped -- so every output is bit-identical with and without it).  The
 * reference spends a thread on every (surfel, keyframe) pair (BS/kernel_opt_pose.cu:263-275).  0 = visit every pair. */
int bslam_set_culling(bslam_context* ctx, int enable);
/* Batched pose loop (bslam_estimate_frame_poses_batched) on keyframe lists of at least `min_keyframes` keyframes: every
 * Gauss-Newton iteration walks a device-side list of the keyframes that are still unconverged instead of the whole keyframe
 * table, so the number of workgroups of an iteration follows the work that is left (on a sequence most keyframes converge in
 * two or three iterations and a few run to the cap of 30, BS/direct_ba_alternating.cc:130).  Results are bit-identical either
 * way.  Default 64; 0 = never (every iteration launches workgroups for every keyframe, converged or not). */
int bslam_set_pose_keyframe_list(bslam_context* ctx, int min_keyframes);
/* (work slot, keyframe) pairs the pose kernel's launches tested / skipped since the last call (HOST out; counted while
 * bslam_profile_enable is on; synchronises the device and resets the counters). */
int bslam_debug_cull_stats(bslam_context* ctx, uint64_t* tested, uint64_t* culled);

/* Kernel timing for the roofline line of bench.py (the role of the reference's cudaEvent
 * pairs, BS/direct_ba.h:513-532): while enabled, every launch of the dominant kernel of a
 * call (the surfel x keyframe pass) is bracketed by HIP events on the launch stream.
 * bslam_profile_read synchronises those events and returns launches and summed ms since
 * the last enable/read. */
int bslam_profile_enable(bslam_context* ctx, int enable);
enum { BSLAM_PROF_POSE_ACCUMULATE = 0, BSLAM_PROF_GEOMETRY = 1, BSLAM_PROF_PCG_INIT = 2, BSLAM_PROF_PCG_STEP1 = 3, BSLAM_PROF_ACTIVATION = 4,
       BSLAM_PROF_EXCHANGE = 5,      /* the K x 32 all-reduce of a batched Gauss-Newton iteration (surfel-sharded runs) */
       BSLAM_PROF_POSE_REDUCE = 6 }; /* row sums + 6x6 solve of a batched Gauss-Newton iteration (single-GPU path) */
int bslam_profile_read(bslam_context* ctx, int kernel, int32_t* launches, float* total_ms);
/* Work counters accumulated since bslam_profile_enable(ctx, 1) by the counting variants of the kernels (8 values, HOST out):
 * [0] (surfel, keyframe) pairs the activation pass actually visited (it stops at a surfel's first associated active keyframe,
 * BS/kernel_surfel_activation.cu:64-79), [1] surfels it set active, [2..7] reserved.  Synchronises the device. */
int bslam_profile_read_counters(bslam_context* ctx, uint64_t* counters8);

/* ------------------------------------------------------------------------- */
/* Pose optimisation                                                          */
/* ------------------------------------------------------------------------- */

/* Replaces AccumulatePoseEstimationCoeffsCUDA (BS/kernels.h:156-174,
 * BS/kernel_opt_pose.cc:39-97): Gauss-Newton coefficients H (21, row-major upper
 * triangle) and b (6) of one keyframe's pose over all surfels.  `residual_count`
 * / `residual_sum` are filled when `debug` != 0 (quirk Q1 of the reference is
 * kept: with descriptor residuals only the first one is counted,
 * BS/kernel_opt_pose.cu:373-381).  H, b, residual_* are HOST pointers, valid on
 * return.  Sums are combined in a fixed order (deterministic), unlike the
 * reference's atomicAdd (BS/gauss_newton.cuh:71,89). */
int bslam_accumulate_pose_estimation_coeffs(
    bslam_context* ctx, void* stream,
    int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* depth_params,
    const bslam_buffer2d* depth_buffer, const bslam_buffer2d* normals_buffer,
    const bslam_buffer2d* color_buffer,
    const bslam_mat3x4* frame_T_global_estimate,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    int debug, uint32_t* residual_count, float* residual_sum,
    float* H, float* b);

/* Replaces the keyframe loop around DirectBA::EstimateFramePose in
 * BS/direct_ba_alternating.cc:543-577 (and EstimateFramePose itself, :42-283):
 * all keyframes with activation != BSLAM_KF_INACTIVE advance in lock-step, one
 * launch per Gauss-Newton iteration over (surfels x keyframes); the 6x6 solve
 * (double LDLT, :206), the update global_T_frame * exp(-x) (:214) and the
 * convergence test (BS/convergence_analysis.h:45-52) run on the device, so there
 * is one host sync per iteration instead of one per keyframe and iteration.
 * Surfels are frozen during this phase in the reference too, so per-keyframe
 * results are those of the sequential loop up to summation order.
 *   poses[K]            in/out, global_T_frame per keyframe
 *   iterations_done[K]  out (may be NULL)
 *   converged[K]        out (may be NULL)
 * `allreduce` (may be NULL) is called once per iteration on the device buffer
 * holding K x 32 floats of partial coefficients when surfels are sharded over
 * several GPUs. */
typedef int (*bslam_allreduce_fn)(void* user, void* device_buffer, size_t float_count, void* stream);

int bslam_estimate_frame_poses_batched(
    bslam_context* ctx, void* stream,
    int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    int max_iterations,
    bslam_se3f* poses, int32_t* iterations_done, int32_t* converged,
    bslam_allreduce_fn allreduce, void* allreduce_user);

/* One batched accumulation without the solve: H/b for all K keyframes at the
 * given frame_T_global (kf.frame_T_global), written to HOST Hb[K][27] (21 H then
 * 6 b) and counts[K] (number of depth-associated surfels).  Parity probe for the
 * batched kernel and the unit the benchmark times. */
int bslam_accumulate_pose_coeffs_batched(
    bslam_context* ctx, void* stream,
    int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    float* Hb, uint32_t* counts);

/* ------------------------------------------------------------------------- */
/* Surfel activation / geometry                                               */
/* ------------------------------------------------------------------------- */

/* Replaces UpdateSurfelActivationCUDA (BS/kernels.h:262-269,
 * BS/kernel_surfel_activation.cc:39-67): clears bit 0 of active_surfels[0..S) and
 * sets it for every surfel associated with a pixel of an ACTIVE keyframe. */
int bslam_update_surfel_activation(
    bslam_context* ctx, void* stream,
    const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    const bslam_buffer2d* active_surfels);

/* Replaces UpdateSurfelNormalsCUDA (BS/kernels.h:225-232,
 * BS/kernel_opt_geometry.cc:39-78). */
int bslam_update_surfel_normals(
    bslam_context* ctx, void* stream,
    const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    const bslam_buffer2d* active_surfels);

/* Replaces OptimizeGeometryIterationCUDA (BS/kernels.h:234-244,
 * BS/kernel_opt_geometry.cc:80-201): normal update, then the position step
 * (geometry-only) or the joint position+descriptor step.  Accumulators stay in
 * registers across keyframes (keyframe order = list order, as the reference's
 * serialised launches), scratch rows 8-16 are not touched. */
int bslam_optimize_geometry_iteration(
    bslam_context* ctx, void* stream,
    int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    const bslam_buffer2d* active_surfels);

/* Replaces OptimizeIntrinsicsCUDA (BS/kernels.h:246-260, BS/kernel_opt_intrinsics.cc:38-283): one
 * Gauss-Newton step on the depth intrinsics (1/fx, 1/fy, -cx/fx, -cy/fy, a, then the per-cell
 * cfactors through the Schur complement, prior 100 a^2) and / or the colour intrinsics.
 * depth_params->cfactor_buffer (device) is updated in place; *a is in/out (the value in
 * depth_params->a is ignored in favour of *a); the out cameras are HOST structs. */
int bslam_optimize_intrinsics(
    bslam_context* ctx, void* stream,
    int optimize_depth_intrinsics, int optimize_color_intrinsics,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* depth_params,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    bslam_camera4f* out_color_camera, bslam_camera4f* out_depth_camera, float* a);

/* Per-surfel association probe (test / debugging aid; the reference has no such
 * export, the tests need it to check the bit-exact parity target of
 * SURVEY.md 8(a5)): for one keyframe writes out_pixel[i] = py * width + px of the
 * associated pixel, or 0xffffffff when surfel i is not associated.  DEVICE ptr. */
int bslam_debug_association(
    bslam_context* ctx, void* stream,
    const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    const bslam_keyframe_view* keyframe,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    uint32_t* out_pixel);

/* Tuning knob of bslam_optimize_geometry_iteration: long keyframe lists are walked in (equally long) launches of at most
 * `keyframes_per_launch` keyframes, with the per-surfel sums carried in library scratch (results are bit-identical to a
 * single launch).  -1 = the library's default (one launch); 0 = always one launch. */
int bslam_set_geometry_keyframe_chunk(bslam_context* ctx, int keyframes_per_launch);

/* Replaces AssignColorsCUDA (BS/kernels.h:301-308, BS/kernel_assign_colors.cc:40-80, .cu:42-125): every surfel's
 * colour row becomes the mean of the bilinearly filtered uchar4 colours of the pixels it is associated with over ALL
 * listed keyframes (activation is ignored, as in the reference); surfels without an observation keep their colour.
 * The reference's scratch rows 8..12 are not written (sums live in registers). */
int bslam_assign_colors(
    bslam_context* ctx, void* stream,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels);

/* Decodes all 65536 u16 image-space normal codes (BS/util.cuh:120-130) with the kernels' own routine into
 * HOST out_xyz[65536 * 3]; lets the tests compare the device's correctly rounded z = -sqrt(1 - x^2 - y^2) with the
 * CPU's bit for bit over the whole domain. */
int bslam_debug_decode_normals(bslam_context* ctx, void* stream, float* out_xyz);

/* Census for the roofline accounting of SURVEY.md 8(d): number of (surfel, keyframe) pairs
 * that pass the z > 0 and image-bounds tests (pairs that do not stop after 12 bytes), and
 * number of associated pairs.  HOST outputs, valid on return. */
int bslam_debug_count_pairs(
    bslam_context* ctx, void* stream,
    const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    uint64_t* in_bounds_pairs, uint64_t* associated_pairs);

/* Per-surfel residual probe for one keyframe (test / debugging aid): writes 8 floats
 * per surfel to DEVICE out: [depth raw residual, depth weight, descriptor r1, w1, r2,
 * w2, flags (bit0 associated, bit1 descriptor residuals valid), 0].  This is how the
 * tests check "residuals within 1e-4 relative" surfel by surfel. */
int bslam_debug_pose_residuals(
    bslam_context* ctx, void* stream,
    int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* depth_params, const bslam_keyframe_view* keyframe,
    uint32_t surfels_size, const bslam_buffer2d* surfels, float* out);

/* Point-wise probe of the device residual / Jacobian formulas (test aid): evaluates, for `count` points, the functions of
 * csrc/device_math.hpp that every kernel calls.  HOST in / out, valid on return.  Floats per point (in -> out):
 *   kind 0 depth / pose          [inv_stddev, n_local(3), lu(3), ls(3)]                        -> [raw residual, J(6)]
 *   kind 1 depth / position      [inv_stddev]                                                   -> [j]
 *   kind 2 depth / intrinsics    [inv_stddev, calibrated depth, px, py, nx, ny, n_global(3), frame_T_global row 0 (3), row 1 (3),
 *                                 n_local(3), cfactor, a, raw_inv_depth]                        -> [corrected_inv_depth, dj(6)]
 *   kind 3 descriptor / pose     [tl, tr, bl, br, tx, ty, fx, fy, ls(3)]                        -> [bilinear value, gx fx, gy fy, J(6)]
 *   kind 4 descriptor / position [tl, tr, bl, br, tx, ty, fx, fy, rn(3), ls(3)]                -> [j]
 *   kind 5 descriptor / colour intrinsics [tl, tr, bl, br, tx, ty, nx, ny]                      -> [j(4)]
 *   kind 6 = kind 0 in the pose kernel's fused-multiply-add form
 *   kind 7 arithmetic check  [x] -> [the kernels' 7-instruction correctly rounded reciprocal of x, 1.0f / x]
 * (tl .. br = the 2x2 texel footprint in [0, 1], tx, ty = fractional offsets of the sample).  tests/test_gpu_jacobians.py holds
 * these to the values of the reference's symbolic derivation (applications/badslam/scripts/jacobians_derivation.py). */
int bslam_debug_jacobians(bslam_context* ctx, void* stream, int kind, int count, const float* in, float* out);
/* Test probe of the wave reduction every per-keyframe sum goes through (wave_column_sums_lds, csrc/device_math.hpp): one wave,
 * lane l holds in[32 l .. 32 l + 32) (HOST, 64 x 32 floats); out[c] (HOST, 32 floats, in / out) receives the total of column c
 * over the 64 lanes from the lane that owns the column -- 0 for an owned column >= live_columns; columns no lane owns in the
 * configuration keep the value they had.  (live_columns, columns_per_round) must be one of the kernels' configurations:
 * (27 | 28, 4 | 8), (6, 4), (12, 4). */
int bslam_debug_wave_column_sums(bslam_context* ctx, void* stream, int live_columns, int columns_per_round, const float* in, float* out);

/* ------------------------------------------------------------------------- */
/* Surfel lifecycle                                          */
/* ------------------------------------------------------------------------- */
/* The reference resolves cell ownership with atomicCAS races; these entry points fix the outcome to
 * "lowest surfel index first" / "raster order first" (one of the outcomes the reference can produce), so
 * results are deterministic.  The supporting-surfel cell buffers (BS/direct_ba.cc:135) are library scratch. */

/* Replaces DetermineSupportingSurfelsAndMergeSurfelsCUDA (BS/kernels.h:105-117): merged surfels get
 * x = NaN (0x7fffffff); *surfel_count is decreased by their number (valid on return). */
int bslam_determine_supporting_surfels_and_merge(
    bslam_context* ctx, void* stream, float merge_dist_factor,
    const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    const bslam_keyframe_view* keyframe, uint32_t surfels_size, const bslam_buffer2d* surfels,
    uint32_t* surfel_count);

/* Replaces DetermineSupportingSurfelsCUDA + CreateSurfelsForKeyframeCUDA, i.e. the body of
 * DirectBA::CreateSurfelsForKeyframe (BS/direct_ba.cc:340-405, BS/kernels.h:94-145).
 * global_T_frame: the keyframe's pose as 3x4; covis_T_frame[i] = covis frame_T_global * global_T_frame
 * (computed by the caller, BS/direct_ba.cc:365-370; only read when filter_new_surfels).
 * Appends at column surfels_size; *new_surfel_count is valid on return.  If the new surfels do not fit,
 * nothing is created and BSLAM_ERR_OUT_OF_MEMORY is returned (the reference logs an error). */
int bslam_create_surfels_for_keyframe(
    bslam_context* ctx, void* stream, int filter_new_surfels, int min_observation_count,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* depth_params,
    const bslam_keyframe_view* keyframe, const bslam_mat3x4* global_T_frame,
    int covis_count, const bslam_keyframe_view* covis_keyframes, const bslam_mat3x4* covis_T_frame,
    uint32_t surfels_size, const bslam_buffer2d* surfels, uint32_t* new_surfel_count);

/* Replaces DeleteSurfelsAndUpdateRadiiCUDA (BS/kernels.h:271-280): surfels observed by fewer than
 * min_observation_count keyframes, or with more free-space violations than observations, get x = NaN;
 * the others get radius^2 = min over their observations. */
int bslam_delete_surfels_and_update_radii(
    bslam_context* ctx, void* stream, int min_observation_count,
    const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t* surfel_count, uint32_t surfels_size, const bslam_buffer2d* surfels);

/* Replaces CompactSurfelsCUDA (BS/kernels.h:292-299): the last valid surfels move into the free spots, in the
 * reference's order; rows 0-7 (and active_surfels, may be NULL) move, *surfels_size becomes surfel_count. */
int bslam_compact_surfels(
    bslam_context* ctx, void* stream, uint32_t surfel_count, uint32_t* surfels_size,
    const bslam_buffer2d* surfels, const bslam_buffer2d* active_surfels);

/* ------------------------------------------------------------------------- */
/* Keyframe preprocessing producers                          */
/* ------------------------------------------------------------------------- */
/* These write the u16 / half / uchar4 keyframe images the bundle adjuster reads.  Buffers are device
 * memory; input and output of one call must not alias. */

/* Replaces ComputeBrightnessCUDA (BS/cuda_image_processing.cuh, kernel BS/cuda_image_processing.cu:165-194):
 * rgb_buffer has 3 bytes per pixel, color_buffer 4 (r, g, b, luma). */
int bslam_compute_brightness(bslam_context* ctx, void* stream,
                             const bslam_buffer2d* rgb_buffer, const bslam_buffer2d* color_buffer);

/* Replaces BilateralFilteringAndDepthCutoffCUDA (BS/cuda_depth_processing.cu:42-132); the exponential is
 * evaluated with the library's deterministic exp (the reference uses -use_fast_math). */
int bslam_bilateral_filter_and_depth_cutoff(
    bslam_context* ctx, void* stream, float sigma_xy, float sigma_value, float radius_factor,
    uint16_t max_depth, float raw_to_float_depth,
    const bslam_buffer2d* input_depth, const bslam_buffer2d* output_depth);

/* Replaces ComputeNormalsCUDA (BS/cuda_depth_processing.cu:134-276). */
int bslam_compute_normals(
    bslam_context* ctx, void* stream, const bslam_camera4f* depth_camera,
    const bslam_depth_params* depth_params, const bslam_buffer2d* input_depth,
    const bslam_buffer2d* output_depth, const bslam_buffer2d* normals_buffer);

/* Replaces ComputePointRadiiAndRemoveIsolatedPixelsCUDA (BS/cuda_depth_processing.cu:278-380). */
int bslam_compute_point_radii_and_remove_isolated_pixels(
    bslam_context* ctx, void* stream, const bslam_camera4f* depth_camera, float raw_to_float_depth,
    const bslam_buffer2d* depth_buffer, const bslam_buffer2d* radius_buffer,
    const bslam_buffer2d* out_depth);

/* Replaces ComputeMinMaxDepthCUDA (BS/cuda_depth_processing.cu:382-465); results valid on return. */
int bslam_compute_min_max_depth(
    bslam_context* ctx, void* stream, const bslam_buffer2d* depth_buffer, float raw_to_float_depth,
    float* min_depth, float* max_depth);

/* ------------------------------------------------------------------------- */
/* Pairwise frame tracking / odometry                        */
/* ------------------------------------------------------------------------- */
/* The image-pair variants of the pose kernels and the pyramid construction used by TrackFramePairwise
 * (BS/pairwise_frame_tracking.cc:256-678; the coarse-to-fine loop itself is host code, see
 * badslam_amd/host/pairwise_frame_tracking.hpp).  Depth pyramids are f32 (0 = invalid), colour pyramids u8
 * single channel ("textures" of the reference are plain u8 images here), normals u16.  GradientXY variant
 * (use_gradmag = false, BS/bad_slam.cc:831). */

/* Replaces ComputeBrightnessCUDA(texture overload) (BS/cuda_image_processing.cu:196-222): luma of a uchar4 image. */
int bslam_compute_brightness_from_color(bslam_context* ctx, void* stream,
                                        const bslam_buffer2d* color_buffer, const bslam_buffer2d* intensity_buffer);
/* Replaces CUDABuffer_<u8>::SetToReadModeNormalized (LV/cuda/cuda_buffer.cu:82-102). */
int bslam_set_to_read_mode_normalized(bslam_context* ctx, void* stream,
                                      const bslam_buffer2d* input_u8, const bslam_buffer2d* output_u8);
/* Replaces CalibrateDepthCUDA (BS/kernel_downsample.cu:292-330). */
int bslam_calibrate_depth(bslam_context* ctx, void* stream, const bslam_depth_params* depth_params,
                          const bslam_buffer2d* depth_buffer, const bslam_buffer2d* out_depth);
/* Replaces CalibrateDepthAndTransformColorToDepthCUDA (BS/kernel_downsample.cu:236-290). */
int bslam_calibrate_depth_and_transform_color_to_depth(
    bslam_context* ctx, void* stream, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* depth_params, const bslam_buffer2d* depth_buffer, const bslam_buffer2d* color_u8,
    const bslam_buffer2d* out_depth, const bslam_buffer2d* out_color);
/* Replaces DownsampleImagesCUDA (BS/kernel_downsample.cu:105-234). */
int bslam_downsample_images(
    bslam_context* ctx, void* stream, const bslam_buffer2d* depth_buffer, const bslam_buffer2d* normals_buffer,
    const bslam_buffer2d* color_u8, const bslam_buffer2d* downsampled_depth,
    const bslam_buffer2d* downsampled_normals, const bslam_buffer2d* downsampled_color);
/* Replaces AccumulatePoseEstimationCoeffsFromImagesCUDA (BS/kernels.h:181-203, BS/kernel_opt_pose.cc:99-192): the
 * "downsampled" images are the tracked frame's pyramid level (colour in the colour camera's intrinsics), the
 * "surfel" images the base frame's (colour in the depth camera's intrinsics).  Cameras are the level's (scaled).
 * H (21, upper triangle row-major) and b (6) are valid on return; visible_count may be NULL. */
int bslam_accumulate_pose_coeffs_from_images(
    bslam_context* ctx, void* stream, int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, float baseline_fx, float threshold_factor,
    const bslam_buffer2d* downsampled_depth, const bslam_buffer2d* downsampled_normals, const bslam_buffer2d* downsampled_color,
    const bslam_mat3x4* estimate_frame_T_surfel_frame,
    const bslam_buffer2d* surfel_depth, const bslam_buffer2d* surfel_normals, const bslam_buffer2d* surfel_color,
    uint32_t* visible_count, float* H, float* b);
/* Replaces ComputeCostAndResidualCountFromImagesCUDA (BS/kernels.h:205-224, BS/kernel_opt_pose.cc:194-270). */
int bslam_compute_cost_and_residual_count_from_images(
    bslam_context* ctx, void* stream, int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, float baseline_fx, float threshold_factor,
    const bslam_buffer2d* downsampled_depth, const bslam_buffer2d* downsampled_normals, const bslam_buffer2d* downsampled_color,
    const bslam_mat3x4* estimate_frame_T_surfel_frame,
    const bslam_buffer2d* surfel_depth, const bslam_buffer2d* surfel_normals, const bslam_buffer2d* surfel_color,
    uint32_t* residual_count, float* residual_sum);
/* The use_gradmag = true branch of the two functions above (BS/kernel_opt_pose.cc:121-135, 212-221; kernels
 * BS/kernel_opt_pose.cu:713-937, 1173-1338): ONE colour residual per pixel, 255 x tex(tracked gradient magnitude) - base gradient
 * magnitude, on images produced by bslam_compute_sobel_gradient_magnitude instead of the brightness images. */
int bslam_accumulate_pose_coeffs_from_images_gradmag(
    bslam_context* ctx, void* stream, int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, float baseline_fx, float threshold_factor,
    const bslam_buffer2d* downsampled_depth, const bslam_buffer2d* downsampled_normals, const bslam_buffer2d* downsampled_color,
    const bslam_mat3x4* estimate_frame_T_surfel_frame,
    const bslam_buffer2d* surfel_depth, const bslam_buffer2d* surfel_normals, const bslam_buffer2d* surfel_color,
    uint32_t* visible_count, float* H, float* b);
int bslam_compute_cost_and_residual_count_from_images_gradmag(
    bslam_context* ctx, void* stream, int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, float baseline_fx, float threshold_factor,
    const bslam_buffer2d* downsampled_depth, const bslam_buffer2d* downsampled_normals, const bslam_buffer2d* downsampled_color,
    const bslam_mat3x4* estimate_frame_T_surfel_frame,
    const bslam_buffer2d* surfel_depth, const bslam_buffer2d* surfel_normals, const bslam_buffer2d* surfel_color,
    uint32_t* residual_count, float* residual_sum);
/* Replaces ComputeSobelGradientMagnitudeCUDA(stream, rgbi_texture, gradmag_buffer) (BS/cuda_image_processing.cu:104-167): Sobel
 * gradient magnitude of the luma channel of a uchar4 colour image, 0..255. */
int bslam_compute_sobel_gradient_magnitude(bslam_context* ctx, void* stream, const bslam_buffer2d* color_buffer, const bslam_buffer2d* gradmag_buffer);
/* Replaces CalibrateAndDownsampleImagesCUDA (BS/kernels.h:355-366, BS/kernel_downsample.cu:40-105, 237-270): first pyramid step of
 * a tracked frame whose level 0 is not used (use_pyramid_level_0 = false): u16 raw depth -> calibrated float depth at half
 * resolution; downsample_color = 0 when the colour image already has half the depth image's resolution. */
int bslam_calibrate_and_downsample_images(
    bslam_context* ctx, void* stream, int downsample_color, const bslam_depth_params* depth_params,
    const bslam_buffer2d* depth_buffer, const bslam_buffer2d* normals_buffer, const bslam_buffer2d* color_u8,
    const bslam_buffer2d* downsampled_depth, const bslam_buffer2d* downsampled_normals, const bslam_buffer2d* downsampled_color);

/* Multi-GPU (surfel-sharded) runs of the PCG and intrinsics entry points: every rank holds its own
 * surfel shard and the full keyframe list; `allreduce` sums a device buffer of floats in place across
 * ranks (ordered after prior work on `stream`, e.g. RCCL ncclAllReduce on that stream).  It is
 * called at fixed points with sizes that are identical on all ranks:
 *   bslam_pcg_init   1 x  [r | M] of the shared unknowns (poses, intrinsics, cfactor cells)
 *   bslam_pcg_init2  1 x  1 float (sharded part of alpha_n)
 *   bslam_pcg_step1  1 x  [g of the shared unknowns | 2 floats]
 *   bslam_pcg_step2  1 x  1 float (sharded part of beta_n)
 *   bslam_optimize_intrinsics  1 x  40 floats (A, b1, colour H, b), and with depth intrinsics 1 x  8 floats per cfactor cell
 * All ranks then hold bit-identical shared unknowns without any broadcast.  NULL (default) = single GPU.
 * The reference is single-GPU; this replaces nothing in it. */
int bslam_set_allreduce(bslam_context* ctx, bslam_allreduce_fn allreduce, void* allreduce_user);

/* The library's own exchange: an RCCL communicator owned by the context (bslam_comm_{init,destroy}).  One
 * process per GPU; rank 0 obtains a unique id and hands its 128 bytes to every rank over any channel the application has
 * (MPI, a socket, torch.distributed, a file); every rank then calls bslam_comm_init.  From then on every exchange point listed
 * above -- and the K x 32 Gauss-Newton rows of bslam_estimate_frame_poses_batched when no hook is passed to it -- is an in-place
 * ncclAllReduce(sum, float) enqueued on the call's own stream, i.e. on the BA stream: no host callback, no other runtime in
 * the loop (the C++ host class needs no Python to shard).  A hook set with bslam_set_allreduce takes precedence.  RCCL is
 * opened with dlopen at the first of these calls; the library has no link dependency on it.  The reference is single-GPU;
 * these replace nothing in it. */
#define BSLAM_COMM_UNIQUE_ID_BYTES 128
int bslam_comm_get_unique_id(void* out_id, size_t bytes);
int bslam_comm_init(bslam_context* ctx, const void* unique_id, int rank, int world_size);
int bslam_comm_destroy(bslam_context* ctx);
/* Rank and size as the communicator itself reports them (ncclCommUserRank / ncclCommCount); 0 and 1 without a communicator. */
int bslam_comm_query(bslam_context* ctx, int* rank, int* world_size);

/* ------------------------------------------------------------------------- */
/* PCG (matrix-free Gauss-Newton step)                                        */
/* ------------------------------------------------------------------------- */

/* Unknown layout of BS/direct_ba_pcg.cc:270-306:
 *   [6 per keyframe except the gauge keyframe | 1 or 3 per surfel |
 *    4 + 1 + cfactor cells | 4 colour intrinsics] */
typedef struct bslam_pcg_layout {
  uint32_t unknown_count;
  uint32_t surfel_unknown_start_index;
  uint32_t depth_intrinsics_unknown_start_index; /* 0xffffffff if unused */
  uint32_t a_unknown_index;                      /* 0xffffffff if unused */
  uint32_t color_intrinsics_unknown_start_index; /* 0xffffffff if unused */
  int32_t gauge_keyframe_id;                     /* pose of this keyframe is fixed */
  int32_t optimize_poses, optimize_geometry;
  int32_t optimize_depth_intrinsics, optimize_color_intrinsics;
  int32_t use_depth_residuals, use_descriptor_residuals;
} bslam_pcg_layout;

/* The five PCG vectors and three scalars (device, float = PCGScalar,
 * BS/kernels.cuh:62), each at least unknown_count long. */
typedef struct bslam_pcg_vectors {
  float* r;
  float* M;
  float* delta;
  float* g;
  float* p;
  float* alpha_n; /* 1 float */
  float* alpha_d; /* 1 float */
  float* beta_n;  /* 1 float */
} bslam_pcg_vectors;

/* Replaces the K x PCGInitCUDA loop + memsets (BS/kernels.h:397-416,
 * BS/direct_ba_pcg.cc:315-365): r0 = -J^T W F, M = diag(J^T W J). */
int bslam_pcg_init(
    bslam_context* ctx, void* stream, const bslam_pcg_layout* layout,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    const bslam_pcg_vectors* v);

/* Replaces PCGInit2CUDA (BS/kernels.h:418-428). */
int bslam_pcg_init2(bslam_context* ctx, void* stream, const bslam_pcg_layout* layout,
                    float a, const bslam_pcg_vectors* v);

/* Replaces the K x PCGStep1CUDA loop incl. its alpha_d / g memsets
 * (BS/kernels.h:430-452, BS/direct_ba_pcg.cc:383-425).  Quirk Q7 (the epsilon
 * term added once per keyframe) is reproduced.  A step belongs to the solve its
 * bslam_pcg_init started: the surfel buffer must not have been modified by the caller
 * since that call (the reference's solve does not touch the surfels either,
 * BS/direct_ba_pcg.cc:339-425) -- the library re-uses its sorted copy of the surfel rows
 * when the preceding call on this context was the init or a step on the same buffer. */
int bslam_pcg_step1(
    bslam_context* ctx, void* stream, const bslam_pcg_layout* layout,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes,
    uint32_t surfels_size, const bslam_buffer2d* surfels,
    const bslam_pcg_vectors* v, int clear_g);

/* Replaces PCGStep2CUDA (BS/kernels.h:454-465); *beta_n_host (HOST) valid on return. */
int bslam_pcg_step2(bslam_context* ctx, void* stream, const bslam_pcg_layout* layout,
                    const bslam_pcg_vectors* v, float* beta_n_host);

/* Replaces PCGStep3CUDA (BS/kernels.h:467-473). */
int bslam_pcg_step3(bslam_context* ctx, void* stream, const bslam_pcg_layout* layout,
                    const bslam_pcg_vectors* v);

/* Replaces UpdateSurfelsFromPCGDeltaCUDA (BS/kernels.h:483-489). */
int bslam_update_surfels_from_pcg_delta(
    bslam_context* ctx, void* stream, uint32_t surfels_size, const bslam_buffer2d* surfels,
    int use_descriptor_residuals, uint32_t surfel_unknown_start_index, const float* pcg_delta);

/* Replaces UpdateCFactorsFromPCGDeltaCUDA (BS/kernels.h:491-495). */
int bslam_update_cfactors_from_pcg_delta(
    bslam_context* ctx, void* stream, const bslam_buffer2d* cfactor_buffer,
    uint32_t cfactor_unknown_start_index, const float* pcg_delta);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* BADSLAM_HIP_H_ */
