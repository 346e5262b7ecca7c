"""ctypes mirror of include/badslam_hip.h (the C ABI of the HIP hot path).

Only POD layouts live here; nothing in this module computes anything.  The struct
layouts follow the reference types cited in the header (CUDABuffer_, CUDAMatrix3x4,
DepthParameters ... of /root/reference/applications/badslam/src/badslam).
"""
import ctypes as C

BSLAM_INVALID_DEPTH_BIT = 1 << 15
BSLAM_UNKNOWN_DEPTH = 65535
BSLAM_SURFEL_ACTIVE_FLAG = 1

SURFEL_X, SURFEL_Y, SURFEL_Z, SURFEL_NORMAL = 0, 1, 2, 3
SURFEL_RADIUS_SQUARED, SURFEL_COLOR, SURFEL_DESCRIPTOR1, SURFEL_DESCRIPTOR2 = 4, 5, 6, 7
SURFEL_ACCUM0 = 8
SURFEL_DATA_ATTRIBUTE_COUNT = 8
SURFEL_ATTRIBUTE_COUNT = 17

KF_ACTIVE, KF_COVISIBLE_ACTIVE, KF_INACTIVE = 0, 1, 2

TEX_FIXED_POINT_1_8 = 0
TEX_EXACT_FLOAT = 1

INVALID_INDEX = 0xFFFFFFFF


class Buffer2D(C.Structure):
    _fields_ = [("address", C.c_void_p), ("height", C.c_int32), ("width", C.c_int32), ("pitch", C.c_size_t)]


class Mat3x4(C.Structure):
    _fields_ = [("m", C.c_float * 12)]


class Mat3x3(C.Structure):
    _fields_ = [("m", C.c_float * 9)]


class Camera4f(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


class DepthParams(C.Structure):
    _fields_ = [("cfactor_buffer", Buffer2D), ("a", C.c_float), ("raw_to_float_depth", C.c_float),
                ("baseline_fx", C.c_float), ("sparse_surfel_cell_size", C.c_int32)]


class KeyframeView(C.Structure):
    _fields_ = [("depth", Buffer2D), ("normals", Buffer2D), ("radius", Buffer2D), ("color", Buffer2D),
                ("frame_T_global", Mat3x4), ("global_R_frame", Mat3x3),
                ("activation", C.c_int32), ("id", C.c_int32)]


class SE3f(C.Structure):
    _fields_ = [("q", C.c_float * 4), ("t", C.c_float * 3)]


class PCGLayout(C.Structure):
    _fields_ = [("unknown_count", C.c_uint32), ("surfel_unknown_start_index", C.c_uint32),
                ("depth_intrinsics_unknown_start_index", C.c_uint32), ("a_unknown_index", C.c_uint32),
                ("color_intrinsics_unknown_start_index", C.c_uint32), ("gauge_keyframe_id", C.c_int32),
                ("optimize_poses", C.c_int32), ("optimize_geometry", C.c_int32),
                ("optimize_depth_intrinsics", C.c_int32), ("optimize_color_intrinsics", C.c_int32),
                ("use_depth_residuals", C.c_int32), ("use_descriptor_residuals", C.c_int32)]


class PCGVectors(C.Structure):
    _fields_ = [("r", C.c_void_p), ("M", C.c_void_p), ("delta", C.c_void_p), ("g", C.c_void_p), ("p", C.c_void_p),
                ("alpha_n", C.c_void_p), ("alpha_d", C.c_void_p), ("beta_n", C.c_void_p)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

P = C.POINTER

# name -> (restype, argtypes); the single source the loader and the symbol test use.
# ctx and stream are void*.
_CAM = P(Camera4f)
_DP = P(DepthParams)
_BUF = P(Buffer2D)
_KFS = P(KeyframeView)
SIGNATURES = {
    "bslam_abi_version": (C.c_int, []),
    "bslam_last_error": (C.c_char_p, []),
    "bslam_create": (C.c_int, [C.c_int, P(C.c_void_p)]),
    "bslam_destroy": (C.c_int, [C.c_void_p]),
    "bslam_set_texture_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "bslam_set_xcd_schedule": (C.c_int, [C.c_void_p, C.c_int]),
    "bslam_set_keyframe_cache": (C.c_int, [C.c_void_p, C.c_int]),
    "bslam_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "bslam_determine_supporting_surfels_and_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, P(Camera4f), P(DepthParams), P(KeyframeView),
                                                              C.c_uint32, P(Buffer2D), P(C.c_uint32)]),
    "bslam_create_surfels_for_keyframe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, P(Camera4f), P(Camera4f), P(DepthParams),
                                                   P(KeyframeView), P(Mat3x4), C.c_int, P(KeyframeView), P(Mat3x4), C.c_uint32, P(Buffer2D),
                                                   P(C.c_uint32)]),
    "bslam_delete_surfels_and_update_radii": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, P(Camera4f), P(DepthParams), C.c_int, P(KeyframeView),
                                                       P(C.c_uint32), C.c_uint32, P(Buffer2D)]),
    "bslam_compute_brightness": (C.c_int, [C.c_void_p, C.c_void_p, P(Buffer2D), P(Buffer2D)]),
    "bslam_bilateral_filter_and_depth_cutoff": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_uint16, C.c_float,
                                                         P(Buffer2D), P(Buffer2D)]),
    "bslam_compute_normals": (C.c_int, [C.c_void_p, C.c_void_p, P(Camera4f), P(DepthParams), P(Buffer2D), P(Buffer2D), P(Buffer2D)]),
    "bslam_compute_point_radii_and_remove_isolated_pixels": (C.c_int, [C.c_void_p, C.c_void_p, P(Camera4f), C.c_float, P(Buffer2D), P(Buffer2D),
                                                                      P(Buffer2D)]),
    "bslam_compute_min_max_depth": (C.c_int, [C.c_void_p, C.c_void_p, P(Buffer2D), C.c_float, P(C.c_float), P(C.c_float)]),
    "bslam_compute_brightness_from_color": (C.c_int, [C.c_void_p, C.c_void_p, P(Buffer2D), P(Buffer2D)]),
    "bslam_set_to_read_mode_normalized": (C.c_int, [C.c_void_p, C.c_void_p, P(Buffer2D), P(Buffer2D)]),
    "bslam_calibrate_depth": (C.c_int, [C.c_void_p, C.c_void_p, P(DepthParams), P(Buffer2D), P(Buffer2D)]),
    "bslam_calibrate_depth_and_transform_color_to_depth": (C.c_int, [C.c_void_p, C.c_void_p, P(Camera4f), P(Camera4f), P(DepthParams), P(Buffer2D),
                                                                    P(Buffer2D), P(Buffer2D), P(Buffer2D)]),
    "bslam_downsample_images": (C.c_int, [C.c_void_p, C.c_void_p, P(Buffer2D), P(Buffer2D), P(Buffer2D), P(Buffer2D), P(Buffer2D), P(Buffer2D)]),
    "bslam_accumulate_pose_coeffs_from_images": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, P(Camera4f), P(Camera4f), C.c_float, C.c_float,
                                                          P(Buffer2D), P(Buffer2D), P(Buffer2D), P(Mat3x4), P(Buffer2D), P(Buffer2D), P(Buffer2D),
                                                          P(C.c_uint32), P(C.c_float), P(C.c_float)]),
    "bslam_compute_cost_and_residual_count_from_images": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, P(Camera4f), P(Camera4f), C.c_float,
                                                                   C.c_float, P(Buffer2D), P(Buffer2D), P(Buffer2D), P(Mat3x4), P(Buffer2D),
                                                                   P(Buffer2D), P(Buffer2D), P(C.c_uint32), P(C.c_float)]),
    "bslam_accumulate_pose_coeffs_from_images_gradmag": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, P(Camera4f), P(Camera4f), C.c_float, C.c_float,
                                                                  P(Buffer2D), P(Buffer2D), P(Buffer2D), P(Mat3x4), P(Buffer2D), P(Buffer2D), P(Buffer2D),
                                                                  P(C.c_uint32), P(C.c_float), P(C.c_float)]),
    "bslam_compute_cost_and_residual_count_from_images_gradmag": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, P(Camera4f), P(Camera4f), C.c_float,
                                                                           C.c_float, P(Buffer2D), P(Buffer2D), P(Buffer2D), P(Mat3x4), P(Buffer2D),
                                                                           P(Buffer2D), P(Buffer2D), P(C.c_uint32), P(C.c_float)]),
    "bslam_compute_sobel_gradient_magnitude": (C.c_int, [C.c_void_p, C.c_void_p, P(Buffer2D), P(Buffer2D)]),
    "bslam_calibrate_and_downsample_images": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, P(DepthParams), P(Buffer2D), P(Buffer2D), P(Buffer2D), P(Buffer2D),
                                                       P(Buffer2D), P(Buffer2D)]),
    "bslam_compact_surfels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, P(C.c_uint32), P(Buffer2D), P(Buffer2D)]),
    "bslam_invalidate_keyframe_cache": (C.c_int, [C.c_void_p]),
    "bslam_comm_get_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "bslam_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "bslam_comm_destroy": (C.c_int, [C.c_void_p]),
    "bslam_comm_query": (C.c_int, [C.c_void_p, P(C.c_int), P(C.c_int)]),
    "bslam_set_culling": (C.c_int, [C.c_void_p, C.c_int]),
    "bslam_set_pose_keyframe_list": (C.c_int, [C.c_void_p, C.c_int]),
    "bslam_debug_cull_stats": (C.c_int, [C.c_void_p, P(C.c_uint64), P(C.c_uint64)]),
    "bslam_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "bslam_profile_read": (C.c_int, [C.c_void_p, C.c_int, P(C.c_int32), P(C.c_float)]),
    "bslam_profile_read_counters": (C.c_int, [C.c_void_p, P(C.c_uint64)]),
    "bslam_set_geometry_keyframe_chunk": (C.c_int, [C.c_void_p, C.c_int]),
    "bslam_assign_colors": (C.c_int, [C.c_void_p, C.c_void_p, _CAM, _CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF]),
    "bslam_debug_decode_normals": (C.c_int, [C.c_void_p, C.c_void_p, P(C.c_float)]),
    "bslam_debug_jacobians": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, P(C.c_float), P(C.c_float)]),
    "bslam_debug_wave_column_sums": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "bslam_debug_count_pairs": (C.c_int, [
        C.c_void_p, C.c_void_p, _CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, P(C.c_uint64), P(C.c_uint64)]),
    "bslam_accumulate_pose_estimation_coeffs": (C.c_int, [
        C.c_void_p, C.c_void_p, C.c_int, C.c_int, _CAM, _CAM, _DP, _BUF, _BUF, _BUF, P(Mat3x4),
        C.c_uint32, _BUF, C.c_int, P(C.c_uint32), P(C.c_float), P(C.c_float), P(C.c_float)]),
    "bslam_estimate_frame_poses_batched": (C.c_int, [
        C.c_void_p, C.c_void_p, C.c_int, C.c_int, _CAM, _CAM, _DP, C.c_int, _KFS,
        C.c_uint32, _BUF, C.c_int, P(SE3f), P(C.c_int32), P(C.c_int32), ALLREDUCE_FN, C.c_void_p]),
    "bslam_accumulate_pose_coeffs_batched": (C.c_int, [
        C.c_void_p, C.c_void_p, C.c_int, C.c_int, _CAM, _CAM, _DP, C.c_int, _KFS,
        C.c_uint32, _BUF, P(C.c_float), P(C.c_uint32)]),
    "bslam_update_surfel_activation": (C.c_int, [
        C.c_void_p, C.c_void_p, _CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, _BUF]),
    "bslam_update_surfel_normals": (C.c_int, [
        C.c_void_p, C.c_void_p, _CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, _BUF]),
    "bslam_optimize_geometry_iteration": (C.c_int, [
        C.c_void_p, C.c_void_p, C.c_int, C.c_int, _CAM, _CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, _BUF]),
    "bslam_optimize_intrinsics": (C.c_int, [
        C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, _KFS, _CAM, _CAM, _DP, C.c_uint32, _BUF, _CAM, _CAM, P(C.c_float)]),
    "bslam_debug_association": (C.c_int, [
        C.c_void_p, C.c_void_p, _CAM, _DP, _KFS, C.c_uint32, _BUF, C.c_void_p]),
    "bslam_debug_pose_residuals": (C.c_int, [
        C.c_void_p, C.c_void_p, C.c_int, C.c_int, _CAM, _CAM, _DP, _KFS, C.c_uint32, _BUF, C.c_void_p]),
    "bslam_pcg_init": (C.c_int, [
        C.c_void_p, C.c_void_p, P(PCGLayout), _CAM, _CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, P(PCGVectors)]),
    "bslam_pcg_init2": (C.c_int, [C.c_void_p, C.c_void_p, P(PCGLayout), C.c_float, P(PCGVectors)]),
    "bslam_pcg_step1": (C.c_int, [
        C.c_void_p, C.c_void_p, P(PCGLayout), _CAM, _CAM, _DP, C.c_int, _KFS, C.c_uint32, _BUF, P(PCGVectors), C.c_int]),
    "bslam_pcg_step2": (C.c_int, [C.c_void_p, C.c_void_p, P(PCGLayout), P(PCGVectors), P(C.c_float)]),
    "bslam_pcg_step3": (C.c_int, [C.c_void_p, C.c_void_p, P(PCGLayout), P(PCGVectors)]),
    "bslam_update_surfels_from_pcg_delta": (C.c_int, [
        C.c_void_p, C.c_void_p, C.c_uint32, _BUF, C.c_int, C.c_uint32, C.c_void_p]),
    "bslam_update_cfactors_from_pcg_delta": (C.c_int, [C.c_void_p, C.c_void_p, _BUF, C.c_uint32, C.c_void_p]),
}
