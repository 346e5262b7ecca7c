// io.hpp -- file formats on either side of the BA path: TUM RGB-D dataset directories
// (associated.txt, calibration.txt, trajectory, 16-bit depth / 8-bit colour PNGs), pose export, the
// calibration export / import and the binary state file v1 (BS/io.cc:38-536).  Host-only C++; the PNG decoder
// uses zlib.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "se3.hpp"

namespace bslam_host {

// ---- PNG (non-interlaced; 8-bit RGB / RGBA / gray are returned as rgb, 16-bit gray as u16) ----
struct PngInfo { int width = 0, height = 0, bit_depth = 0, channels = 0; };
bool ReadPngInfo(const std::string& path, PngInfo* info);
// 8-bit colour image as tightly packed rgb (3 bytes per pixel); gray is replicated, alpha dropped
bool ReadPngRgb8(const std::string& path, int* width, int* height, std::vector<uint8_t>* rgb);
// 16-bit (or 8-bit) single-channel image as u16 (big-endian samples swapped to host order)
bool ReadPngGray16(const std::string& path, int* width, int* height, std::vector<uint16_t>* gray);

// ---- TUM RGB-D (LV/rgbd_video_io_tum_dataset.h:40-240) ----
// InterpolatePose (:42-72): clamps outside the trajectory, slerp + lerp inside
bool InterpolatePose(double timestamp, const std::vector<double>& pose_timestamps, const std::vector<SE3f>& poses, SE3f* pose);
// ReadTUMRGBDTrajectory (:74-126): lines "timestamp tx ty tz qx qy qz qw", '#' comments
bool ReadTUMRGBDTrajectory(const std::string& path, std::vector<double>* pose_timestamps, std::vector<SE3f>* poses_global_T_frame);

struct TumFrame {
  std::string rgb_timestamp_string, depth_timestamp_string;
  double rgb_timestamp = 0, depth_timestamp = 0;
  std::string rgb_path, depth_path;
  SE3f rgb_global_T_frame, depth_global_T_frame;   // identity without a trajectory
};
struct TumDataset {
  int width = 0, height = 0;
  float camera_parameters[4] = {0, 0, 0, 0};   // fx fy cx+0.5 cy+0.5: the pixel-corner convention of PinholeCamera4f (:230-233)
  std::vector<TumFrame> frames;
};
// ReadTUMRGBDDatasetAssociatedAndCalibrated (:128-240); trajectory_filename may be empty
bool ReadTUMRGBDDatasetAssociatedAndCalibrated(const std::string& dataset_folder_path, const std::string& trajectory_filename, TumDataset* dataset);

// ---- exports (BS/io.cc:537-624) ----
// SavePoses (:537-568): one line per frame, "timestamp tx ty tz qx qy qz qw" of start_frame_T_global * global_T_frame, 17 significant digits
bool SavePoses(const std::vector<std::string>& timestamp_strings, const std::vector<SE3f>& global_T_frame, int start_frame, const std::string& path);
// SaveCalibration (:570-624) / LoadCalibration (:626-700): <base>.depth_intrinsics.txt, <base>.color_intrinsics.txt
// ("fx fy cx-0.5 cy-0.5") and <base>.deformation.txt ("w h", a, then w*h cfactors row-major)
bool SaveCalibration(const std::string& base_path, const float depth_camera[4], const float color_camera[4], float a, int cfactor_width,
                     int cfactor_height, const float* cfactor_row_major);
bool LoadCalibration(const std::string& base_path, float depth_camera[4], float color_camera[4], float* a, int cfactor_width, int cfactor_height,
                     float* cfactor_row_major);

// ---- state file v1 (BS/io.cc:38-536, config block BS/bad_slam_config.cc:28-200) ----
// The file holds the SLAM configuration, the per-frame poses of the video, the cameras, the depth deformation, the
// keyframe list (metadata only: keyframe images are re-created from the dataset on load) and rows 0-7 of the
// surfel buffer.  All integers are little-endian i32 / u32, bools one byte, SE3f = 7 floats (qx qy qz qw tx ty tz).
struct BadSlamConfigV1 {   // field order = file order (BS/bad_slam_config.cc:47-96)
  float raw_to_float_depth = 1.f / 5000.f;
  int32_t start_frame = 0, end_frame = 0x7fffffff;
  float target_frame_rate = 0;
  int32_t fps_restriction = 30, pyramid_level_for_depth = 0, pyramid_level_for_color = 0;
  float max_depth = 3.f, baseline_fx = 40.f;
  int32_t median_filter_and_densify_iterations = 0;
  float bilateral_filter_sigma_xy = 3.f, bilateral_filter_radius_factor = 2.f, bilateral_filter_sigma_inv_depth = 0.005f;
  int32_t max_surfel_count = 25 * 1000 * 1000, sparse_surfel_cell_size = 4;
  float surfel_merge_dist_factor = 0.8f;
  int32_t min_observation_count_while_bootstrapping_1 = 1, min_observation_count_while_bootstrapping_2 = 2, min_observation_count = 2, num_scales = 5;
  bool use_motion_model = true;
  int32_t keyframe_interval = 10, max_num_ba_iterations_per_keyframe = 10;
  bool disable_deactivation = false, use_geometric_residuals = true, use_photometric_residuals = true, optimize_intrinsics = false;
  int32_t intrinsics_optimization_interval = 10;
  bool do_surfel_updates = true, parallel_ba = true, use_pcg = false, estimate_poses = true;
  int32_t min_free_gpu_memory_mb = 250;
  bool enable_loop_detection = true, parallel_loop_detection = true;
  std::string loop_detection_vocabulary_path, loop_detection_pattern_path;
  float loop_detection_image_frequency = 0.5f;
  int32_t loop_detection_images_width = 640, loop_detection_images_height = 480;
};
struct StateKeyframeV1 {
  int32_t id = -1;   // -1: deleted keyframe (no further fields in the file)
  int32_t frame_index = 0, activation = 0, last_active_in_ba_iteration = -1, last_covis_in_ba_iteration = -1;
};
struct StateV1 {
  int32_t base_kf_id = -1;
  std::vector<SE3f> motion_model_base_kf_tr_frame;
  std::vector<int32_t> queued_keyframes_frame_indices;
  std::vector<SE3f> queued_keyframes_last_kf_tr_this_kf;
  int32_t last_frame_index = 0;
  BadSlamConfigV1 config;
  std::vector<SE3f> frame_global_T_frame;                 // one pose per video frame
  int32_t color_camera_width = 0, color_camera_height = 0, depth_camera_width = 0, depth_camera_height = 0, pyramid_level_for_color = 0;
  float color_camera_parameters[4] = {0, 0, 0, 0}, depth_camera_parameters[4] = {0, 0, 0, 0};
  int32_t cfactor_width = 0, cfactor_height = 0;
  std::vector<float> cfactor;                             // row-major, cfactor_width floats per row (the file's stride padding is dropped)
  float a = 0, raw_to_float_depth = 0, baseline_fx = 0;
  int32_t sparse_surfel_cell_size = 0;
  std::vector<StateKeyframeV1> keyframes;
  int32_t surfel_count = 0, surfels_size = 0;
  std::vector<float> surfels;                             // 8 rows x surfels_size, row-major
  int32_t ba_iteration_count = 0, last_ba_iteration_count = -1;
  bool use_depth_residuals = true, use_descriptor_residuals = true;
  int32_t min_observation_count_while_bootstrapping_1 = 1, min_observation_count_while_bootstrapping_2 = 2, min_observation_count = 2;
  float surfel_merge_dist_factor = 0.8f;
};
bool SaveState(const StateV1& state, const std::string& path);
bool LoadState(const std::string& path, StateV1* state, std::string* error);

}  // namespace bslam_host
