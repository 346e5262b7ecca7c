// io.hpp -- file formats on either side of the BA path: TUM RGB-D dataset directories
// (associated.txt, calibration.txt, trajectory, 16-bit depth / 8-bit colour PNGs), pose export and the
// calibration export / import.  Host-only C++; the PNG decoder uses zlib.  The BadSlam state file (BS/io.cc:38-536)
// serialises the whole SLAM front end (config, odometry state) and is not part of this module.
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

}  // namespace bslam_host
