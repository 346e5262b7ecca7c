// pairwise_frame_tracking.hpp -- host side of the odometry: the coarse-to-fine direct alignment
// of a tracked frame against a base keyframe.  Mirrors vis::TrackFramePairwise (BS/pairwise_frame_tracking.cc:256-678)
// with both of its switches (use_pyramid_level_0, use_gradmag; BadSlam::RunOdometry passes true / false), plus the input
// preparation BadSlam::RunOdometry does before the call (BS/bad_slam.cc:859-897).  The kernels are behind the C ABI (include/badslam_hip.h, odometry section).
#pragma once

#include <memory>
#include <vector>

#include "direct_ba.hpp"

namespace bslam_host {

// vis::PairwiseFrameTrackingBuffers (BS/pairwise_frame_tracking.h): per-scale images of both frames
struct PairwiseFrameTrackingBuffers {
  PairwiseFrameTrackingBuffers(int depth_width, int depth_height, int color_width, int color_height, int num_scales);
  int num_scales;
  std::vector<std::unique_ptr<DeviceBuffer<float>>> base_depth, tracked_depth;
  std::vector<std::unique_ptr<DeviceBuffer<u16>>> base_normals, tracked_normals;   // level 0 unused: the frames' own buffers
  std::vector<std::unique_ptr<DeviceBuffer<u8>>> base_color, tracked_color;
  std::unique_ptr<DeviceBuffer<u8>> base_gradmag, tracked_gradmag;                  // luma of the two uchar4 images
};

// out_base_T_frame: pose of the tracked frame in the base frame.  iterations_per_scale (may be null) receives
// num_scales entries.  Inputs are the keyframe-format device images (depth u16, normals u16, colour uchar4).
void TrackFramePairwise(bslam_context* ctx, hipStream_t stream, PairwiseFrameTrackingBuffers* buffers, const PinholeCamera4f& color_camera,
                        const PinholeCamera4f& depth_camera, const bslam_depth_params& depth_params, bool use_depth_residuals,
                        bool use_descriptor_residuals, const DeviceBuffer<u16>& tracked_depth, const DeviceBuffer<u16>& tracked_normals,
                        const DeviceBuffer<uchar4_t>& tracked_color, const DeviceBuffer<u16>& base_depth, const DeviceBuffer<u16>& base_normals,
                        const DeviceBuffer<uchar4_t>& base_color, bool test_different_initial_estimates, const SE3f& base_T_frame_initial_estimate_1,
                        const SE3f& base_T_frame_initial_estimate_2, SE3f* out_base_T_frame, int* iterations_per_scale,
                        bool use_pyramid_level_0 = true, bool use_gradmag = false);

}  // namespace bslam_host
