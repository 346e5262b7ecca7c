// bad_slam.hpp -- the sequential front end around DirectBA: vis::BadSlam (BS/bad_slam.{h,cc}) without its threads,
// GUI and loop detector.  Per frame: device preprocessing (BS/bad_slam.cc:639-760), pairwise tracking against the last
// keyframe with the constant-motion initial estimates (:763-950), a keyframe every `keyframe_interval` frames
// (:953-1097) and the planned bundle-adjustment iterations (:212-282, :481-536), after which the poses of the
// non-keyframes follow their neighbouring keyframes (BS/trajectory_deformation.cc:45-146).
//
// Not built: parallel_ba (BA thread), real-time pacing (target_frame_rate), loop detection / pose-graph
// optimisation (DBoW2, opengv, g2o are not in this image), median_filter_and_densify_iterations > 0,
// pyramid_level_for_depth / _color > 0, keyframe merging on low memory.  Those switches must keep their "off" values.
#pragma once

#include <memory>
#include <vector>

#include "direct_ba.hpp"
#include "io.hpp"
#include "pairwise_frame_tracking.hpp"

namespace bslam_host {

class BadSlam {
 public:
  // config: BS/bad_slam_config.h (defaults of BadSlamConfigV1).  Cameras in the pixel-corner convention of
  // PinholeCamera4f, already scaled to the pyramid levels in use (level 0 only).
  BadSlam(const BadSlamConfigV1& config, const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera, int device = 0);
  ~BadSlam();

  // vis::BadSlam::ProcessFrame (BS/bad_slam.cc:170-282).  depth_image: raw u16 depth (0 = no measurement, as in the
  // dataset PNGs), rgb_image: 3 bytes per pixel; both HOST arrays of the cameras' sizes.  Frames must arrive with
  // consecutive indices starting at config.start_frame.
  void ProcessFrame(int frame_index, const u16* depth_image, const u8* rgb_image, bool force_keyframe = false);

  // vis::BadSlam::RunBundleAdjustment (BS/bad_slam.cc:481-536)
  void RunBundleAdjustment(u32 frame_index, bool optimize_depth_intrinsics, bool optimize_color_intrinsics, bool optimize_poses, bool optimize_geometry,
                           int min_iterations, int max_iterations, int active_keyframe_window_start, int active_keyframe_window_end,
                           bool increase_ba_iteration_count, int* iterations_done, bool* converged);

  DirectBA& direct_ba() { return *direct_ba_; }
  const BadSlamConfigV1& config() const { return config_; }
  // global_T_frame of every processed frame (index = frame_index - config.start_frame)
  const std::vector<SE3f>& frame_poses() const { return frame_global_T_frame_; }
  int last_frame_index() const { return last_frame_index_; }
  bool keyframe_created() const { return keyframe_created_; }
  bool pose_estimated() const { return pose_estimated_; }
  int num_planned_ba_iterations() const { return num_planned_ba_iterations_; }
  const Keyframe* base_kf() const { return base_kf_; }
  const std::vector<SE3f>& motion_model_base_kf_tr_frame() const { return base_kf_tr_frame_; }

 private:
  void PreprocessFrame(const u16* depth_image, const u8* rgb_image);                       // :639-760
  void PredictFramePose(SE3f* estimate_1, SE3f* estimate_2) const;                          // :763-825
  void RunOdometry(int frame_index);                                                        // :827-950
  std::shared_ptr<Keyframe> CreateKeyframe(int frame_index);                                // :953-1097
  SE3f& FramePose(int frame_index) { return frame_global_T_frame_.at(static_cast<size_t>(frame_index - config_.start_frame)); }

  BadSlamConfigV1 config_;
  std::unique_ptr<DirectBA> direct_ba_;
  hipStream_t stream_ = nullptr;

  // buffers of the frame being processed (BS/bad_slam.h:283-296)
  std::unique_ptr<DeviceBuffer<u8>> rgb_buffer_;
  std::unique_ptr<DeviceBuffer<uchar4_t>> color_buffer_;
  std::unique_ptr<DeviceBuffer<u16>> depth_buffer_, filtered_depth_buffer_A_, filtered_depth_buffer_B_, normals_buffer_, radius_buffer_;
  std::unique_ptr<PairwiseFrameTrackingBuffers> pairwise_tracking_buffers_;

  Keyframe* base_kf_ = nullptr;
  SE3f base_kf_global_T_frame_;
  std::vector<SE3f> base_kf_tr_frame_;   // motion model: poses of the last (up to three) frames relative to the base keyframe, newest last
  std::vector<SE3f> frame_global_T_frame_;
  int last_frame_index_ = -1;
  int num_planned_ba_iterations_ = 0;
  int bundle_adjustment_counter_ = 0;
  bool pose_estimated_ = false, keyframe_created_ = false;
};

// BS/trajectory_deformation.cc:33-43 / :45-146 on a plain pose vector (frame_poses[i] = pose of frame start_frame + i)
void RememberKeyframePoses(const DirectBA& ba, std::vector<SE3f>* original_keyframe_T_global);
void ExtrapolateAndInterpolateKeyframePoseChanges(u32 start_frame, u32 end_frame, const DirectBA& ba, const std::vector<SE3f>& original_keyframe_T_global,
                                                  std::vector<SE3f>* frame_poses);

}  // namespace bslam_host
