// bad_slam.cpp -- see bad_slam.hpp.  Host logic only; every image operation goes through the C ABI.
#include "bad_slam.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>

namespace bslam_host {

namespace {

void CheckRc(int rc, const char* what) {
  if (rc != BSLAM_OK) throw std::runtime_error(std::string(what) + ": " + bslam_last_error());
}

void CheckHip(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// Eigen QuaternionBase::slerp (Eigen/src/Geometry/Quaternion.h), then Sophus' setQuaternion (normalise)
void SlerpInto(const SE3f& a, const SE3f& b, float t, SE3f* out) {
  const float one = 1.0f - 1.1920929e-07f;
  const float d = a.qx * b.qx + a.qy * b.qy + a.qz * b.qz + a.qw * b.qw;
  const float abs_d = std::fabs(d);
  float scale0, scale1;
  if (abs_d >= one) {
    scale0 = 1.0f - t;
    scale1 = t;
  } else {
    const float theta = std::acos(abs_d);
    const float sin_theta = std::sin(theta);
    scale0 = std::sin((1.0f - t) * theta) / sin_theta;
    scale1 = std::sin(t * theta) / sin_theta;
  }
  if (d < 0) scale1 = -scale1;
  float q[4] = {scale0 * a.qx + scale1 * b.qx, scale0 * a.qy + scale1 * b.qy, scale0 * a.qz + scale1 * b.qz, scale0 * a.qw + scale1 * b.qw};
  const float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  out->qx = q[0] / n; out->qy = q[1] / n; out->qz = q[2] / n; out->qw = q[3] / n;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// BS/trajectory_deformation.cc
// ------------------------------------------------------------------------------------------------
void RememberKeyframePoses(const DirectBA& ba, std::vector<SE3f>* original_keyframe_T_global) {
  original_keyframe_T_global->resize(ba.keyframes().size());
  for (size_t i = 0; i < ba.keyframes().size(); ++i)
    if (ba.keyframes()[i]) (*original_keyframe_T_global)[i] = ba.keyframes()[i]->frame_T_global();
}

void ExtrapolateAndInterpolateKeyframePoseChanges(u32 start_frame, u32 end_frame, const DirectBA& ba, const std::vector<SE3f>& original_keyframe_T_global,
                                                  std::vector<SE3f>* frame_poses) {
  const auto& keyframes = ba.keyframes();
  if (frame_poses->empty() || keyframes.empty()) return;
  end_frame = std::min<u32>(end_frame, start_frame + static_cast<u32>(frame_poses->size()) - 1);
  size_t prev_keyframe_index = 0, next_keyframe_index = 0;
  // (the reference dereferences keyframes[0] unconditionally; it is never deleted, BS/direct_ba.cc:300-303)
  for (u32 other_frame_index = start_frame; other_frame_index <= end_frame; ++other_frame_index) {
    while (next_keyframe_index < keyframes.size() && keyframes[next_keyframe_index]->frame_index() <= other_frame_index) {
      prev_keyframe_index = next_keyframe_index;
      ++next_keyframe_index;
      while (next_keyframe_index < keyframes.size() && !keyframes[next_keyframe_index]) ++next_keyframe_index;
    }
    const Keyframe* prev_keyframe = keyframes[prev_keyframe_index].get();
    const Keyframe* next_keyframe = (next_keyframe_index < keyframes.size()) ? keyframes[next_keyframe_index].get() : nullptr;
    if (prev_keyframe->frame_index() == other_frame_index) continue;   // a keyframe itself

    SE3f& global_T_other = (*frame_poses)[other_frame_index - start_frame];
    SE3f new_global_T_other_frame;
    if (next_keyframe == nullptr || prev_keyframe->frame_index() > other_frame_index) {   // extrapolate at either end
      const SE3f old_kf_T_other_frame = original_keyframe_T_global[prev_keyframe_index] * global_T_other;
      new_global_T_other_frame = prev_keyframe->global_T_frame() * old_kf_T_other_frame;
    } else {                                                                                // interpolate
      const SE3f other_T_global = global_T_other.Inverse();
      const SE3f from_prev = other_T_global * (prev_keyframe->global_T_frame() * (original_keyframe_T_global[prev_keyframe_index] * global_T_other));
      const SE3f from_next = other_T_global * (next_keyframe->global_T_frame() * (original_keyframe_T_global[next_keyframe_index] * global_T_other));
      const float factor = (other_frame_index - prev_keyframe->frame_index()) * 1.0f / (next_keyframe->frame_index() - prev_keyframe->frame_index());
      SE3f interpolated;
      interpolated.tx = (1 - factor) * from_prev.tx + factor * from_next.tx;
      interpolated.ty = (1 - factor) * from_prev.ty + factor * from_next.ty;
      interpolated.tz = (1 - factor) * from_prev.tz + factor * from_next.tz;
      SlerpInto(from_prev, from_next, factor, &interpolated);
      new_global_T_other_frame = global_T_other * interpolated;
    }
    global_T_other = new_global_T_other_frame;
  }
}

// ------------------------------------------------------------------------------------------------
// BadSlam
// ------------------------------------------------------------------------------------------------
BadSlam::BadSlam(const BadSlamConfigV1& config, const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera, int device) : config_(config) {
  if (config_.parallel_ba) config_.parallel_ba = false;   // sequential only (see the header)
  if (config_.enable_loop_detection) config_.enable_loop_detection = false;
  if (config_.target_frame_rate != 0) throw std::invalid_argument("BadSlam: real-time pacing (target_frame_rate) is not built");
  if (config_.median_filter_and_densify_iterations != 0) throw std::invalid_argument("BadSlam: median_filter_and_densify_iterations must be 0");
  if (config_.pyramid_level_for_depth != 0 || config_.pyramid_level_for_color != 0) throw std::invalid_argument("BadSlam: pyramid levels for depth / colour must be 0");
  if (config_.keyframe_interval < 1) throw std::invalid_argument("BadSlam: keyframe_interval must be >= 1");
  // BS/bad_slam.cc:121-138
  direct_ba_.reset(new DirectBA(config_.max_surfel_count, config_.raw_to_float_depth, config_.baseline_fx, config_.sparse_surfel_cell_size,
                                config_.surfel_merge_dist_factor, config_.min_observation_count_while_bootstrapping_1,
                                config_.min_observation_count_while_bootstrapping_2, config_.min_observation_count, color_camera, depth_camera,
                                config_.pyramid_level_for_color, config_.use_geometric_residuals, config_.use_photometric_residuals, nullptr, SE3f(),
                                device));
  const int dw = depth_camera.width(), dh = depth_camera.height(), cw = color_camera.width(), ch = color_camera.height();
  rgb_buffer_.reset(new DeviceBuffer<u8>(ch, cw * 3));
  color_buffer_.reset(new DeviceBuffer<uchar4_t>(ch, cw));
  depth_buffer_.reset(new DeviceBuffer<u16>(dh, dw));
  filtered_depth_buffer_A_.reset(new DeviceBuffer<u16>(dh, dw));
  filtered_depth_buffer_B_.reset(new DeviceBuffer<u16>(dh, dw));
  normals_buffer_.reset(new DeviceBuffer<u16>(dh, dw));
  radius_buffer_.reset(new DeviceBuffer<u16>(dh, dw));
  pairwise_tracking_buffers_.reset(new PairwiseFrameTrackingBuffers(dw, dh, cw, ch, config_.num_scales));
}

BadSlam::~BadSlam() = default;

void BadSlam::PreprocessFrame(const u16* depth_image, const u8* rgb_image) {
  bslam_context* ctx = direct_ba_->context();
  const PinholeCamera4f& depth_camera = direct_ba_->depth_camera();
  const int dw = depth_camera.width(), cw = direct_ba_->color_camera().width();
  depth_buffer_->Upload(stream_, depth_image, static_cast<size_t>(dw) * sizeof(u16));
  rgb_buffer_->Upload(stream_, rgb_image, static_cast<size_t>(cw) * 3);
  bslam_buffer2d rgb_pod = rgb_buffer_->ToPod();
  rgb_pod.width = cw;   // 3 bytes per pixel
  const bslam_buffer2d color_pod = color_buffer_->ToPod(), depth_pod = depth_buffer_->ToPod(), a_pod = filtered_depth_buffer_A_->ToPod(),
                       b_pod = filtered_depth_buffer_B_->ToPod(), normals_pod = normals_buffer_->ToPod(), radius_pod = radius_buffer_->ToPod();
  CheckRc(bslam_compute_brightness(ctx, stream_, &rgb_pod, &color_pod), "bslam_compute_brightness");
  CheckRc(bslam_bilateral_filter_and_depth_cutoff(ctx, stream_, config_.bilateral_filter_sigma_xy, config_.bilateral_filter_sigma_inv_depth,
                                                  config_.bilateral_filter_radius_factor,
                                                  static_cast<uint16_t>(config_.max_depth / config_.raw_to_float_depth), config_.raw_to_float_depth,
                                                  &depth_pod, &a_pod),
          "bslam_bilateral_filter_and_depth_cutoff");
  const bslam_camera4f depth_cam = depth_camera.pod();
  const bslam_depth_params dp = direct_ba_->depth_params();
  CheckRc(bslam_compute_normals(ctx, stream_, &depth_cam, &dp, &a_pod, &b_pod, &normals_pod), "bslam_compute_normals");
  CheckRc(bslam_compute_point_radii_and_remove_isolated_pixels(ctx, stream_, &depth_cam, config_.raw_to_float_depth, &b_pod, &radius_pod, &a_pod),
          "bslam_compute_point_radii_and_remove_isolated_pixels");
  // the frame's final depth image is filtered_depth_buffer_A_ (:759)
}

// Motion model (BS/bad_slam.cc:903-951): the two start poses handed to the tracker.  Kept here as ONE short history of frame
// poses relative to the base keyframe, newest last; the inter-frame motions are formed from it on demand.
//   estimate 1: the last inter-frame motion applied once more to the last frame,
//   estimate 2: the motion before that applied twice to the frame before the last (i.e. "ignore the last frame").
void BadSlam::PredictFramePose(SE3f* estimate_1, SE3f* estimate_2) const {
  const std::vector<SE3f>& poses = base_kf_tr_frame_;
  if (poses.empty()) throw std::logic_error("PredictFramePose without a motion model entry");
  const size_t n = poses.size();
  const SE3f& last = poses[n - 1];
  *estimate_1 = last;
  if (config_.use_motion_model && n >= 2) {
    const SE3f step = poses[n - 2].Inverse() * last;        // previous frame -> last frame
    *estimate_1 = last * step;
  }
  *estimate_2 = *estimate_1;
  if (config_.use_motion_model && n >= 3) {
    const SE3f older_step = poses[n - 3].Inverse() * poses[n - 2];
    *estimate_2 = poses[n - 2] * older_step * older_step;
  }
}

void BadSlam::RunOdometry(int frame_index) {
  SE3f estimate_1, estimate_2;
  PredictFramePose(&estimate_1, &estimate_2);
  const bslam_depth_params dp = direct_ba_->depth_params();
  SE3f base_T_frame_estimate;
  // TrackFramePairwise prepares its own inputs (brightness of both colour images, depth calibration, pyramids), as
  // BadSlam::RunOdometry does before its call (:840-897)
  TrackFramePairwise(direct_ba_->context(), stream_, pairwise_tracking_buffers_.get(), direct_ba_->color_camera(), direct_ba_->depth_camera(), dp,
                     direct_ba_->use_depth_residuals(), direct_ba_->use_descriptor_residuals(), *filtered_depth_buffer_A_, *normals_buffer_, *color_buffer_,
                     base_kf_->depth_buffer(), base_kf_->normals_buffer(), base_kf_->color_buffer(), /*test_different_initial_estimates*/ true, estimate_1,
                     estimate_2, &base_T_frame_estimate, nullptr);
  FramePose(frame_index) = base_kf_global_T_frame_ * base_T_frame_estimate;
  last_frame_index_ = frame_index;
  // the history holds at most three frames
  constexpr size_t kMotionModelFrames = 3;
  base_kf_tr_frame_.push_back(base_T_frame_estimate);
  if (base_kf_tr_frame_.size() > kMotionModelFrames) base_kf_tr_frame_.erase(base_kf_tr_frame_.begin(), base_kf_tr_frame_.end() - kMotionModelFrames);
}

std::shared_ptr<Keyframe> BadSlam::CreateKeyframe(int frame_index) {
  float min_depth = 0, max_depth = 0;
  const bslam_buffer2d depth_pod = filtered_depth_buffer_A_->ToPod();
  CheckRc(bslam_compute_min_max_depth(direct_ba_->context(), stream_, &depth_pod, config_.raw_to_float_depth, &min_depth, &max_depth),
          "bslam_compute_min_max_depth");
  auto new_keyframe = std::make_shared<Keyframe>(stream_, static_cast<u32>(frame_index), min_depth, max_depth, *filtered_depth_buffer_A_, *normals_buffer_,
                                                 *radius_buffer_, *color_buffer_, FramePose(frame_index));
  CheckHip(hipStreamSynchronize(stream_), "hipStreamSynchronize");   // the frame buffers are reused by the next frame
  base_kf_ = new_keyframe.get();
  base_kf_global_T_frame_ = base_kf_->global_T_frame();
  direct_ba_->AddKeyframe(new_keyframe);   // AddKeyframeToBA without a loop detector (:1120-1158)
  const int keyframes_added = static_cast<int>(direct_ba_->keyframes().size());

  // The new keyframe is the frame the history ends with: from now on poses are expressed relative to it
  // (what BS/bad_slam.cc:1054-1066 does to its two arrays).
  if (base_kf_tr_frame_.empty()) {
    base_kf_tr_frame_.push_back(SE3f());
  } else {
    const SE3f new_base_T_old_base = base_kf_tr_frame_.back().Inverse();
    for (SE3f& pose : base_kf_tr_frame_) pose = new_base_T_old_base * pose;
    base_kf_tr_frame_.back() = SE3f();   // exactly the identity, not a product that rounds to it
  }
  if (!config_.estimate_poses) return new_keyframe;

  if (keyframes_added >= 2) {   // :1074-1094
    if (!config_.do_surfel_updates) direct_ba_->CreateSurfelsForKeyframe(stream_, true, new_keyframe);
    num_planned_ba_iterations_ += config_.max_num_ba_iterations_per_keyframe;
  } else {
    direct_ba_->CreateSurfelsForKeyframe(stream_, false, new_keyframe);
    CheckHip(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  }
  return new_keyframe;
}

void BadSlam::RunBundleAdjustment(u32 frame_index, bool optimize_depth_intrinsics, bool optimize_color_intrinsics, bool optimize_poses,
                                  bool optimize_geometry, int min_iterations, int max_iterations, int active_keyframe_window_start,
                                  int active_keyframe_window_end, bool increase_ba_iteration_count, int* iterations_done, bool* converged) {
  std::vector<SE3f> original_keyframe_T_global;
  RememberKeyframePoses(*direct_ba_, &original_keyframe_T_global);
  direct_ba_->BundleAdjustment(stream_, optimize_depth_intrinsics, optimize_color_intrinsics, config_.do_surfel_updates, optimize_poses, optimize_geometry,
                               min_iterations, max_iterations, config_.use_pcg, active_keyframe_window_start, active_keyframe_window_end,
                               increase_ba_iteration_count, iterations_done, converged);
  // In the reference a keyframe's pose IS its video frame's pose (BS/keyframe.h:160-180); here the two are separate
  // stores, so the frames that are keyframes take over the optimised poses first.
  for (const auto& kf : direct_ba_->keyframes()) {
    if (!kf) continue;
    const size_t i = static_cast<size_t>(static_cast<int>(kf->frame_index()) - config_.start_frame);
    if (i < frame_global_T_frame_.size()) frame_global_T_frame_[i] = kf->global_T_frame();
  }
  ExtrapolateAndInterpolateKeyframePoseChanges(static_cast<u32>(config_.start_frame), frame_index, *direct_ba_, original_keyframe_T_global,
                                               &frame_global_T_frame_);
  if (base_kf_) base_kf_global_T_frame_ = base_kf_->global_T_frame();
}

void BadSlam::ProcessFrame(int frame_index, const u16* depth_image, const u8* rgb_image, bool force_keyframe) {
  const int expected = config_.start_frame + static_cast<int>(frame_global_T_frame_.size());
  if (frame_index != expected) throw std::invalid_argument("ProcessFrame: expected frame index " + std::to_string(expected));
  // a new frame starts at the pose of the previous one until the odometry has run (the reference's video frames start
  // at identity and are only read after RunOdometry has set them)
  frame_global_T_frame_.push_back(frame_global_T_frame_.empty() ? SE3f() : frame_global_T_frame_.back());

  PreprocessFrame(depth_image, rgb_image);

  pose_estimated_ = false;
  if (config_.estimate_poses && base_kf_) {
    RunOdometry(frame_index);
    pose_estimated_ = true;
  }

  const bool create_keyframe = force_keyframe || ((frame_index - config_.start_frame) % config_.keyframe_interval == 0);
  if (create_keyframe) CreateKeyframe(frame_index);
  keyframe_created_ = create_keyframe;

  if (num_planned_ba_iterations_ > 0) {   // offline, sequential mode of :212-282
    ++bundle_adjustment_counter_;
    const size_t keyframes_size = direct_ba_->keyframes().size();
    const bool optimize_depth_intrinsics =
        config_.optimize_intrinsics &&
        (keyframes_size >= 10 && (keyframes_size <= 20 || (bundle_adjustment_counter_ % config_.intrinsics_optimization_interval == 0)));
    const bool optimize_color_intrinsics = optimize_depth_intrinsics;
    int iterations_done = 0;
    bool converged = false;
    RunBundleAdjustment(static_cast<u32>(frame_index), optimize_depth_intrinsics && config_.use_geometric_residuals,
                        optimize_color_intrinsics && config_.use_photometric_residuals, /*optimize_poses*/ true, /*optimize_geometry*/ true,
                        /*min_iterations*/ 0, num_planned_ba_iterations_, config_.disable_deactivation ? 0 : -1,
                        config_.disable_deactivation ? static_cast<int>(direct_ba_->keyframes().size()) - 1 : -1,
                        /*increase_ba_iteration_count*/ true, &iterations_done, &converged);
    num_planned_ba_iterations_ = converged ? 0 : std::max(0, num_planned_ba_iterations_ - iterations_done);
  }
}

}  // namespace bslam_host
