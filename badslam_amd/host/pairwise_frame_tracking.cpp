// pairwise_frame_tracking.cpp -- see pairwise_frame_tracking.hpp.
#include "pairwise_frame_tracking.hpp"

#include <cmath>
#include <stdexcept>
#include <string>

namespace bslam_host {

namespace {
void Check(int rc, const char* what) {
  if (rc != BSLAM_OK) throw std::runtime_error(std::string(what) + ": " + bslam_last_error());
}
// CameraImpl::Scaled (LV/camera.h:1696-1705) with PixelMapping4::ScaleParameters (:1086-1096)
bslam_camera4f Scaled(const PinholeCamera4f& c, float factor) {
  const float* p = c.parameters();
  bslam_camera4f s;
  s.fx = p[0] * factor; s.fy = p[1] * factor; s.cx = p[2] * factor; s.cy = p[3] * factor;
  s.width = static_cast<int>(factor * c.width() + 0.5f);
  s.height = static_cast<int>(factor * c.height() + 0.5f);
  return s;
}
// IsScaleNPoseEstimationConverged (BS/convergence_analysis.h:56-63)
bool IsScaleNPoseEstimationConverged(const float* x, float scaling_factor) {
  constexpr float translation_threshold = 1e-08f, rotation_threshold = 1e-08f;
  float n = 0;
  for (int i = 0; i < 6; ++i) { const float v = (i < 3) ? x[i] : x[i] * (translation_threshold / rotation_threshold); n += v * v; }
  return n < scaling_factor * scaling_factor * translation_threshold;
}
}  // namespace

PairwiseFrameTrackingBuffers::PairwiseFrameTrackingBuffers(int depth_width, int depth_height, int color_width, int color_height, int num_scales_)
    : num_scales(num_scales_), base_depth(num_scales_), tracked_depth(num_scales_), base_normals(num_scales_), tracked_normals(num_scales_),
      base_color(num_scales_), tracked_color(num_scales_) {
  for (int scale = 0; scale < num_scales; ++scale) {   // BS/pairwise_frame_tracking.cc:38-80
    const int w = static_cast<int>(depth_width / std::pow(2, scale)), h = static_cast<int>(depth_height / std::pow(2, scale));
    base_depth[scale].reset(new DeviceBuffer<float>(h, w));
    tracked_depth[scale].reset(new DeviceBuffer<float>(h, w));
    base_color[scale].reset(new DeviceBuffer<u8>(h, w));
    tracked_color[scale].reset(new DeviceBuffer<u8>(h, w));
    if (scale >= 1) {
      base_normals[scale].reset(new DeviceBuffer<u16>(h, w));
      tracked_normals[scale].reset(new DeviceBuffer<u16>(h, w));
    }
  }
  base_gradmag.reset(new DeviceBuffer<u8>(color_height, color_width));
  tracked_gradmag.reset(new DeviceBuffer<u8>(color_height, color_width));
}

void TrackFramePairwise(bslam_context* ctx, hipStream_t stream, PairwiseFrameTrackingBuffers* buffers, const PinholeCamera4f& color_camera,
                        const PinholeCamera4f& depth_camera, const bslam_depth_params& dp, bool use_depth_residuals, bool use_descriptor_residuals,
                        const DeviceBuffer<u16>& tracked_depth_u16, const DeviceBuffer<u16>& tracked_normals_l0, const DeviceBuffer<uchar4_t>& tracked_color_rgba,
                        const DeviceBuffer<u16>& base_depth_u16, const DeviceBuffer<u16>& base_normals_l0, const DeviceBuffer<uchar4_t>& base_color_rgba,
                        bool test_different_initial_estimates, const SE3f& init1, const SE3f& init2, SE3f* out_base_T_frame, int* iterations_per_scale,
                        bool use_pyramid_level_0, bool use_gradmag) {
  if (depth_camera.width() != color_camera.width()) throw std::invalid_argument("TrackFramePairwise: depth and colour images must have the same size here");
  const int num_scales = buffers->num_scales;
  const bslam_camera4f color_cam = color_camera.pod(), depth_cam = depth_camera.pod();
  std::vector<bslam_buffer2d> base_depth(num_scales), base_normals(num_scales), base_color(num_scales), tracked_depth(num_scales),
      tracked_normals(num_scales), tracked_color(num_scales);
  for (int s = 0; s < num_scales; ++s) {
    base_depth[s] = buffers->base_depth[s]->ToPod();
    base_color[s] = buffers->base_color[s]->ToPod();
    tracked_depth[s] = buffers->tracked_depth[s]->ToPod();
    tracked_color[s] = buffers->tracked_color[s]->ToPod();
    base_normals[s] = s ? buffers->base_normals[s]->ToPod() : base_normals_l0.ToPod();
    tracked_normals[s] = s ? buffers->tracked_normals[s]->ToPod() : tracked_normals_l0.ToPod();
  }
  // --- input preparation (BadSlam::RunOdometry, BS/bad_slam.cc:859-897)
  const bslam_buffer2d base_rgba = base_color_rgba.ToPod(), tracked_rgba = tracked_color_rgba.ToPod();
  const bslam_buffer2d base_gm = buffers->base_gradmag->ToPod(), tracked_gm = buffers->tracked_gradmag->ToPod();
  const bslam_buffer2d base_d16 = base_depth_u16.ToPod(), tracked_d16 = tracked_depth_u16.ToPod();
  // gradient-magnitude images instead of brightness images when use_gradmag (:859-869, 888-898)
  auto colour_cue = [&](const bslam_buffer2d* rgba, const bslam_buffer2d* out) {
    if (use_gradmag) Check(bslam_compute_sobel_gradient_magnitude(ctx, stream, rgba, out), "bslam_compute_sobel_gradient_magnitude");
    else Check(bslam_compute_brightness_from_color(ctx, stream, rgba, out), "bslam_compute_brightness_from_color");
  };
  colour_cue(&base_rgba, &base_gm);
  Check(bslam_calibrate_depth_and_transform_color_to_depth(ctx, stream, &color_cam, &depth_cam, &dp, &base_d16, &base_gm, &base_depth[0], &base_color[0]),
        "bslam_calibrate_depth_and_transform_color_to_depth");
  colour_cue(&tracked_rgba, &tracked_gm);
  // --- pyramids (BS/pairwise_frame_tracking.cc:283-341)
  const bslam_buffer2d tracked_normals_in = tracked_normals_l0.ToPod();
  if (use_pyramid_level_0) {
    Check(bslam_calibrate_depth(ctx, stream, &dp, &tracked_d16, &tracked_depth[0]), "bslam_calibrate_depth");
    Check(bslam_set_to_read_mode_normalized(ctx, stream, &tracked_gm, &tracked_color[0]), "bslam_set_to_read_mode_normalized");
  } else if (num_scales > 1) {   // :303-323
    Check(bslam_calibrate_and_downsample_images(ctx, stream, depth_camera.width() == color_camera.width(), &dp, &tracked_d16, &tracked_normals_in, &tracked_gm,
                                                &tracked_depth[1], &tracked_normals[1], &tracked_color[1]),
          "bslam_calibrate_and_downsample_images");
  }
  for (int s = 1; s < num_scales; ++s) {
    if (s >= 2 || use_pyramid_level_0)
      Check(bslam_downsample_images(ctx, stream, &tracked_depth[s - 1], &tracked_normals[s - 1], &tracked_color[s - 1], &tracked_depth[s], &tracked_normals[s],
                                    &tracked_color[s]),
            "bslam_downsample_images");
    Check(bslam_downsample_images(ctx, stream, &base_depth[s - 1], &base_normals[s - 1], &base_color[s - 1], &base_depth[s], &base_normals[s], &base_color[s]),
          "bslam_downsample_images");
  }
  // --- coarse to fine (:343-640)
  constexpr int kMaxIterationsPerScale = 30;
  SE3f estimate = init1, chosen_initial = init1;
  for (int scale = num_scales - 1; scale >= (use_pyramid_level_0 ? 0 : 1); --scale) {   // :367
    const float scaling_factor = static_cast<float>(std::pow(2, scale));
    const bslam_camera4f tcc = Scaled(color_camera, 1.f / scaling_factor), tdc = Scaled(depth_camera, 1.f / scaling_factor);
    const float threshold_factor = scaling_factor;
    auto cost_of = [&](const SE3f& base_T_frame, u32* count, float* cost) {
      const bslam_mat3x4 M = base_T_frame.Inverse().Matrix3x4();
      Check((use_gradmag ? bslam_compute_cost_and_residual_count_from_images_gradmag : bslam_compute_cost_and_residual_count_from_images)(
                ctx, stream, use_depth_residuals, use_descriptor_residuals, &tcc, &tdc, dp.baseline_fx, threshold_factor, &tracked_depth[scale],
                &tracked_normals[scale], &tracked_color[scale], &M, &base_depth[scale], &base_normals[scale], &base_color[scale], count, cost),
            "bslam_compute_cost_and_residual_count_from_images");
    };
    if (scale != num_scales - 1 || test_different_initial_estimates) {   // :428-489
      const SE3f last = (scale != num_scales - 1) ? estimate : init1;
      const SE3f other = (scale != num_scales - 1) ? chosen_initial : init2;
      u32 count_last = 0, count_other = 0;
      float cost_last = 0, cost_other = 0;
      cost_of(last, &count_last, &cost_last);
      cost_of(other, &count_other, &cost_other);
      if (count_last > 2 * count_other) estimate = last;
      else if (count_other > 2 * count_last) estimate = other;
      else if (cost_last < cost_other) estimate = last;
      else estimate = other;
      if (scale == num_scales - 1) chosen_initial = estimate;
    }
    int iteration;
    for (iteration = 0; iteration < kMaxIterationsPerScale; ++iteration) {
      const bslam_mat3x4 M = estimate.Inverse().Matrix3x4();
      float H[21], b[6], x[6];
      Check((use_gradmag ? bslam_accumulate_pose_coeffs_from_images_gradmag : bslam_accumulate_pose_coeffs_from_images)(
                ctx, stream, use_depth_residuals, use_descriptor_residuals, &tcc, &tdc, dp.baseline_fx, threshold_factor, &tracked_depth[scale],
                &tracked_normals[scale], &tracked_color[scale], &M, &base_depth[scale], &base_normals[scale], &base_color[scale], nullptr, H, b),
            "bslam_accumulate_pose_coeffs_from_images");
      SolveLDLTUpper(6, H, b, x);   // :561
      float damping = 1.f;          // :573-579
      if (scale == num_scales - 2) damping = 0.5f;
      else if (scale == num_scales - 1) damping = 0.25f;
      float step[6];
      for (int i = 0; i < 6; ++i) step[i] = -damping * x[i];
      estimate = estimate * SE3f::Exp(step);
      if (IsScaleNPoseEstimationConverged(x, scaling_factor)) { ++iteration; break; }
    }
    if (iterations_per_scale) iterations_per_scale[scale] = iteration;
  }
  *out_base_T_frame = estimate;
}

}  // namespace bslam_host
