// direct_ba.hpp -- C++ host side of the bundle-adjustment hot path on MI355X.
//
// Mirrors the reference's class DirectBA (BS/direct_ba.h:65-550, BS = applications/badslam/src/
// badslam of pangfumin/badslam) for the part of its interface that lies on the hot path:
// constructor argument list, AddKeyframe, EstimateFramePose, BundleAdjustment (alternating and
// PCG), the accessors callers use.  All device work goes through the C ABI of
// include/badslam_hip.h; this class owns the scene state exactly as the reference's does
// (surfel SoA, active mask, cfactor image, PCG vectors, keyframe buffers).
//
// Not on the hot path and therefore absent ("out of scope" / "next" rows): surfel
// creation / merge / deletion / compaction (do_surfel_updates must be false, surfels are
// uploaded with SetSurfels), keyframe merging, visualisation.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/badslam_hip.h"
#include "se3.hpp"

namespace bslam_host {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;

// = vis::PinholeCamera4f (libvis/src/libvis/camera.h:1740-1743): fx, fy, cx, cy (pixel-corner
// convention) + image size.
class PinholeCamera4f {
 public:
  PinholeCamera4f() : width_(0), height_(0), p_{0, 0, 0, 0} {}
  PinholeCamera4f(int width, int height, const float* parameters) : width_(width), height_(height) {
    for (int i = 0; i < 4; ++i) p_[i] = parameters[i];
  }
  int width() const { return width_; }
  int height() const { return height_; }
  const float* parameters() const { return p_; }
  bslam_camera4f pod() const { return bslam_camera4f{p_[0], p_[1], p_[2], p_[3], width_, height_}; }

 private:
  int width_, height_;
  float p_[4];
};

// Device image with the layout of vis::CUDABuffer<T> (pitched 2-D).
template <typename T>
class DeviceBuffer {
 public:
  DeviceBuffer(int height, int width);
  ~DeviceBuffer();
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  int width() const { return width_; }
  int height() const { return height_; }
  size_t pitch() const { return pitch_; }
  T* address() const { return data_; }
  bslam_buffer2d ToPod() const { return bslam_buffer2d{data_, height_, width_, pitch_}; }
  void Upload(hipStream_t stream, const T* host, size_t host_pitch_bytes);
  void Download(hipStream_t stream, T* host, size_t host_pitch_bytes) const;
  void Clear(int byte_value, hipStream_t stream);

 private:
  T* data_;
  int height_, width_;
  size_t pitch_;
};

struct uchar4_t { u8 x, y, z, w; };

// = vis::Keyframe (BS/keyframe.h) reduced to what the path reads.
class Keyframe {
 public:
  enum class Activation { kActive = 0, kCovisibleActive = 1, kInactive = 2 };   // BS/keyframe.h:54-67

  // Mirrors the reference constructor that takes prepared buffers (BS/keyframe.cc:35-80); the
  // images are HOST arrays in the formats of SURVEY.md A.2 and are uploaded.
  Keyframe(hipStream_t stream, u32 frame_index, float min_depth, float max_depth, int width, int height,
           const u16* depth, const u16* normals, const u16* radius, const uchar4_t* color, const SE3f& global_T_frame);

  // Mirrors the reference constructor from raw images (BS/keyframe.cc:82-161): uploads the u16 depth and the
  // 3-byte rgb image (HOST arrays) and runs ComputeBrightness, ComputeNormals, ComputePointRadiiAndRemoveIsolatedPixels
  // and ComputeMinMaxDepth through the C ABI of `ctx`.
  Keyframe(bslam_context* ctx, hipStream_t stream, u32 frame_index, const bslam_depth_params& depth_params, const bslam_camera4f& depth_camera,
           const u16* depth_image, int color_width, int color_height, const u8* rgb_image, const SE3f& global_T_frame);

  // The reference's prepared-buffer constructor proper (BS/keyframe.cc:35-80): DEVICE images, copied on `stream`.
  Keyframe(hipStream_t stream, u32 frame_index, float min_depth, float max_depth, const DeviceBuffer<u16>& depth, const DeviceBuffer<u16>& normals,
           const DeviceBuffer<u16>& radius, const DeviceBuffer<uchar4_t>& color, const SE3f& global_T_frame);

  const DeviceBuffer<u16>& depth_buffer() const { return depth_; }
  const DeviceBuffer<u16>& normals_buffer() const { return normals_; }
  const DeviceBuffer<u16>& radius_buffer() const { return radius_; }
  const DeviceBuffer<uchar4_t>& color_buffer() const { return color_; }
  DeviceBuffer<u16>& mutable_depth_buffer() { return depth_; }
  DeviceBuffer<u16>& mutable_normals_buffer() { return normals_; }

  const SE3f& global_T_frame() const { return global_T_frame_; }
  const SE3f& frame_T_global() const { return frame_T_global_; }
  void set_global_T_frame(const SE3f& T) { global_T_frame_ = T; frame_T_global_ = T.Inverse(); }   // BS/keyframe.h:160-173

  Activation activation() const { return activation_; }
  void SetActivation(Activation a) { activation_ = a; }
  int id() const { return id_; }
  void SetID(int id) { id_ = id; }
  u32 frame_index() const { return frame_index_; }
  float min_depth() const { return min_depth_; }
  float max_depth() const { return max_depth_; }
  std::vector<int>& co_visibility_list() { return co_visibility_list_; }
  int last_active_in_ba_iteration() const { return last_active_in_ba_iteration_; }
  void SetLastActiveInBAIteration(int v) { last_active_in_ba_iteration_ = v; }
  int last_covis_in_ba_iteration() const { return last_covis_in_ba_iteration_; }
  void SetLastCovisInBAIteration(int v) { last_covis_in_ba_iteration_ = v; }

  bslam_keyframe_view view() const;

 private:
  u32 frame_index_;
  int id_ = -1;
  int last_active_in_ba_iteration_ = -1, last_covis_in_ba_iteration_ = -1;
  float min_depth_, max_depth_;
  DeviceBuffer<u16> depth_, normals_, radius_;
  DeviceBuffer<uchar4_t> color_;
  SE3f global_T_frame_, frame_T_global_;
  Activation activation_ = Activation::kActive;
  std::vector<int> co_visibility_list_;
};

// = vis::CameraFrustum (libvis/src/libvis/camera_frustum.h): bounding-box test, then the
// separating-axis test on planes and edge cross products.
class CameraFrustum {
 public:
  CameraFrustum(const PinholeCamera4f& camera, float min_depth, float max_depth, const SE3f& global_T_camera);
  bool Intersects(CameraFrustum* other);

 private:
  void ComputeAxesAndPlanes();
  Vec3f points_[8];
  Vec3f axes_[6];
  Vec3f plane_n_[6];
  float plane_d_[6];
  bool computed_ = false;
  float bb_min_[3], bb_max_[3];
};

class Timer;   // reference API placeholder (BS/direct_ba.h:158); only a null pointer is accepted

class DirectBA {
 public:
  // Argument list of BS/direct_ba.h:73-88 (render_window must be null: no display on the GPU box).
  DirectBA(int max_surfel_count, float raw_to_float_depth, float baseline_fx, int sparse_surfel_cell_size,
           float surfel_merge_dist_factor, int min_observation_count_while_bootstrapping_1,
           int min_observation_count_while_bootstrapping_2, int min_observation_count,
           const PinholeCamera4f& color_camera_initial_estimate, const PinholeCamera4f& depth_camera_initial_estimate,
           int pyramid_level_for_color, bool use_depth_residuals, bool use_descriptor_residuals,
           void* render_window, const SE3f& global_T_anchor_frame, int device = 0);
  ~DirectBA();

  void AddKeyframe(const std::shared_ptr<Keyframe>& new_keyframe);   // BS/direct_ba.cc:196-204

  // BS/direct_ba_alternating.cc:42-283: Gauss-Newton on one frame against the surfel model, host
  // loop with one accumulation call + double LDLT + SE3 update per iteration, as the reference.
  void EstimateFramePose(hipStream_t stream, const SE3f& global_T_frame_initial_estimate, const DeviceBuffer<u16>& depth_buffer,
                         const DeviceBuffer<u16>& normals_buffer, const DeviceBuffer<uchar4_t>& color_buffer,
                         SE3f* out_global_T_frame_estimate, bool called_within_ba);

  // BS/direct_ba.h:143-162 / BS/direct_ba.cc:407-453
  void BundleAdjustment(hipStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics, bool do_surfel_updates,
                        bool optimize_poses, bool optimize_geometry, int min_iterations, int max_iterations, bool use_pcg,
                        int active_keyframe_window_start, int active_keyframe_window_end, bool increase_ba_iteration_count,
                        int* iterations_done = nullptr, bool* converged = nullptr, double time_limit = 0, Timer* timer = nullptr,
                        int pcg_max_inner_iterations = 30, int pcg_max_keyframes = 2500,
                        std::function<bool(int)> progress_function = nullptr);

  // Keyframe management of the outer seam (BS/direct_ba.h:95-116).  The reference's LoopDetector* parameters are
  // dropped: loop detection stays with the caller, which removes the image from its own detector.
  void DeleteKeyframe(int keyframe_index);                                    // BS/direct_ba.cc:207-229
  // Deletes up to approx_merge_count keyframes that are close to both neighbours on the trajectory (never keyframe 0);
  // returns the deleted ids in the order of deletion.                         // BS/direct_ba.cc:251-338
  std::vector<int> MergeKeyframes(hipStream_t stream, size_t approx_merge_count);
  void UpdateKeyframeCoVisibility(const std::shared_ptr<Keyframe>& keyframe);  // BS/direct_ba.cc:710-737
  void AssignColors(hipStream_t stream);                                       // BS/direct_ba.cc:456-459
  // ExportToPointCloud (BS/direct_ba.cc:461-546): the valid (non-NaN) surfels as host arrays; normals are
  // re-normalised 10-bit normals as in the reference.
  struct PointCloud {
    std::vector<float> positions;   // 3 per point
    std::vector<u8> colors;         // 3 per point (r, g, b)
    std::vector<float> normals;     // 3 per point
    size_t size() const { return positions.size() / 3; }
  };
  void ExportToPointCloud(hipStream_t stream, PointCloud* cloud) const;

  // Scene state in / out.
  void SetSurfels(hipStream_t stream, const float* host_rows, size_t host_pitch_bytes, u32 count);
  void GetSurfels(hipStream_t stream, float* host_rows, size_t host_pitch_bytes, int rows) const;
  void GetActiveSurfels(hipStream_t stream, u8* host) const;

  // Pose phase of the alternating scheme: true (default) = all keyframes advance in lock-step in
  // one launch per Gauss-Newton iteration (bslam_estimate_frame_poses_batched); false = the
  // reference's sequential EstimateFramePose calls.
  void SetBatchedPoseOptimization(bool enable) { batched_pose_optimization_ = enable; }
  // PerformBASchemeEndTasks (merge / delete / radius update / compaction at the end of a BA scheme) is on as in the
  // reference; switching it off keeps the surfel set fixed across BundleAdjustment calls (parity tests against plain loops).
  void SetSchemeEndTasks(bool enable) { scheme_end_tasks_ = enable; }
  // DirectBA::CreateSurfelsForKeyframe (BS/direct_ba.h:118-121)
  void CreateSurfelsForKeyframe(hipStream_t stream, bool filter_new_surfels, const std::shared_ptr<Keyframe>& keyframe);
  // new Keyframe(stream, frame_index, depth_params(), depth_camera(), depth_image, color_image, pose) + AddKeyframe
  // (the idiom of the reference's tests, e.g. BS/test/test_pose_optimization_geometric_residual.cc:108-118)
  std::shared_ptr<Keyframe> AddKeyframeFromImages(hipStream_t stream, u32 frame_index, const u16* depth_image, const u8* rgb_image, const SE3f& global_T_frame);
  // The reference picks the PCG gauge keyframe with rand() % K (BS/direct_ba_pcg.cc:328); a fixed
  // id >= 0 makes runs reproducible.
  void SetPCGGaugeKeyframe(int id) { fixed_gauge_keyframe_ = id; }
  void SetTextureMode(int mode);
  // Must be called after rewriting the content of a keyframe's depth / normal image or of the
  // cfactor image in place (the class does it itself for the changes it makes).
  void InvalidateKeyframeCache();
  void SetTimingsStream(std::ostream* s) { timings_stream_ = s; }   // --save_timings format of BS/direct_ba_alternating.cc:630-688
  // Surfel-sharded multi-GPU runs: `fn` sums device buffers across ranks (RCCL / torch.distributed); it is
  // used by the batched pose step and, through the context, by the PCG and intrinsics entry points.
  void SetAllReduce(bslam_allreduce_fn fn, void* user);
  // The library's own exchange: an RCCL communicator in this DirectBA's kernel context (bslam_comm_init).  rank 0 obtains the
  // 128-byte id with bslam_comm_get_unique_id and hands it to every rank; after InitComm every BA scheme of this object
  // (alternating pose step, PCG, intrinsics) sums its shared quantities over the ranks on the BA stream -- no callback.
  void InitComm(const void* unique_id, int rank, int world_size);
  void DestroyComm();

  void Lock() const { ba_thread_mutex_.lock(); }
  void Unlock() const { ba_thread_mutex_.unlock(); }
  const std::vector<std::shared_ptr<Keyframe>>& keyframes() const { return keyframes_; }
  const PinholeCamera4f& depth_camera() const { return depth_camera_; }
  const PinholeCamera4f& color_camera() const { return color_camera_; }
  bslam_depth_params depth_params() const;
  float a() const { return a_; }
  void SetA(float a) { a_ = a; }
  void SetDepthCamera(const PinholeCamera4f& camera) { depth_camera_ = camera; }   // BS/direct_ba.h (used by the intrinsics tests)
  void SetColorCamera(const PinholeCamera4f& camera) { color_camera_ = camera; }
  const DeviceBuffer<float>& cfactor_buffer() const { return *cfactor_buffer_; }
  // host row-major cfactor image -> device (LoadCalibration, BS/io.cc:694); invalidates the derived records
  void UploadCFactor(hipStream_t stream, const float* host) {
    cfactor_buffer_->Upload(stream, host, static_cast<size_t>(cfactor_buffer_->width()) * sizeof(float));
    InvalidateKeyframeCache();
  }
  u32 surfels_size() const { return surfels_size_; }
  u32 surfel_count() const { return surfel_count_; }
  const DeviceBuffer<float>& surfels() const { return *surfels_; }
  bool use_depth_residuals() const { return use_depth_residuals_; }
  bool use_descriptor_residuals() const { return use_descriptor_residuals_; }
  int ba_iteration_count() const { return ba_iteration_count_; }
  int last_ba_iteration_count() const { return last_ba_iteration_count_; }
  void SetBAIterationCounts(int count, int last) { ba_iteration_count_ = count; last_ba_iteration_count_ = last; }   // LoadState, BS/io.cc:466-467
  int max_surfel_count() const { return surfels_->width(); }
  int min_observation_count_while_bootstrapping_1() const { return min_observation_count_while_bootstrapping_1_; }
  int min_observation_count_while_bootstrapping_2() const { return min_observation_count_while_bootstrapping_2_; }
  int min_observation_count() const { return min_observation_count_; }
  float surfel_merge_dist_factor() const { return surfel_merge_dist_factor_; }
  int GetMinObservationCount() const {   // BS/direct_ba.h:212-218
    return (keyframes_.size() < 10) ? ((keyframes_.size() < 5) ? min_observation_count_while_bootstrapping_1_
                                                              : min_observation_count_while_bootstrapping_2_)
                                    : min_observation_count_;
  }
  bslam_context* context() const { return ctx_; }

 private:
  void BundleAdjustmentAlternating(hipStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics, bool do_surfel_updates,
                                   bool optimize_poses, bool optimize_geometry, int min_iterations, int max_iterations,
                                   int active_keyframe_window_start, int active_keyframe_window_end, bool increase_ba_iteration_count,
                                   int* num_iterations_done, bool* converged, double time_limit, std::function<bool(int)> progress_function);
  void BundleAdjustmentPCG(hipStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics, bool do_surfel_updates,
                           bool optimize_poses, bool optimize_geometry, int min_iterations, int max_iterations, int max_inner_iterations,
                           int max_keyframe_count, int active_keyframe_window_start, int active_keyframe_window_end,
                           bool increase_ba_iteration_count, int* num_iterations_done, bool* converged, double time_limit,
                           std::function<bool(int)> progress_function);
  void DetermineNewKeyframeCoVisibility(const std::shared_ptr<Keyframe>& new_keyframe);   // BS/direct_ba.cc:231-249
  void DetermineCovisibleActiveKeyframes();                                               // BS/direct_ba.cc:548-564
  std::vector<bslam_keyframe_view> KeyframeViews() const;
  // Merge for the keyframes last active in this BA iteration, DeleteSurfelsAndUpdateRadii, Compact (BS/direct_ba.cc:566-653)
  void PerformBASchemeEndTasks(hipStream_t stream, bool do_surfel_updates);
  // DetermineSupportingSurfelsAndMergeSurfelsCUDA for the given keyframes, then CompactSurfelsCUDA incl. the active flags
  // (BS/direct_ba_alternating.cc:486-533)
  void MergeAndCompact(hipStream_t stream, const std::vector<u32>& keyframe_ids);
  void Check(int rc, const char* what) const;

  bslam_context* ctx_ = nullptr;
  int device_;
  PinholeCamera4f color_camera_, depth_camera_;
  int pyramid_level_for_color_;
  bool use_depth_residuals_, use_descriptor_residuals_;
  int min_observation_count_while_bootstrapping_1_, min_observation_count_while_bootstrapping_2_, min_observation_count_;
  float surfel_merge_dist_factor_;
  SE3f global_T_anchor_frame_;
  float a_ = 0.f, raw_to_float_depth_, baseline_fx_;
  int sparse_surfel_cell_size_;
  std::unique_ptr<DeviceBuffer<float>> cfactor_buffer_;
  std::unique_ptr<DeviceBuffer<float>> surfels_;
  std::unique_ptr<DeviceBuffer<u8>> active_surfels_;
  u32 surfels_size_ = 0, surfel_count_ = 0;
  int ba_iteration_count_ = 0, last_ba_iteration_count_ = -1;
  std::vector<std::shared_ptr<Keyframe>> keyframes_;
  mutable std::mutex ba_thread_mutex_;
  // PCG vectors, allocated lazily (BS/direct_ba_pcg.cc:255-268)
  std::unique_ptr<DeviceBuffer<float>> pcg_r_, pcg_M_, pcg_delta_, pcg_g_, pcg_p_, pcg_scalars_;
  bool batched_pose_optimization_ = true;
  bool scheme_end_tasks_ = true;
  int fixed_gauge_keyframe_ = -1;
  std::ostream* timings_stream_ = nullptr;
  bool comm_ = false, sharded_ = false;
  bslam_allreduce_fn allreduce_ = nullptr;
  void* allreduce_user_ = nullptr;
  // phase timing (BS/direct_ba.h:513-532)
  // phase timing events of a BA iteration (--save_timings lines): [0,1] activation, [2,3] geometry, [4,5] poses, [6,7] PCG,
  // [8,9] surfel creation, [10,11] initial surfel merge, [12,13] surfel compaction, [14,15] intrinsics
  hipEvent_t ev_[16];
};

}  // namespace bslam_host
