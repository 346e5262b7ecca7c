// direct_ba.cpp -- see direct_ba.hpp.  Host loops of the reference's DirectBA re-written over
// the C ABI of include/badslam_hip.h (BS = applications/badslam/src/badslam of pangfumin/badslam).
#include "direct_ba.hpp"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace bslam_host {

#define HIP_OR_THROW(expr)                                                                         \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// ------------------------------------------------------------------------------------------------
// DeviceBuffer
// ------------------------------------------------------------------------------------------------
template <typename T>
DeviceBuffer<T>::DeviceBuffer(int height, int width) : data_(nullptr), height_(height), width_(width) {
  // pitched like cudaMallocPitch (libvis/src/libvis/cuda/cuda_buffer_inl.h:39): rows padded to 256 bytes
  pitch_ = ((static_cast<size_t>(width) * sizeof(T) + 255) / 256) * 256;
  HIP_OR_THROW(hipMalloc(reinterpret_cast<void**>(&data_), pitch_ * static_cast<size_t>(std::max(1, height))));
}
template <typename T>
DeviceBuffer<T>::~DeviceBuffer() {
  if (data_) { hipError_t e = hipFree(data_); (void)e; }
}
template <typename T>
void DeviceBuffer<T>::Upload(hipStream_t stream, const T* host, size_t host_pitch_bytes) {
  HIP_OR_THROW(hipMemcpy2DAsync(data_, pitch_, host, host_pitch_bytes, static_cast<size_t>(width_) * sizeof(T), height_, hipMemcpyHostToDevice, stream));
  HIP_OR_THROW(hipStreamSynchronize(stream));   // the host array may be pageable and short-lived
}
template <typename T>
void DeviceBuffer<T>::Download(hipStream_t stream, T* host, size_t host_pitch_bytes) const {
  HIP_OR_THROW(hipMemcpy2DAsync(host, host_pitch_bytes, data_, pitch_, static_cast<size_t>(width_) * sizeof(T), height_, hipMemcpyDeviceToHost, stream));
  HIP_OR_THROW(hipStreamSynchronize(stream));
}
template <typename T>
void DeviceBuffer<T>::Clear(int byte_value, hipStream_t stream) {
  HIP_OR_THROW(hipMemsetAsync(data_, byte_value, pitch_ * static_cast<size_t>(height_), stream));
}
template class DeviceBuffer<float>;
template class DeviceBuffer<u8>;
template class DeviceBuffer<u16>;
template class DeviceBuffer<uchar4_t>;

// ------------------------------------------------------------------------------------------------
// Keyframe
// ------------------------------------------------------------------------------------------------
Keyframe::Keyframe(hipStream_t stream, u32 frame_index, float min_depth, float max_depth, int width, int height,
                   const u16* depth, const u16* normals, const u16* radius, const uchar4_t* color, const SE3f& global_T_frame)
    : frame_index_(frame_index), min_depth_(min_depth), max_depth_(max_depth),
      depth_(height, width), normals_(height, width), radius_(height, width), color_(height, width) {
  if (!(min_depth > 0.f)) throw std::invalid_argument("Keyframe min depth must be larger than 0 (BS/keyframe.cc:57-59)");
  depth_.Upload(stream, depth, static_cast<size_t>(width) * sizeof(u16));
  normals_.Upload(stream, normals, static_cast<size_t>(width) * sizeof(u16));
  radius_.Upload(stream, radius, static_cast<size_t>(width) * sizeof(u16));
  color_.Upload(stream, color, static_cast<size_t>(width) * sizeof(uchar4_t));
  set_global_T_frame(global_T_frame);
}

Keyframe::Keyframe(bslam_context* ctx, hipStream_t stream, u32 frame_index, const bslam_depth_params& depth_params, const bslam_camera4f& depth_camera,
                   const u16* depth_image, int color_width, int color_height, const u8* rgb_image, const SE3f& global_T_frame)
    : frame_index_(frame_index), min_depth_(0.f), max_depth_(0.f),
      depth_(depth_camera.height, depth_camera.width), normals_(depth_camera.height, depth_camera.width),
      radius_(depth_camera.height, depth_camera.width), color_(color_height, color_width) {
  const int w = depth_camera.width, h = depth_camera.height;
  auto check = [](int rc, const char* what) { if (rc != BSLAM_OK) throw std::runtime_error(std::string(what) + ": " + bslam_last_error()); };
  DeviceBuffer<u8> rgb(color_height, color_width * 3);
  rgb.Upload(stream, rgb_image, static_cast<size_t>(color_width) * 3);
  bslam_buffer2d rgb_pod = rgb.ToPod();
  rgb_pod.width = color_width;   // 3 bytes per pixel
  const bslam_buffer2d color_pod = color_.ToPod();
  check(bslam_compute_brightness(ctx, stream, &rgb_pod, &color_pod), "bslam_compute_brightness");
  DeviceBuffer<u16> raw(h, w), stage1(h, w);
  raw.Upload(stream, depth_image, static_cast<size_t>(w) * sizeof(u16));
  const bslam_buffer2d raw_pod = raw.ToPod(), stage1_pod = stage1.ToPod(), normals_pod = normals_.ToPod(), radius_pod = radius_.ToPod(),
                       depth_pod = depth_.ToPod();
  check(bslam_compute_normals(ctx, stream, &depth_camera, &depth_params, &raw_pod, &stage1_pod, &normals_pod), "bslam_compute_normals");
  check(bslam_compute_point_radii_and_remove_isolated_pixels(ctx, stream, &depth_camera, depth_params.raw_to_float_depth, &stage1_pod, &radius_pod,
                                                             &depth_pod),
        "bslam_compute_point_radii_and_remove_isolated_pixels");
  check(bslam_compute_min_max_depth(ctx, stream, &stage1_pod, depth_params.raw_to_float_depth, &min_depth_, &max_depth_), "bslam_compute_min_max_depth");
  HIP_OR_THROW(hipStreamSynchronize(stream));   // the temporaries die here
  set_global_T_frame(global_T_frame);
}

template <typename T>
static void CopyDeviceImage(hipStream_t stream, const DeviceBuffer<T>& from, DeviceBuffer<T>* to) {
  if (from.width() != to->width() || from.height() != to->height()) throw std::invalid_argument("Keyframe: image sizes differ");
  HIP_OR_THROW(hipMemcpy2DAsync(to->address(), to->pitch(), from.address(), from.pitch(), static_cast<size_t>(from.width()) * sizeof(T), from.height(),
                                hipMemcpyDeviceToDevice, stream));
}

Keyframe::Keyframe(hipStream_t stream, u32 frame_index, float min_depth, float max_depth, const DeviceBuffer<u16>& depth, const DeviceBuffer<u16>& normals,
                   const DeviceBuffer<u16>& radius, const DeviceBuffer<uchar4_t>& color, const SE3f& global_T_frame)
    : frame_index_(frame_index), min_depth_(min_depth), max_depth_(max_depth), depth_(depth.height(), depth.width()),
      normals_(normals.height(), normals.width()), radius_(radius.height(), radius.width()), color_(color.height(), color.width()) {
  CopyDeviceImage(stream, depth, &depth_);
  CopyDeviceImage(stream, normals, &normals_);
  CopyDeviceImage(stream, radius, &radius_);
  CopyDeviceImage(stream, color, &color_);
  set_global_T_frame(global_T_frame);
}

bslam_keyframe_view Keyframe::view() const {
  bslam_keyframe_view v;
  v.depth = depth_.ToPod();
  v.normals = normals_.ToPod();
  v.radius = radius_.ToPod();
  v.color = color_.ToPod();
  v.frame_T_global = frame_T_global_.Matrix3x4();
  v.global_R_frame = global_T_frame_.Rotation3x3();
  v.activation = static_cast<int>(activation_);
  v.id = id_;
  return v;
}

// ------------------------------------------------------------------------------------------------
// CameraFrustum (libvis/src/libvis/camera_frustum.h:40-222)
// ------------------------------------------------------------------------------------------------
static inline Vec3f sub(Vec3f a, Vec3f b) { return Vec3f{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3f crossv(Vec3f a, Vec3f b) { return Vec3f{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline float dotv(Vec3f a, Vec3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

CameraFrustum::CameraFrustum(const PinholeCamera4f& camera, float min_depth, float max_depth, const SE3f& global_T_camera) {
  const float* p = camera.parameters();
  const bslam_mat3x4 M = global_T_camera.Matrix3x4();
  auto unproject_corner = [&](int x, int y) { return Vec3f{(x - p[2]) / p[0], (y - p[3]) / p[1], 1.f}; };
  auto transform = [&](Vec3f d, float depth) {
    const Vec3f q{depth * d.x, depth * d.y, depth * d.z};
    return Vec3f{M.m[0] * q.x + M.m[1] * q.y + M.m[2] * q.z + M.m[3], M.m[4] * q.x + M.m[5] * q.y + M.m[6] * q.z + M.m[7],
                 M.m[8] * q.x + M.m[9] * q.y + M.m[10] * q.z + M.m[11]};
  };
  const Vec3f corners[4] = {unproject_corner(0, 0), unproject_corner(camera.width(), 0), unproject_corner(0, camera.height()),
                            unproject_corner(camera.width(), camera.height())};
  for (int i = 0; i < 3; ++i) { bb_min_[i] = std::numeric_limits<float>::infinity(); bb_max_[i] = -std::numeric_limits<float>::infinity(); }
  for (int c = 0; c < 4; ++c) {
    points_[2 * c] = transform(corners[c], min_depth);
    points_[2 * c + 1] = transform(corners[c], max_depth);
  }
  for (int i = 0; i < 8; ++i) {
    const float v[3] = {points_[i].x, points_[i].y, points_[i].z};
    for (int d = 0; d < 3; ++d) { bb_min_[d] = std::min(bb_min_[d], v[d]); bb_max_[d] = std::max(bb_max_[d], v[d]); }
  }
}

void CameraFrustum::ComputeAxesAndPlanes() {
  axes_[0] = sub(points_[7], points_[6]);
  axes_[1] = sub(points_[3], points_[2]);
  axes_[2] = sub(points_[5], points_[4]);
  axes_[3] = sub(points_[1], points_[0]);
  axes_[4] = sub(points_[2], points_[6]);
  axes_[5] = sub(points_[0], points_[2]);
  auto plane = [&](int i, Vec3f n, Vec3f through) { plane_n_[i] = n; plane_d_[i] = -dotv(n, through); };
  const Vec3f fwd = crossv(axes_[5], axes_[4]);
  plane(0, fwd, points_[1]);
  plane(1, Vec3f{-fwd.x, -fwd.y, -fwd.z}, points_[0]);
  plane(2, crossv(axes_[0], axes_[4]), points_[6]);
  plane(3, crossv(axes_[1], axes_[5]), points_[2]);
  plane(4, crossv(axes_[4], axes_[2]), points_[4]);
  plane(5, crossv(axes_[5], axes_[0]), points_[6]);
  computed_ = true;
}

bool CameraFrustum::Intersects(CameraFrustum* other) {
  for (int d = 0; d < 3; ++d)   // Eigen AlignedBox::intersection(..).isEmpty()
    if (std::max(bb_min_[d], other->bb_min_[d]) > std::min(bb_max_[d], other->bb_max_[d])) return false;
  if (!computed_) ComputeAxesAndPlanes();
  for (int pl = 0; pl < 6; ++pl) {
    int v = 0;
    for (; v < 8; ++v) if (dotv(plane_n_[pl], other->points_[v]) + plane_d_[pl] < 0) break;
    if (v == 8) return false;
  }
  if (!other->computed_) other->ComputeAxesAndPlanes();
  for (int pl = 0; pl < 6; ++pl) {
    int v = 0;
    for (; v < 8; ++v) if (dotv(other->plane_n_[pl], points_[v]) + other->plane_d_[pl] < 0) break;
    if (v == 8) return false;
  }
  for (int a = 0; a < 6; ++a) {
    for (int b = 0; b < 6; ++b) {
      // NOTE: the reference crosses axes_[this_edge] with axes_[other_edge] of the SAME frustum
      // (libvis/src/libvis/camera_frustum.h:124); kept as is.
      const Vec3f dir = crossv(axes_[a], axes_[b]);
      if (dotv(dir, dir) < 1e-5f) continue;
      float tmin = std::numeric_limits<float>::infinity(), tmax = -tmin, omin = tmin, omax = -tmin;
      for (int pnt = 0; pnt < 8; ++pnt) {
        const float tv = dotv(dir, points_[pnt]);
        tmin = std::min(tmin, tv); tmax = std::max(tmax, tv);
        const float ov = dotv(dir, other->points_[pnt]);
        omin = std::min(omin, ov); omax = std::max(omax, ov);
      }
      if (tmax <= omin || tmin >= omax) return false;
    }
  }
  return true;
}

// ------------------------------------------------------------------------------------------------
// DirectBA
// ------------------------------------------------------------------------------------------------
void DirectBA::Check(int rc, const char* what) const {
  if (rc != BSLAM_OK) throw std::runtime_error(std::string(what) + " failed: " + bslam_last_error());
}

DirectBA::DirectBA(int max_surfel_count, float raw_to_float_depth, float baseline_fx, int sparse_surfel_cell_size,
                   float surfel_merge_dist_factor, int min_observation_count_while_bootstrapping_1,
                   int min_observation_count_while_bootstrapping_2, int min_observation_count,
                   const PinholeCamera4f& color_camera_initial_estimate, const PinholeCamera4f& depth_camera_initial_estimate,
                   int pyramid_level_for_color, bool use_depth_residuals, bool use_descriptor_residuals,
                   void* render_window, const SE3f& global_T_anchor_frame, int device)
    : device_(device), color_camera_(color_camera_initial_estimate), depth_camera_(depth_camera_initial_estimate),
      pyramid_level_for_color_(pyramid_level_for_color), use_depth_residuals_(use_depth_residuals),
      use_descriptor_residuals_(use_descriptor_residuals),
      min_observation_count_while_bootstrapping_1_(min_observation_count_while_bootstrapping_1),
      min_observation_count_while_bootstrapping_2_(min_observation_count_while_bootstrapping_2),
      min_observation_count_(min_observation_count), surfel_merge_dist_factor_(surfel_merge_dist_factor),
      global_T_anchor_frame_(global_T_anchor_frame), raw_to_float_depth_(raw_to_float_depth), baseline_fx_(baseline_fx),
      sparse_surfel_cell_size_(sparse_surfel_cell_size) {
  if (render_window) throw std::invalid_argument("render_window must be null: visualisation is out of scope");
  if (sparse_surfel_cell_size < 1) throw std::invalid_argument("sparse_surfel_cell_size must be >= 1");
  Check(bslam_create(device, &ctx_), "bslam_create");
  // this class owns every image the derived per-pixel records depend on and invalidates the cache
  // whenever it changes one of them in place
  Check(bslam_set_keyframe_cache(ctx_, 1), "bslam_set_keyframe_cache");
  HIP_OR_THROW(hipSetDevice(device));
  // BS/direct_ba.cc:108-125
  cfactor_buffer_.reset(new DeviceBuffer<float>((depth_camera_.height() - 1) / sparse_surfel_cell_size + 1,
                                                (depth_camera_.width() - 1) / sparse_surfel_cell_size + 1));
  cfactor_buffer_->Clear(0, nullptr);
  HIP_OR_THROW(hipDeviceSynchronize());
  surfels_.reset(new DeviceBuffer<float>(BSLAM_SURFEL_ATTRIBUTE_COUNT, max_surfel_count));
  active_surfels_.reset(new DeviceBuffer<u8>(1, max_surfel_count));
  for (auto& e : ev_) HIP_OR_THROW(hipEventCreate(&e));
}

DirectBA::~DirectBA() {
  for (auto& e : ev_) { hipError_t r = hipEventDestroy(e); (void)r; }
  keyframes_.clear();
  if (ctx_) bslam_destroy(ctx_);
}

void DirectBA::SetAllReduce(bslam_allreduce_fn fn, void* user) {
  allreduce_ = fn;
  allreduce_user_ = user;
  Check(bslam_set_allreduce(ctx_, fn, user), "bslam_set_allreduce");
}

void DirectBA::InitComm(const void* unique_id, int rank, int world_size) {
  Check(bslam_comm_init(ctx_, unique_id, rank, world_size), "bslam_comm_init");
  sharded_ = world_size > 1 || sharded_;
  comm_ = true;
}

void DirectBA::DestroyComm() {
  Check(bslam_comm_destroy(ctx_), "bslam_comm_destroy");
  comm_ = false;
}

void DirectBA::InvalidateKeyframeCache() { Check(bslam_invalidate_keyframe_cache(ctx_), "bslam_invalidate_keyframe_cache"); }

void DirectBA::SetTextureMode(int mode) { Check(bslam_set_texture_mode(ctx_, mode), "bslam_set_texture_mode"); }

bslam_depth_params DirectBA::depth_params() const {
  bslam_depth_params dp;
  dp.cfactor_buffer = cfactor_buffer_->ToPod();
  dp.a = a_;
  dp.raw_to_float_depth = raw_to_float_depth_;
  dp.baseline_fx = baseline_fx_;
  dp.sparse_surfel_cell_size = sparse_surfel_cell_size_;
  return dp;
}

void DirectBA::AddKeyframe(const std::shared_ptr<Keyframe>& new_keyframe) {
  new_keyframe->SetID(static_cast<int>(keyframes_.size()));
  DetermineNewKeyframeCoVisibility(new_keyframe);
  keyframes_.push_back(new_keyframe);
}

void DirectBA::DetermineNewKeyframeCoVisibility(const std::shared_ptr<Keyframe>& new_keyframe) {
  CameraFrustum new_frustum(depth_camera_, new_keyframe->min_depth(), new_keyframe->max_depth(), new_keyframe->global_T_frame());
  for (const auto& keyframe : keyframes_) {
    if (!keyframe) continue;
    CameraFrustum keyframe_frustum(depth_camera_, keyframe->min_depth(), keyframe->max_depth(), keyframe->global_T_frame());
    if (new_frustum.Intersects(&keyframe_frustum)) {
      new_keyframe->co_visibility_list().push_back(keyframe->id());
      keyframe->co_visibility_list().push_back(new_keyframe->id());
      if (keyframe->activation() == Keyframe::Activation::kInactive) keyframe->SetActivation(Keyframe::Activation::kCovisibleActive);
    }
  }
}

void DirectBA::DetermineCovisibleActiveKeyframes() {
  for (const auto& keyframe : keyframes_) {
    if (!keyframe) continue;
    if (keyframe->activation() == Keyframe::Activation::kActive) {
      for (int covisible_index : keyframe->co_visibility_list()) {
        auto& other = keyframes_[covisible_index];
        if (other && other->activation() == Keyframe::Activation::kInactive) other->SetActivation(Keyframe::Activation::kCovisibleActive);
      }
    }
  }
}

// BS/direct_ba.cc:207-229
void DirectBA::DeleteKeyframe(int keyframe_index) {
  if (keyframe_index < 0 || keyframe_index >= static_cast<int>(keyframes_.size()) || !keyframes_[keyframe_index])
    throw std::invalid_argument("DeleteKeyframe: no keyframe with index " + std::to_string(keyframe_index));
  const std::shared_ptr<Keyframe> frame_to_delete = keyframes_[keyframe_index];
  for (int covis_index : frame_to_delete->co_visibility_list()) {
    Keyframe* covis_frame = keyframes_[covis_index].get();
    if (!covis_frame) continue;
    auto& list = covis_frame->co_visibility_list();
    for (size_t i = 0; i < list.size(); ++i) {
      if (list[i] == keyframe_index) { list.erase(list.begin() + i); break; }
    }
  }
  keyframes_[keyframe_index].reset();
  InvalidateKeyframeCache();   // the derived per-keyframe images are indexed by table position
}

// BS/direct_ba.cc:710-737
void DirectBA::UpdateKeyframeCoVisibility(const std::shared_ptr<Keyframe>& keyframe) {
  for (int covis_index : keyframe->co_visibility_list()) {
    Keyframe* covis_frame = keyframes_[covis_index].get();
    if (!covis_frame) continue;
    auto& list = covis_frame->co_visibility_list();
    for (size_t i = 0; i < list.size(); ++i) {
      if (list[i] == keyframe->id()) { list.erase(list.begin() + i); break; }
    }
  }
  keyframe->co_visibility_list().clear();
  CameraFrustum frustum(depth_camera_, keyframe->min_depth(), keyframe->max_depth(), keyframe->global_T_frame());
  for (const auto& other : keyframes_) {
    if (!other) continue;
    // Quirk kept: the reference does not exclude the keyframe itself (:724-735), so it intersects its own frustum and
    // enters its own list twice (unlike DetermineNewKeyframeCoVisibility, which runs before the keyframe is stored).
    CameraFrustum other_frustum(depth_camera_, other->min_depth(), other->max_depth(), other->global_T_frame());
    if (frustum.Intersects(&other_frustum)) {
      keyframe->co_visibility_list().push_back(other->id());
      other->co_visibility_list().push_back(keyframe->id());
    }
  }
}

// BS/direct_ba.cc:251-338
std::vector<int> DirectBA::MergeKeyframes(hipStream_t /*stream*/, size_t approx_merge_count) {
  constexpr float kPiHalf = 1.57079632679489661923f;
  constexpr float kMaxAngleDifference = 0.5f * kPiHalf;
  constexpr float kMaxEuclideanDistance = 0.3f;
  std::vector<int> deleted;
  if (keyframes_.size() <= 1) return deleted;
  struct MergeDistance { float distance; int prev_id, id, next_id; };
  std::vector<MergeDistance> distances;
  float prev_half_distance = 0;
  int prev_keyframe_id = 0;
  const auto z_axis = [](const SE3f& T, float* z) { const bslam_mat3x4 m = T.Matrix3x4(); z[0] = m.m[2]; z[1] = m.m[6]; z[2] = m.m[10]; };
  const auto translation = [](const SE3f& T, float* t) { const bslam_mat3x4 m = T.Matrix3x4(); t[0] = m.m[3]; t[1] = m.m[7]; t[2] = m.m[11]; };
  for (size_t keyframe_id = 0; keyframe_id + 1 < keyframes_.size(); ++keyframe_id) {
    const auto& keyframe = keyframes_[keyframe_id];
    if (!keyframe) continue;
    const Keyframe* next_keyframe = nullptr;
    for (size_t next_id = keyframe_id + 1; next_id < keyframes_.size(); ++next_id) {
      if (keyframes_[next_id]) { next_keyframe = keyframes_[next_id].get(); break; }
    }
    if (!next_keyframe) break;
    float za[3], zb[3], ta[3], tb[3];
    z_axis(keyframe->global_T_frame(), za);
    z_axis(next_keyframe->global_T_frame(), zb);
    // (clamped: two equal rotations can give a dot product of 1 + 1 ulp, for which acosf returns NaN -- the
    // reference has no guard there)
    const float angle_difference = std::acos(std::min(1.f, std::max(-1.f, za[0] * zb[0] + za[1] * zb[1] + za[2] * zb[2])));
    if (angle_difference > kMaxAngleDifference) continue;
    translation(keyframe->global_T_frame(), ta);
    translation(next_keyframe->global_T_frame(), tb);
    const float dx = ta[0] - tb[0], dy = ta[1] - tb[1], dz = ta[2] - tb[2];
    const float euclidean_distance = std::sqrt(dx * dx + dy * dy + dz * dz);
    if (euclidean_distance > kMaxEuclideanDistance) continue;
    // 90 degrees count like half a metre (:296-297)
    const float next_half_distance = euclidean_distance + (0.5f / kPiHalf) * angle_difference;
    if (keyframe_id > 0) distances.push_back({prev_half_distance + next_half_distance, prev_keyframe_id, static_cast<int>(keyframe_id), next_keyframe->id()});
    prev_half_distance = next_half_distance;
    prev_keyframe_id = static_cast<int>(keyframe_id);
  }
  const size_t sorted = std::min(approx_merge_count, distances.size());
  std::partial_sort(distances.begin(), distances.begin() + sorted, distances.end(),
                    [](const MergeDistance& a, const MergeDistance& b) { return a.distance < b.distance; });
  for (size_t i = 0; i < sorted; ++i) {
    const MergeDistance& merge = distances[i];
    if (!keyframes_[merge.prev_id] || !keyframes_[merge.id] || !keyframes_[merge.next_id]) continue;   // a neighbour went in an earlier merge (:322-327)
    DeleteKeyframe(merge.id);   // the reference "merges" by deleting, too (:329-332)
    deleted.push_back(merge.id);
  }
  return deleted;
}

// BS/direct_ba.cc:456-459
void DirectBA::AssignColors(hipStream_t stream) {
  if (surfels_size_ == 0) return;
  const std::vector<bslam_keyframe_view> views = KeyframeViews();
  const bslam_camera4f color_cam = color_camera_.pod(), depth_cam = depth_camera_.pod();
  const bslam_depth_params dp = depth_params();
  const bslam_buffer2d surfels = surfels_->ToPod();
  Check(bslam_assign_colors(ctx_, stream, &color_cam, &depth_cam, &dp, static_cast<int>(views.size()), views.data(), surfels_size_, &surfels),
        "bslam_assign_colors");
}

// BS/direct_ba.cc:461-546
void DirectBA::ExportToPointCloud(hipStream_t stream, PointCloud* cloud) const {
  cloud->positions.clear();
  cloud->colors.clear();
  cloud->normals.clear();
  if (surfels_size_ == 0) return;
  const size_t n = surfels_size_;
  std::vector<float> rows(6 * n);   // rows 0..5: x, y, z, normal, radius^2, colour
  GetSurfels(stream, rows.data(), n * sizeof(float), 6);
  const float* x = rows.data();
  const float* y = x + n;
  const float* z = y + n;
  const u32* normal = reinterpret_cast<const u32*>(z + n);
  const u32* color = reinterpret_cast<const u32*>(rows.data() + 5 * n);
  const auto ten_bit = [](u32 v) {   // TenBitSignedToFloat, BS/util_nvcc_only.cuh:74-78
    const int16_t s = static_cast<int16_t>((v & 0x03ffu) << 6) >> 6;
    return static_cast<float>(s) * (1.0f / 511);
  };
  cloud->positions.reserve(3 * surfel_count_);
  for (size_t i = 0; i < n; ++i) {
    if (std::isnan(x[i])) continue;   // deleted surfels carry NaN in x (:469-476)
    cloud->positions.insert(cloud->positions.end(), {x[i], y[i], z[i]});
    cloud->colors.insert(cloud->colors.end(), {static_cast<u8>(color[i] & 0xff), static_cast<u8>((color[i] >> 8) & 0xff), static_cast<u8>((color[i] >> 16) & 0xff)});
    const float nx = ten_bit(normal[i]), ny = ten_bit(normal[i] >> 10), nz = ten_bit(normal[i] >> 20);
    const float factor = 1.0f / std::sqrt(nx * nx + ny * ny + nz * nz);
    cloud->normals.insert(cloud->normals.end(), {factor * nx, factor * ny, factor * nz});
  }
}

std::vector<bslam_keyframe_view> DirectBA::KeyframeViews() const {
  // Deleted keyframes (null entries) are skipped by every kernel wrapper of the reference
  // (BS/kernel_opt_geometry.cc:115); here they are simply left out of the table.
  std::vector<bslam_keyframe_view> v;
  v.reserve(keyframes_.size());
  for (const auto& kf : keyframes_) if (kf) v.push_back(kf->view());
  return v;
}

std::shared_ptr<Keyframe> DirectBA::AddKeyframeFromImages(hipStream_t stream, u32 frame_index, const u16* depth_image, const u8* rgb_image,
                                                          const SE3f& global_T_frame) {
  const bslam_depth_params dp = depth_params();
  const bslam_camera4f depth_cam = depth_camera_.pod();
  auto kf = std::make_shared<Keyframe>(ctx_, stream, frame_index, dp, depth_cam, depth_image, color_camera_.width(), color_camera_.height(), rgb_image,
                                       global_T_frame);
  AddKeyframe(kf);
  return kf;
}

// BS/direct_ba.cc:340-405
void DirectBA::CreateSurfelsForKeyframe(hipStream_t stream, bool filter_new_surfels, const std::shared_ptr<Keyframe>& keyframe) {
  const bslam_camera4f color_cam = color_camera_.pod(), depth_cam = depth_camera_.pod();
  const bslam_depth_params dp = depth_params();
  const bslam_buffer2d surfels = surfels_->ToPod();
  const bslam_keyframe_view view = keyframe->view();
  const bslam_mat3x4 global_T_frame = keyframe->global_T_frame().Matrix3x4();
  std::vector<bslam_keyframe_view> covis;
  std::vector<bslam_mat3x4> covis_T_frame;
  for (int index : keyframe->co_visibility_list()) {   // :365-370
    const auto& other = keyframes_[index];
    if (!other) continue;
    covis.push_back(other->view());
    covis_T_frame.push_back((other->frame_T_global() * keyframe->global_T_frame()).Matrix3x4());
  }
  u32 new_surfel_count = 0;
  Check(bslam_create_surfels_for_keyframe(ctx_, stream, filter_new_surfels, GetMinObservationCount(), &color_cam, &depth_cam, &dp, &view,
                                          &global_T_frame, static_cast<int>(covis.size()), covis.data(), covis_T_frame.data(), surfels_size_,
                                          &surfels, &new_surfel_count),
        "bslam_create_surfels_for_keyframe");
  Lock();
  surfels_size_ += new_surfel_count;
  surfel_count_ += new_surfel_count;
  Unlock();
}

void DirectBA::MergeAndCompact(hipStream_t stream, const std::vector<u32>& keyframe_ids) {
  const bslam_camera4f depth_cam = depth_camera_.pod();
  const bslam_depth_params dp = depth_params();
  const bslam_buffer2d surfels = surfels_->ToPod(), active = active_surfels_->ToPod();
  u32 surfel_count = surfel_count_;
  HIP_OR_THROW(hipEventRecord(ev_[10], stream));   // "BA initial surfel merge" (BS/direct_ba_alternating.cc:486-509)
  for (u32 id : keyframe_ids) {
    const auto& kf = keyframes_[id];
    if (!kf) continue;
    const bslam_keyframe_view view = kf->view();
    Check(bslam_determine_supporting_surfels_and_merge(ctx_, stream, surfel_merge_dist_factor_, &depth_cam, &dp, &view, surfels_size_, &surfels,
                                                       &surfel_count),
          "bslam_determine_supporting_surfels_and_merge");
  }
  HIP_OR_THROW(hipEventRecord(ev_[11], stream));
  Lock();
  surfel_count_ = surfel_count;
  Unlock();
  HIP_OR_THROW(hipEventRecord(ev_[12], stream));   // "BA surfel compaction" (:511-533)
  if (!keyframe_ids.empty()) {
    u32 surfels_size = surfels_size_;
    Check(bslam_compact_surfels(ctx_, stream, surfel_count_, &surfels_size, &surfels, &active), "bslam_compact_surfels");
    Lock();
    surfels_size_ = surfels_size;
    Unlock();
  }
  HIP_OR_THROW(hipEventRecord(ev_[13], stream));
}

// BS/direct_ba.cc:566-653
void DirectBA::PerformBASchemeEndTasks(hipStream_t stream, bool do_surfel_updates) {
  u32 surfel_count = surfel_count_;
  u32 surfels_size = surfels_size_;
  const bslam_camera4f depth_cam = depth_camera_.pod();
  const bslam_depth_params dp = depth_params();
  const bslam_buffer2d surfels = surfels_->ToPod();
  if (do_surfel_updates) {
    for (const auto& kf : keyframes_) {
      if (!kf || kf->last_active_in_ba_iteration() != ba_iteration_count_) continue;
      const bslam_keyframe_view view = kf->view();
      Check(bslam_determine_supporting_surfels_and_merge(ctx_, stream, surfel_merge_dist_factor_, &depth_cam, &dp, &view, surfels_size, &surfels,
                                                         &surfel_count),
            "bslam_determine_supporting_surfels_and_merge");
    }
  }
  const std::vector<bslam_keyframe_view> views = KeyframeViews();
  Check(bslam_delete_surfels_and_update_radii(ctx_, stream, GetMinObservationCount(), &depth_cam, &dp, static_cast<int>(views.size()), views.data(),
                                              &surfel_count, surfels_size, &surfels),
        "bslam_delete_surfels_and_update_radii");
  Check(bslam_compact_surfels(ctx_, stream, surfel_count, &surfels_size, &surfels, nullptr), "bslam_compact_surfels");
  Lock();
  surfels_size_ = surfels_size;
  surfel_count_ = surfel_count;
  Unlock();
}

void DirectBA::SetSurfels(hipStream_t stream, const float* host_rows, size_t host_pitch_bytes, u32 count) {
  if (count > static_cast<u32>(surfels_->width())) throw std::invalid_argument("SetSurfels: count exceeds max_surfel_count");
  if (count > 0) {
    HIP_OR_THROW(hipMemcpy2DAsync(surfels_->address(), surfels_->pitch(), host_rows, host_pitch_bytes, static_cast<size_t>(count) * sizeof(float),
                                  BSLAM_SURFEL_DATA_ATTRIBUTE_COUNT, hipMemcpyHostToDevice, stream));
    HIP_OR_THROW(hipStreamSynchronize(stream));
  }
  surfels_size_ = count;
  surfel_count_ = count;
}

void DirectBA::GetSurfels(hipStream_t stream, float* host_rows, size_t host_pitch_bytes, int rows) const {
  if (surfels_size_ == 0) return;
  HIP_OR_THROW(hipMemcpy2DAsync(host_rows, host_pitch_bytes, surfels_->address(), surfels_->pitch(), static_cast<size_t>(surfels_size_) * sizeof(float),
                                rows, hipMemcpyDeviceToHost, stream));
  HIP_OR_THROW(hipStreamSynchronize(stream));
}

void DirectBA::GetActiveSurfels(hipStream_t stream, u8* host) const {
  if (surfels_size_ == 0) return;
  HIP_OR_THROW(hipMemcpyAsync(host, active_surfels_->address(), surfels_size_, hipMemcpyDeviceToHost, stream));
  HIP_OR_THROW(hipStreamSynchronize(stream));
}

// BS/direct_ba_alternating.cc:42-283
void DirectBA::EstimateFramePose(hipStream_t stream, const SE3f& global_T_frame_initial_estimate, const DeviceBuffer<u16>& depth_buffer,
                                 const DeviceBuffer<u16>& normals_buffer, const DeviceBuffer<uchar4_t>& color_buffer,
                                 SE3f* out_global_T_frame_estimate, bool /*called_within_ba*/) {
  SE3f global_T_frame_estimate = global_T_frame_initial_estimate;
  const bslam_camera4f color_cam = color_camera_.pod(), depth_cam = depth_camera_.pod();
  const bslam_depth_params dp = depth_params();
  const bslam_buffer2d depth = depth_buffer.ToPod(), normals = normals_buffer.ToPod(), color = color_buffer.ToPod();
  const bslam_buffer2d surfels = surfels_->ToPod();
  constexpr int kMaxIterations = 30;
  for (int iteration = 0; iteration < kMaxIterations; ++iteration) {
    const bslam_mat3x4 frame_T_global_estimate = global_T_frame_estimate.Inverse().Matrix3x4();
    float H[21], b[6];
    if (surfels_size_ == 0) {
      std::memset(H, 0, sizeof(H));
      std::memset(b, 0, sizeof(b));
    } else {
      Check(bslam_accumulate_pose_estimation_coeffs(ctx_, stream, use_depth_residuals_, use_descriptor_residuals_, &color_cam, &depth_cam, &dp,
                                                    &depth, &normals, &color, &frame_T_global_estimate, surfels_size_, &surfels, 0, nullptr,
                                                    nullptr, H, b),
            "bslam_accumulate_pose_estimation_coeffs");
    }
    float x[6];
    SolveLDLTUpper(6, H, b, x);                                         // :206
    float neg[6];
    for (int i = 0; i < 6; ++i) neg[i] = -1.f * x[i];                   // kDamping = 1 (:213)
    global_T_frame_estimate = global_T_frame_estimate * SE3f::Exp(neg); // :214
    if (IsScale1PoseEstimationConverged(x)) break;                      // :231
  }
  *out_global_T_frame_estimate = global_T_frame_estimate;
}

void DirectBA::BundleAdjustment(hipStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics, bool do_surfel_updates,
                                bool optimize_poses, bool optimize_geometry, int min_iterations, int max_iterations, bool use_pcg,
                                int active_keyframe_window_start, int active_keyframe_window_end, bool increase_ba_iteration_count,
                                int* iterations_done, bool* converged, double time_limit, Timer* timer, int pcg_max_inner_iterations,
                                int pcg_max_keyframes, std::function<bool(int)> progress_function) {
  if (timer) throw std::invalid_argument("Timer objects are not supported; pass nullptr");
  if (optimize_depth_intrinsics && !use_depth_residuals_) optimize_depth_intrinsics = false;      // BS/direct_ba.cc:427-430
  if (optimize_color_intrinsics && !use_descriptor_residuals_) optimize_color_intrinsics = false; // :431-434
  HIP_OR_THROW(hipSetDevice(device_));
  if (use_pcg)
    BundleAdjustmentPCG(stream, optimize_depth_intrinsics, optimize_color_intrinsics, do_surfel_updates, optimize_poses, optimize_geometry,
                        min_iterations, max_iterations, pcg_max_inner_iterations, pcg_max_keyframes, active_keyframe_window_start,
                        active_keyframe_window_end, increase_ba_iteration_count, iterations_done, converged, time_limit, progress_function);
  else
    BundleAdjustmentAlternating(stream, optimize_depth_intrinsics, optimize_color_intrinsics, do_surfel_updates, optimize_poses,
                                optimize_geometry, min_iterations, max_iterations, active_keyframe_window_start, active_keyframe_window_end,
                                increase_ba_iteration_count, iterations_done, converged, time_limit, progress_function);
}

// BS/direct_ba_alternating.cc:285-738
void DirectBA::BundleAdjustmentAlternating(hipStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics,
                                           bool do_surfel_updates, bool optimize_poses, bool optimize_geometry, int min_iterations,
                                           int max_iterations, int active_keyframe_window_start, int active_keyframe_window_end,
                                           bool increase_ba_iteration_count, int* num_iterations_done, bool* converged, double time_limit,
                                           std::function<bool(int)> progress_function) {
  if (converged) *converged = false;
  if (num_iterations_done) *num_iterations_done = 0;
  const auto t_start = std::chrono::steady_clock::now();

  Lock();
  const int fixed_ba_iteration_count = ba_iteration_count_;
  Unlock();
  if (!increase_ba_iteration_count && fixed_ba_iteration_count != last_ba_iteration_count_) {   // :311-317
    last_ba_iteration_count_ = fixed_ba_iteration_count;
    if (scheme_end_tasks_) PerformBASchemeEndTasks(stream, do_surfel_updates);
  }
  std::vector<u32> keyframes_with_new_surfels;

  const bool fixed_active_keyframe_set = active_keyframe_window_start > 0 || active_keyframe_window_end > 0;
  const bool whole_window = active_keyframe_window_start == 0 && active_keyframe_window_end == static_cast<int>(keyframes_.size()) - 1;

  HIP_OR_THROW(hipMemsetAsync(active_surfels_->address(), 0, surfels_size_ * sizeof(u8), stream));   // :338

  const bslam_buffer2d surfels = surfels_->ToPod(), active = active_surfels_->ToPod();

  for (int iteration = 0; iteration < max_iterations; ++iteration) {
    if (progress_function && !progress_function(iteration)) break;
    if (num_iterations_done) ++*num_iterations_done;
    const bslam_camera4f color_cam = color_camera_.pod(), depth_cam = depth_camera_.pod();   // intrinsics may change per iteration

    if (fixed_active_keyframe_set) {   // :352-371
      Lock();
      for (u32 i = 0; i < keyframes_.size(); ++i) {
        if (!keyframes_[i]) continue;
        const bool in_window = i >= static_cast<u32>(active_keyframe_window_start) && i <= static_cast<u32>(active_keyframe_window_end);
        keyframes_[i]->SetActivation(in_window ? Keyframe::Activation::kActive : Keyframe::Activation::kInactive);
      }
      DetermineCovisibleActiveKeyframes();
      Unlock();
    }

    // --- SURFEL CREATION (:396-431) ---
    keyframes_with_new_surfels.clear();
    if (surfels_size_ != surfel_count_) throw std::logic_error("surfels_size_ != surfel_count_ at the start of a BA iteration");
    const u32 old_surfels_size = surfels_size_;
    if (optimize_geometry && do_surfel_updates) {
      Lock();
      for (const auto& kf : keyframes_) {
        if (!kf) continue;
        if (kf->activation() == Keyframe::Activation::kActive && kf->last_active_in_ba_iteration() != fixed_ba_iteration_count) {
          kf->SetLastActiveInBAIteration(fixed_ba_iteration_count);
          keyframes_with_new_surfels.push_back(static_cast<u32>(kf->id()));
        } else if (kf->activation() == Keyframe::Activation::kCovisibleActive && kf->last_covis_in_ba_iteration() != fixed_ba_iteration_count) {
          kf->SetLastCovisInBAIteration(fixed_ba_iteration_count);
        }
      }
      Unlock();
      HIP_OR_THROW(hipEventRecord(ev_[8], stream));
      for (u32 id : keyframes_with_new_surfels) CreateSurfelsForKeyframe(stream, /*filter_new_surfels*/ true, keyframes_[id]);
      HIP_OR_THROW(hipEventRecord(ev_[9], stream));
    }

    const bslam_depth_params dp = depth_params();
    std::vector<bslam_keyframe_view> views = KeyframeViews();
    const int K = static_cast<int>(views.size());

    // --- SURFEL ACTIVATION (:433-452) ---
    HIP_OR_THROW(hipEventRecord(ev_[0], stream));
    if (optimize_geometry && surfels_size_ > old_surfels_size)
      HIP_OR_THROW(hipMemsetAsync(active_surfels_->address() + old_surfels_size, BSLAM_SURFEL_ACTIVE_FLAG, (surfels_size_ - old_surfels_size) * sizeof(u8), stream));
    if (!whole_window) {
      HIP_OR_THROW(hipMemsetAsync(active_surfels_->address(), BSLAM_SURFEL_ACTIVE_FLAG, old_surfels_size * sizeof(u8), stream));
    } else {
      Check(bslam_update_surfel_activation(ctx_, stream, &depth_cam, &dp, K, views.data(), old_surfels_size, &surfels, &active), "bslam_update_surfel_activation");
    }
    HIP_OR_THROW(hipEventRecord(ev_[1], stream));

    // --- GEOMETRY OPTIMIZATION (:467-481) ---
    if (optimize_geometry) {
      HIP_OR_THROW(hipEventRecord(ev_[2], stream));
      Check(bslam_optimize_geometry_iteration(ctx_, stream, use_depth_residuals_, use_descriptor_residuals_, &color_cam, &depth_cam, &dp, K,
                                              views.data(), surfels_size_, &surfels, &active),
            "bslam_optimize_geometry_iteration");
      HIP_OR_THROW(hipEventRecord(ev_[3], stream));
    }

    // --- SURFEL MERGE + COMPACTION (:486-533) ---
    if (do_surfel_updates) MergeAndCompact(stream, keyframes_with_new_surfels);

    // --- POSE OPTIMIZATION (:543-577) ---
    size_t num_converged = 0;
    if (optimize_poses) {
      HIP_OR_THROW(hipEventRecord(ev_[4], stream));
      std::vector<SE3f> estimates(keyframes_.size());
      if (batched_pose_optimization_) {
        std::vector<bslam_se3f> poses;
        std::vector<size_t> index;
        for (size_t i = 0; i < keyframes_.size(); ++i) {
          if (!keyframes_[i]) continue;
          poses.push_back(keyframes_[i]->global_T_frame().ToPod());
          index.push_back(i);
        }
        std::vector<int32_t> iters(poses.size()), conv(poses.size());
        Check(bslam_estimate_frame_poses_batched(ctx_, stream, use_depth_residuals_, use_descriptor_residuals_, &color_cam, &depth_cam, &dp, K,
                                                 views.data(), surfels_size_, &surfels, 30, poses.data(), iters.data(), conv.data(),
                                                 allreduce_, allreduce_user_),
              "bslam_estimate_frame_poses_batched");
        for (size_t j = 0; j < index.size(); ++j) estimates[index[j]] = SE3f::FromPod(poses[j]);
      } else {
        for (size_t i = 0; i < keyframes_.size(); ++i) {
          const auto& kf = keyframes_[i];
          if (!kf || kf->activation() == Keyframe::Activation::kInactive) continue;
          EstimateFramePose(stream, kf->global_T_frame(), kf->depth_buffer(), kf->normals_buffer(), kf->color_buffer(), &estimates[i], true);
        }
      }
      for (size_t i = 0; i < keyframes_.size(); ++i) {
        const auto& kf = keyframes_[i];
        if (!kf || kf->activation() == Keyframe::Activation::kInactive) { ++num_converged; continue; }
        const SE3f pose_difference = kf->frame_T_global() * estimates[i];
        float lg[6];
        pose_difference.Log(lg);
        const bool frame_moved = !IsScale1PoseEstimationConverged(lg);
        Lock();
        kf->set_global_T_frame(estimates[i]);
        if (frame_moved) {
          kf->SetActivation(Keyframe::Activation::kActive);
        } else {
          kf->SetActivation(Keyframe::Activation::kInactive);
          ++num_converged;
        }
        Unlock();
      }
      HIP_OR_THROW(hipEventRecord(ev_[5], stream));
    }

    // --- INTRINSICS OPTIMIZATION (:579-624) ---
    if (optimize_depth_intrinsics || optimize_color_intrinsics) {
      HIP_OR_THROW(hipEventRecord(ev_[14], stream));
      std::vector<bslam_keyframe_view> all_views = KeyframeViews();   // poses may have changed above
      bslam_camera4f out_color = color_cam, out_depth = depth_cam;
      float out_a = a_;
      Check(bslam_optimize_intrinsics(ctx_, stream, optimize_depth_intrinsics, optimize_color_intrinsics, static_cast<int>(all_views.size()),
                                      all_views.data(), &color_cam, &depth_cam, &dp, surfels_size_, &surfels, &out_color, &out_depth, &out_a),
            "bslam_optimize_intrinsics");
      if (surfels_size_ > 0) {
        Lock();
        if (optimize_color_intrinsics) { const float p[4] = {out_color.fx, out_color.fy, out_color.cx, out_color.cy}; color_camera_ = PinholeCamera4f(out_color.width, out_color.height, p); }
        if (optimize_depth_intrinsics) {
          const float p[4] = {out_depth.fx, out_depth.fy, out_depth.cx, out_depth.cy};
          depth_camera_ = PinholeCamera4f(out_depth.width, out_depth.height, p);
          a_ = out_a;
        }
        Unlock();
      }
      HIP_OR_THROW(hipEventRecord(ev_[15], stream));
    }

    // --- TIMING (:626-689), same line format and order as --save_timings ---
    if (timings_stream_) {
      HIP_OR_THROW(hipStreamSynchronize(stream));
      float ms = 0.f;
      *timings_stream_ << "BA_count " << fixed_ba_iteration_count << " inner_iteration " << iteration << " keyframe_count " << keyframes_.size()
                       << " surfel_count " << surfel_count_ << std::endl;
      if (optimize_geometry && do_surfel_updates) { HIP_OR_THROW(hipEventElapsedTime(&ms, ev_[8], ev_[9])); *timings_stream_ << "BA_surfel_creation " << ms << std::endl; }
      HIP_OR_THROW(hipEventElapsedTime(&ms, ev_[0], ev_[1]));
      *timings_stream_ << "BA_surfel_activation " << ms << std::endl;
      if (optimize_geometry) { HIP_OR_THROW(hipEventElapsedTime(&ms, ev_[2], ev_[3])); *timings_stream_ << "BA_geometry_optimization " << ms << std::endl; }
      if (do_surfel_updates) {
        HIP_OR_THROW(hipEventElapsedTime(&ms, ev_[10], ev_[11])); *timings_stream_ << "BA_initial_surfel_merge " << ms << std::endl;
        HIP_OR_THROW(hipEventElapsedTime(&ms, ev_[12], ev_[13])); *timings_stream_ << "BA_surfel_compaction " << ms << std::endl;
      }
      if (optimize_poses) { HIP_OR_THROW(hipEventElapsedTime(&ms, ev_[4], ev_[5])); *timings_stream_ << "BA_pose_optimization " << ms << std::endl; }
      if (optimize_depth_intrinsics || optimize_color_intrinsics) { HIP_OR_THROW(hipEventElapsedTime(&ms, ev_[14], ev_[15])); *timings_stream_ << "BA_intrinsics_optimization " << ms << std::endl; }
    }

    // --- CONVERGENCE (:692-717) ---
    if (iteration >= min_iterations - 1 && (num_converged == keyframes_.size() || !optimize_poses)) {
      if (converged) *converged = true;
      break;
    }
    if (time_limit > 0) {
      const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
      if (elapsed > time_limit) break;
    }
    Lock();
    DetermineCovisibleActiveKeyframes();
    Unlock();
  }

  if (increase_ba_iteration_count) {   // :728-733
    if (scheme_end_tasks_) PerformBASchemeEndTasks(stream, do_surfel_updates);
    ++ba_iteration_count_;
  }
}

// BS/direct_ba_pcg.cc:43-819
void DirectBA::BundleAdjustmentPCG(hipStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics, bool do_surfel_updates,
                                   bool optimize_poses, bool optimize_geometry, int min_iterations, int max_iterations,
                                   int max_inner_iterations, int max_keyframe_count, int /*active_keyframe_window_start*/,
                                   int /*active_keyframe_window_end*/, bool increase_ba_iteration_count, int* num_iterations_done,
                                   bool* converged, double time_limit, std::function<bool(int)> progress_function) {
  if (num_iterations_done) *num_iterations_done = 0;
  if (converged) *converged = false;
  for (const auto& kf : keyframes_)
    if (!kf) throw std::runtime_error("The PCG-based solver does not support deleted keyframes (BS/direct_ba_pcg.cc:138-143)");
  if (keyframes_.empty()) return;
  if (static_cast<int>(keyframes_.size()) > max_keyframe_count) throw std::invalid_argument("keyframe count exceeds pcg_max_keyframes");
  const auto t_start = std::chrono::steady_clock::now();
  if (!increase_ba_iteration_count && ba_iteration_count_ != last_ba_iteration_count_) {   // :155-161
    last_ba_iteration_count_ = ba_iteration_count_;
    if (scheme_end_tasks_) PerformBASchemeEndTasks(stream, do_surfel_updates);
  }

  const bslam_buffer2d surfels = surfels_->ToPod(), active = active_surfels_->ToPod();
  const int K = static_cast<int>(keyframes_.size());
  std::vector<u32> keyframes_with_new_surfels;

  for (int iteration = 0; iteration < max_iterations; ++iteration) {
    if (progress_function && !progress_function(iteration)) break;
    if (num_iterations_done) ++*num_iterations_done;

    keyframes_with_new_surfels.clear();
    if (optimize_geometry && do_surfel_updates) {   // :184-205
      for (const auto& kf : keyframes_) {
        if (kf->activation() == Keyframe::Activation::kActive && kf->last_active_in_ba_iteration() != ba_iteration_count_) {
          kf->SetLastActiveInBAIteration(ba_iteration_count_);
          CreateSurfelsForKeyframe(stream, /*filter_new_surfels*/ true, kf);
          keyframes_with_new_surfels.push_back(static_cast<u32>(kf->id()));
        } else if (kf->activation() == Keyframe::Activation::kCovisibleActive && kf->last_covis_in_ba_iteration() != ba_iteration_count_) {
          kf->SetLastCovisInBAIteration(ba_iteration_count_);
        }
      }
    }

    const bslam_camera4f color_cam = color_camera_.pod(), depth_cam = depth_camera_.pod();
    bslam_depth_params dp = depth_params();
    std::vector<bslam_keyframe_view> views = KeyframeViews();

    HIP_OR_THROW(hipMemsetAsync(active_surfels_->address(), BSLAM_SURFEL_ACTIVE_FLAG, surfels_size_ * sizeof(u8), stream));   // :209
    if (optimize_geometry)                                                                                                    // :218
      Check(bslam_update_surfel_normals(ctx_, stream, &depth_cam, &dp, K, views.data(), surfels_size_, &surfels, &active), "bslam_update_surfel_normals");

    // unknown layout (:229-306)
    const u32 max_unknown_count = 3u * static_cast<u32>(surfels_->width()) + 6u * static_cast<u32>(max_keyframe_count - 1) +
                                  (4u + 1u + static_cast<u32>(cfactor_buffer_->width() * cfactor_buffer_->height())) + 4u;
    if (!pcg_r_ || static_cast<u32>(pcg_r_->width()) < max_unknown_count) {
      pcg_r_.reset(new DeviceBuffer<float>(1, max_unknown_count));
      pcg_M_.reset(new DeviceBuffer<float>(1, max_unknown_count));
      pcg_delta_.reset(new DeviceBuffer<float>(1, max_unknown_count));
      pcg_g_.reset(new DeviceBuffer<float>(1, max_unknown_count));
      pcg_p_.reset(new DeviceBuffer<float>(1, max_unknown_count));
      pcg_scalars_.reset(new DeviceBuffer<float>(1, 4));
    }
    bslam_pcg_layout L;
    std::memset(&L, 0, sizeof(L));
    constexpr u32 kInvalid = 0xffffffffu;
    u32 cur = 0;
    if (optimize_poses) cur += 6u * static_cast<u32>(K - 1);
    L.surfel_unknown_start_index = kInvalid;
    if (optimize_geometry) { L.surfel_unknown_start_index = cur; cur += (use_descriptor_residuals_ ? 3u : 1u) * surfels_size_; }
    L.depth_intrinsics_unknown_start_index = kInvalid;
    L.a_unknown_index = kInvalid;
    if (optimize_depth_intrinsics) {
      L.depth_intrinsics_unknown_start_index = cur;
      cur += 4u + 1u + static_cast<u32>(cfactor_buffer_->width() * cfactor_buffer_->height());
      L.a_unknown_index = L.depth_intrinsics_unknown_start_index + 4;
    }
    L.color_intrinsics_unknown_start_index = kInvalid;
    if (optimize_color_intrinsics) { L.color_intrinsics_unknown_start_index = cur; cur += 4u; }
    L.unknown_count = cur;
    // :328 picks rand() % K.  In a surfel-sharded run every rank must pick the SAME gauge keyframe (the all-reduced r, M, g and
    // the pose rows share one unknown layout), and rand() sequences are per process: there the choice is derived from the BA
    // iteration counter, which all ranks advance in lock step.
    const bool exchange = allreduce_ != nullptr || comm_;
    L.gauge_keyframe_id = fixed_gauge_keyframe_ >= 0 ? fixed_gauge_keyframe_ % K
                          : (exchange ? static_cast<int>(static_cast<unsigned>(ba_iteration_count_) % static_cast<unsigned>(K)) : std::rand() % K);
    L.optimize_poses = optimize_poses; L.optimize_geometry = optimize_geometry;
    L.optimize_depth_intrinsics = optimize_depth_intrinsics; L.optimize_color_intrinsics = optimize_color_intrinsics;
    L.use_depth_residuals = use_depth_residuals_; L.use_descriptor_residuals = use_descriptor_residuals_;
    auto kf_pose_unknown_index = [&](int id) -> u32 {
      if (id == L.gauge_keyframe_id) return kInvalid;
      return static_cast<u32>(6 * (id < L.gauge_keyframe_id ? id : id - 1));
    };

    float* sc = pcg_scalars_->address();
    bslam_pcg_vectors V{pcg_r_->address(), pcg_M_->address(), pcg_delta_->address(), pcg_g_->address(), pcg_p_->address(), sc + 0, sc + 1, sc + 2};

    HIP_OR_THROW(hipEventRecord(ev_[6], stream));
    Check(bslam_pcg_init(ctx_, stream, &L, &color_cam, &depth_cam, &dp, K, views.data(), surfels_size_, &surfels, &V), "bslam_pcg_init");
    Check(bslam_pcg_init2(ctx_, stream, &L, a_, &V), "bslam_pcg_init2");

    float prev_r_norm = std::numeric_limits<float>::infinity();
    int num_iterations_without_improvement = 0;
    for (int step = 0; step < max_inner_iterations; ++step) {   // :382-471
      if (step > 0) std::swap(V.alpha_n, V.beta_n);
      Check(bslam_pcg_step1(ctx_, stream, &L, &color_cam, &depth_cam, &dp, K, views.data(), surfels_size_, &surfels, &V, step > 0), "bslam_pcg_step1");
      float r_norm = 0.f;
      Check(bslam_pcg_step2(ctx_, stream, &L, &V, &r_norm), "bslam_pcg_step2");
      r_norm = std::sqrt(r_norm);
      if (r_norm < prev_r_norm - 1e-3) {
        num_iterations_without_improvement = 0;
      } else {
        ++num_iterations_without_improvement;
        if (num_iterations_without_improvement >= 3) break;
      }
      prev_r_norm = r_norm;
      if (step < max_inner_iterations - 1) Check(bslam_pcg_step3(ctx_, stream, &L, &V), "bslam_pcg_step3");
    }
    HIP_OR_THROW(hipEventRecord(ev_[7], stream));

    // apply delta (:552-642)
    size_t num_converged = 0;
    if (optimize_poses) {
      std::vector<float> delta(6 * static_cast<size_t>(K - 1) + 1);
      if (K > 1) {
        HIP_OR_THROW(hipMemcpyAsync(delta.data(), pcg_delta_->address(), 6 * static_cast<size_t>(K - 1) * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_OR_THROW(hipStreamSynchronize(stream));
      }
      for (const auto& kf : keyframes_) {
        if (kf->id() == L.gauge_keyframe_id) { ++num_converged; continue; }
        const SE3f d = SE3f::Exp(&delta[kf_pose_unknown_index(kf->id())]);
        kf->set_global_T_frame(kf->global_T_frame() * d);
        float lg[6];
        d.Log(lg);
        if (IsScale1PoseEstimationConverged(lg)) ++num_converged;
      }
    }
    if (optimize_geometry)
      Check(bslam_update_surfels_from_pcg_delta(ctx_, stream, surfels_size_, &surfels, use_descriptor_residuals_, L.surfel_unknown_start_index,
                                                pcg_delta_->address()),
            "bslam_update_surfels_from_pcg_delta");
    if (optimize_depth_intrinsics) {   // :590-625, inverse-parameter space
      float buf[5];
      HIP_OR_THROW(hipMemcpyAsync(buf, pcg_delta_->address() + L.depth_intrinsics_unknown_start_index, 5 * sizeof(float), hipMemcpyDeviceToHost, stream));
      HIP_OR_THROW(hipStreamSynchronize(stream));
      const float* p = depth_camera_.parameters();
      const double old_fx_inv = 1. / p[0], old_fy_inv = 1. / p[1];
      const double old_cx_inv = -(p[2] - 0.5) * old_fx_inv, old_cy_inv = -(p[3] - 0.5) * old_fy_inv;
      const double new_fx = 1. / (old_fx_inv + buf[0]), new_fy = 1. / (old_fy_inv + buf[1]);
      const double new_cx = -(new_fx * (old_cx_inv + buf[2])) + 0.5, new_cy = -(new_fy * (old_cy_inv + buf[3])) + 0.5;
      const float np[4] = {static_cast<float>(new_fx), static_cast<float>(new_fy), static_cast<float>(new_cx), static_cast<float>(new_cy)};
      depth_camera_ = PinholeCamera4f(depth_camera_.width(), depth_camera_.height(), np);
      a_ += buf[4];
      const bslam_buffer2d cf = cfactor_buffer_->ToPod();
      Check(bslam_update_cfactors_from_pcg_delta(ctx_, stream, &cf, L.depth_intrinsics_unknown_start_index + 5, pcg_delta_->address()),
            "bslam_update_cfactors_from_pcg_delta");
      InvalidateKeyframeCache();
    }
    if (optimize_color_intrinsics) {   // :628-642
      float buf[4];
      HIP_OR_THROW(hipMemcpyAsync(buf, pcg_delta_->address() + L.color_intrinsics_unknown_start_index, 4 * sizeof(float), hipMemcpyDeviceToHost, stream));
      HIP_OR_THROW(hipStreamSynchronize(stream));
      const float* p = color_camera_.parameters();
      const float np[4] = {p[0] + buf[0], p[1] + buf[1], p[2] + buf[2], p[3] + buf[3]};
      color_camera_ = PinholeCamera4f(color_camera_.width(), color_camera_.height(), np);
    }

    if (do_surfel_updates) MergeAndCompact(stream, keyframes_with_new_surfels);   // :651-690

    if (timings_stream_) {
      HIP_OR_THROW(hipStreamSynchronize(stream));
      float ms = 0.f;
      HIP_OR_THROW(hipEventElapsedTime(&ms, ev_[6], ev_[7]));
      *timings_stream_ << "BA_count " << ba_iteration_count_ << " inner_iteration " << iteration << " keyframe_count " << keyframes_.size()
                       << " surfel_count " << surfel_count_ << std::endl;
      *timings_stream_ << "BA_PCG " << ms << std::endl;
    }

    // convergence (:745-760)
    if (iteration >= min_iterations - 1 && (num_converged == keyframes_.size() || !optimize_poses)) {
      if (converged) *converged = true;
      break;
    }
    if (time_limit > 0) {
      const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
      if (elapsed > time_limit) break;
    }
  }
  if (increase_ba_iteration_count) {   // :764-773
    if (scheme_end_tasks_) PerformBASchemeEndTasks(stream, do_surfel_updates);
    ++ba_iteration_count_;
  } else if (do_surfel_updates) {      // :775-815
    MergeAndCompact(stream, keyframes_with_new_surfels);
  }
}

}  // namespace bslam_host
