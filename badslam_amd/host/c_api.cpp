// c_api.cpp -- flat C API over bslam_host::DirectBA so that the Python tests (ctypes) and other
// FFIs can drive the C++ host class.  Exceptions never cross the boundary: they are turned into a
// negative return code + bsh_last_error().
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <string>

#include "bad_slam.hpp"
#include "direct_ba.hpp"
#include "io.hpp"
#include "pairwise_frame_tracking.hpp"

using namespace bslam_host;

static thread_local std::string g_err;

#define BSH_TRY(...)                          \
  try { __VA_ARGS__; return 0; }              \
  catch (const std::exception& e) { g_err = e.what(); return -1; } \
  catch (...) { g_err = "unknown exception"; return -1; }

static SE3f pose_from7(const float* p) { bslam_se3f q; std::memcpy(q.q, p, 16); std::memcpy(q.t, p + 4, 12); return SE3f::FromPod(q); }
static void pose_to7(const SE3f& T, float* p) { const bslam_se3f q = T.ToPod(); std::memcpy(p, q.q, 16); std::memcpy(p + 4, q.t, 12); }

extern "C" {

const char* bsh_last_error(void) { return g_err.c_str(); }

void* bsh_create(int max_surfel_count, float raw_to_float_depth, float baseline_fx, int sparse_surfel_cell_size,
                 float surfel_merge_dist_factor, int min_obs_boot1, int min_obs_boot2, int min_obs,
                 const float* color_params, int color_w, int color_h, const float* depth_params, int depth_w, int depth_h,
                 int pyramid_level_for_color, int use_depth_residuals, int use_descriptor_residuals, int device) {
  try {
    return new DirectBA(max_surfel_count, raw_to_float_depth, baseline_fx, sparse_surfel_cell_size, surfel_merge_dist_factor, min_obs_boot1,
                        min_obs_boot2, min_obs, PinholeCamera4f(color_w, color_h, color_params), PinholeCamera4f(depth_w, depth_h, depth_params),
                        pyramid_level_for_color, use_depth_residuals != 0, use_descriptor_residuals != 0, nullptr, SE3f(), device);
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}

static std::map<void*, std::unique_ptr<std::ofstream>> g_timings;
void bsh_destroy(void* ba) { delete static_cast<DirectBA*>(ba); g_timings.erase(ba); }

// the kernel library's context of this DirectBA (bslam_profile_* on the BA's own launches)
void* bsh_context(void* ba) { return static_cast<DirectBA*>(ba)->context(); }

int bsh_add_keyframe(void* ba_, void* stream, uint32_t frame_index, float min_depth, float max_depth, const uint16_t* depth,
                     const uint16_t* normals, const uint16_t* radius, const uint8_t* color, const float* pose7) {
  DirectBA* ba = static_cast<DirectBA*>(ba_);
  try {
    auto kf = std::make_shared<Keyframe>(static_cast<hipStream_t>(stream), frame_index, min_depth, max_depth, ba->depth_camera().width(),
                                         ba->depth_camera().height(), depth, normals, radius, reinterpret_cast<const uchar4_t*>(color),
                                         pose_from7(pose7));
    ba->AddKeyframe(kf);
    return kf->id();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

int bsh_set_surfels(void* ba, void* stream, const float* rows, size_t pitch_bytes, uint32_t count) {
  BSH_TRY(static_cast<DirectBA*>(ba)->SetSurfels(static_cast<hipStream_t>(stream), rows, pitch_bytes, count));
}
int bsh_get_surfels(void* ba, void* stream, float* rows, size_t pitch_bytes, int nrows) {
  BSH_TRY(static_cast<DirectBA*>(ba)->GetSurfels(static_cast<hipStream_t>(stream), rows, pitch_bytes, nrows));
}
int bsh_get_active_surfels(void* ba, void* stream, uint8_t* out) {
  BSH_TRY(static_cast<DirectBA*>(ba)->GetActiveSurfels(static_cast<hipStream_t>(stream), out));
}
uint32_t bsh_surfels_size(void* ba) { return static_cast<DirectBA*>(ba)->surfels_size(); }
int bsh_keyframe_count(void* ba) { return static_cast<int>(static_cast<DirectBA*>(ba)->keyframes().size()); }

int bsh_get_keyframe_pose(void* ba, int id, float* pose7) {
  BSH_TRY(pose_to7(static_cast<DirectBA*>(ba)->keyframes().at(id)->global_T_frame(), pose7));
}
int bsh_set_keyframe_pose(void* ba, int id, const float* pose7) {
  BSH_TRY(static_cast<DirectBA*>(ba)->keyframes().at(id)->set_global_T_frame(pose_from7(pose7)));
}
int bsh_get_keyframe_activation(void* ba, int id) { return static_cast<int>(static_cast<DirectBA*>(ba)->keyframes().at(id)->activation()); }
int bsh_set_keyframe_activation(void* ba, int id, int activation) {
  BSH_TRY(static_cast<DirectBA*>(ba)->keyframes().at(id)->SetActivation(static_cast<Keyframe::Activation>(activation)));
}
int bsh_keyframe_covisibility(void* ba, int id, int* out, int capacity) {
  auto& list = static_cast<DirectBA*>(ba)->keyframes().at(id)->co_visibility_list();
  const int n = static_cast<int>(list.size());
  for (int i = 0; i < n && i < capacity; ++i) out[i] = list[i];
  return n;
}
// Replaces a keyframe's depth / normals image (the reference's tests do this through const_cast, e.g.
// BS/test/test_geometry_optimization_geometric_residual.cc:124-139).
int bsh_upload_keyframe_depth(void* ba, void* stream, int id, const uint16_t* depth) {
  BSH_TRY({
    auto& kf = static_cast<DirectBA*>(ba)->keyframes().at(id);
    kf->mutable_depth_buffer().Upload(static_cast<hipStream_t>(stream), depth, static_cast<size_t>(kf->depth_buffer().width()) * 2);
    static_cast<DirectBA*>(ba)->InvalidateKeyframeCache();
  });
}
int bsh_upload_keyframe_normals(void* ba, void* stream, int id, const uint16_t* normals) {
  BSH_TRY({
    auto& kf = static_cast<DirectBA*>(ba)->keyframes().at(id);
    kf->mutable_normals_buffer().Upload(static_cast<hipStream_t>(stream), normals, static_cast<size_t>(kf->normals_buffer().width()) * 2);
    static_cast<DirectBA*>(ba)->InvalidateKeyframeCache();
  });
}

int bsh_add_keyframe_from_images(void* ba_, void* stream, uint32_t frame_index, const uint16_t* depth, const uint8_t* rgb, const float* pose7) {
  DirectBA* ba = static_cast<DirectBA*>(ba_);
  try {
    return ba->AddKeyframeFromImages(static_cast<hipStream_t>(stream), frame_index, depth, rgb, pose_from7(pose7))->id();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
int bsh_get_keyframe_images(void* ba, void* stream, int id, uint16_t* depth, uint16_t* normals, uint16_t* radius, uint8_t* color, float* min_max) {
  BSH_TRY({
    const auto& kf = static_cast<DirectBA*>(ba)->keyframes().at(id);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t w = static_cast<size_t>(kf->depth_buffer().width());
    kf->depth_buffer().Download(s, depth, w * 2);
    kf->normals_buffer().Download(s, normals, w * 2);
    kf->radius_buffer().Download(s, radius, w * 2);
    kf->color_buffer().Download(s, reinterpret_cast<uchar4_t*>(color), static_cast<size_t>(kf->color_buffer().width()) * 4);
    min_max[0] = kf->min_depth();
    min_max[1] = kf->max_depth();
  });
}
// Keyframe management and export (BS/direct_ba.h:95-116, 123-126)
int bsh_keyframe_is_deleted(void* ba, int id) { return static_cast<DirectBA*>(ba)->keyframes().at(id) ? 0 : 1; }
int bsh_delete_keyframe(void* ba, int id) { BSH_TRY(static_cast<DirectBA*>(ba)->DeleteKeyframe(id)); }
int bsh_merge_keyframes(void* ba, void* stream, uint64_t approx_merge_count, int* deleted_ids, int capacity, int* deleted_count) {
  BSH_TRY({
    const std::vector<int> ids = static_cast<DirectBA*>(ba)->MergeKeyframes(static_cast<hipStream_t>(stream), static_cast<size_t>(approx_merge_count));
    for (size_t i = 0; i < ids.size() && static_cast<int>(i) < capacity; ++i) deleted_ids[i] = ids[i];
    *deleted_count = static_cast<int>(ids.size());
  });
}
int bsh_update_keyframe_covisibility(void* ba, int id) {
  BSH_TRY({
    DirectBA* b = static_cast<DirectBA*>(ba);
    if (!b->keyframes().at(id)) throw std::invalid_argument("keyframe was deleted");
    b->UpdateKeyframeCoVisibility(b->keyframes().at(id));
  });
}
int bsh_assign_colors(void* ba, void* stream) { BSH_TRY(static_cast<DirectBA*>(ba)->AssignColors(static_cast<hipStream_t>(stream))); }
// Fills up to `capacity` points; *count receives the number of valid surfels.
int bsh_export_point_cloud(void* ba, void* stream, uint64_t capacity, float* positions, uint8_t* colors, float* normals, uint64_t* count) {
  BSH_TRY({
    DirectBA::PointCloud cloud;
    static_cast<DirectBA*>(ba)->ExportToPointCloud(static_cast<hipStream_t>(stream), &cloud);
    const size_t n = std::min<size_t>(cloud.size(), capacity);
    if (positions) std::memcpy(positions, cloud.positions.data(), n * 3 * sizeof(float));
    if (colors) std::memcpy(colors, cloud.colors.data(), n * 3);
    if (normals) std::memcpy(normals, cloud.normals.data(), n * 3 * sizeof(float));
    *count = cloud.size();
  });
}
int bsh_set_scheme_end_tasks(void* ba, int enable) { BSH_TRY(static_cast<DirectBA*>(ba)->SetSchemeEndTasks(enable != 0)); }
int bsh_create_surfels_for_keyframe(void* ba, void* stream, int filter_new_surfels, int keyframe_id) {
  BSH_TRY({
    DirectBA* b = static_cast<DirectBA*>(ba);
    b->CreateSurfelsForKeyframe(static_cast<hipStream_t>(stream), filter_new_surfels != 0, b->keyframes().at(keyframe_id));
  });
}

int bsh_set_options(void* ba, int batched_pose_optimization, int pcg_gauge_keyframe, int texture_mode) {
  BSH_TRY({
    DirectBA* b = static_cast<DirectBA*>(ba);
    b->SetBatchedPoseOptimization(batched_pose_optimization != 0);
    b->SetPCGGaugeKeyframe(pcg_gauge_keyframe);
    b->SetTextureMode(texture_mode);
  });
}
// --save_timings (BS/main.cc:660-662): the BA phase timing lines of BS/direct_ba_alternating.cc:630-688 go to `path`
// (nullptr / "": stop).  The stream lives as long as the DirectBA it was set on.
int bsh_set_timings_file(void* ba, const char* path) {
  BSH_TRY({
    DirectBA* b = static_cast<DirectBA*>(ba);
    b->SetTimingsStream(nullptr);
    g_timings.erase(ba);
    if (path && *path) {
      auto f = std::make_unique<std::ofstream>(path);
      if (!*f) throw std::runtime_error(std::string("cannot open ") + path);
      b->SetTimingsStream(f.get());
      g_timings[ba] = std::move(f);
    }
  });
}
int bsh_set_allreduce(void* ba, bslam_allreduce_fn fn, void* user) { BSH_TRY(static_cast<DirectBA*>(ba)->SetAllReduce(fn, user)); }
int bsh_comm_init(void* ba, const void* unique_id, int rank, int world_size) { BSH_TRY(static_cast<DirectBA*>(ba)->InitComm(unique_id, rank, world_size)); }
int bsh_comm_destroy(void* ba) { BSH_TRY(static_cast<DirectBA*>(ba)->DestroyComm()); }

int bsh_estimate_frame_pose(void* ba_, void* stream, int keyframe_id, const float* init7, float* out7) {
  BSH_TRY({
    DirectBA* ba = static_cast<DirectBA*>(ba_);
    const auto& kf = ba->keyframes().at(keyframe_id);
    SE3f out;
    ba->EstimateFramePose(static_cast<hipStream_t>(stream), pose_from7(init7), kf->depth_buffer(), kf->normals_buffer(), kf->color_buffer(), &out, false);
    pose_to7(out, out7);
  });
}

int bsh_bundle_adjustment(void* ba, void* stream, int optimize_depth_intrinsics, int optimize_color_intrinsics, int do_surfel_updates,
                          int optimize_poses, int optimize_geometry, int min_iterations, int max_iterations, int use_pcg,
                          int active_keyframe_window_start, int active_keyframe_window_end, int increase_ba_iteration_count,
                          int pcg_max_inner_iterations, int* iterations_done, int* converged) {
  BSH_TRY({
    bool conv = false;
    int iters = 0;
    static_cast<DirectBA*>(ba)->BundleAdjustment(static_cast<hipStream_t>(stream), optimize_depth_intrinsics != 0, optimize_color_intrinsics != 0,
                                                 do_surfel_updates != 0, optimize_poses != 0, optimize_geometry != 0, min_iterations, max_iterations,
                                                 use_pcg != 0, active_keyframe_window_start, active_keyframe_window_end,
                                                 increase_ba_iteration_count != 0, &iters, &conv, 0, nullptr, pcg_max_inner_iterations);
    if (iterations_done) *iterations_done = iters;
    if (converged) *converged = conv ? 1 : 0;
  });
}

int bsh_set_intrinsics(void* ba_, const float* color4, const float* depth4, float a) {
  BSH_TRY({
    DirectBA* ba = static_cast<DirectBA*>(ba_);
    if (color4) ba->SetColorCamera(PinholeCamera4f(ba->color_camera().width(), ba->color_camera().height(), color4));
    if (depth4) ba->SetDepthCamera(PinholeCamera4f(ba->depth_camera().width(), ba->depth_camera().height(), depth4));
    ba->SetA(a);
  });
}
int bsh_get_cfactor(void* ba_, void* stream, float* out) {
  BSH_TRY({
    DirectBA* ba = static_cast<DirectBA*>(ba_);
    ba->cfactor_buffer().Download(static_cast<hipStream_t>(stream), out, static_cast<size_t>(ba->cfactor_buffer().width()) * sizeof(float));
  });
}

int bsh_get_intrinsics(void* ba_, float* color4, float* depth4, float* a) {
  BSH_TRY({
    DirectBA* ba = static_cast<DirectBA*>(ba_);
    std::memcpy(color4, ba->color_camera().parameters(), 16);
    std::memcpy(depth4, ba->depth_camera().parameters(), 16);
    *a = ba->a();
  });
}

// ---- file formats (io.hpp) ----
int bsh_png_info(const char* path, int* whbc /* width, height, bit depth, channels */) {
  PngInfo info;
  if (!ReadPngInfo(path, &info)) { g_err = std::string("cannot read PNG header of ") + path; return -1; }
  whbc[0] = info.width; whbc[1] = info.height; whbc[2] = info.bit_depth; whbc[3] = info.channels;
  return 0;
}
int bsh_read_png_gray16(const char* path, uint16_t* out, size_t capacity) {
  int w, h;
  std::vector<uint16_t> img;
  if (!ReadPngGray16(path, &w, &h, &img) || img.size() > capacity) { g_err = std::string("cannot read 16-bit PNG ") + path; return -1; }
  std::memcpy(out, img.data(), img.size() * sizeof(uint16_t));
  return 0;
}
int bsh_read_png_rgb8(const char* path, uint8_t* out, size_t capacity) {
  int w, h;
  std::vector<uint8_t> img;
  if (!ReadPngRgb8(path, &w, &h, &img) || img.size() > capacity) { g_err = std::string("cannot read 8-bit PNG ") + path; return -1; }
  std::memcpy(out, img.data(), img.size());
  return 0;
}
void* bsh_tum_open(const char* folder, const char* trajectory_filename) {
  auto ds = std::make_unique<TumDataset>();
  if (!ReadTUMRGBDDatasetAssociatedAndCalibrated(folder, trajectory_filename ? trajectory_filename : "", ds.get())) {
    g_err = std::string("cannot read TUM RGB-D dataset ") + folder;
    return nullptr;
  }
  return ds.release();
}
void bsh_tum_close(void* ds) { delete static_cast<TumDataset*>(ds); }
int bsh_tum_frame_count(void* ds) { return static_cast<int>(static_cast<TumDataset*>(ds)->frames.size()); }
int bsh_tum_camera(void* ds_, float* params4, int* width, int* height) {
  const TumDataset* ds = static_cast<TumDataset*>(ds_);
  std::memcpy(params4, ds->camera_parameters, 16);
  *width = ds->width; *height = ds->height;
  return 0;
}
int bsh_tum_frame(void* ds_, int i, char* rgb_path, char* depth_path, char* rgb_ts, char* depth_ts, size_t capacity, float* rgb_pose7, float* depth_pose7) {
  const TumDataset* ds = static_cast<TumDataset*>(ds_);
  if (i < 0 || i >= static_cast<int>(ds->frames.size())) { g_err = "frame index out of range"; return -1; }
  const TumFrame& f = ds->frames[static_cast<size_t>(i)];
  std::snprintf(rgb_path, capacity, "%s", f.rgb_path.c_str());
  std::snprintf(depth_path, capacity, "%s", f.depth_path.c_str());
  std::snprintf(rgb_ts, capacity, "%s", f.rgb_timestamp_string.c_str());
  std::snprintf(depth_ts, capacity, "%s", f.depth_timestamp_string.c_str());
  pose_to7(f.rgb_global_T_frame, rgb_pose7);
  pose_to7(f.depth_global_T_frame, depth_pose7);
  return 0;
}
int bsh_save_poses(int count, const char* const* timestamp_strings, const float* poses7, int start_frame, const char* path) {
  std::vector<std::string> ts(static_cast<size_t>(count));
  std::vector<SE3f> poses(static_cast<size_t>(count));
  for (int i = 0; i < count; ++i) { ts[i] = timestamp_strings[i]; poses[i] = pose_from7(poses7 + 7 * i); }
  if (!SavePoses(ts, poses, start_frame, path)) { g_err = std::string("cannot write ") + path; return -1; }
  return 0;
}
int bsh_save_calibration_arrays(const char* base, const float* depth4, const float* color4, float a, int w, int h, const float* cfactor) {
  if (!SaveCalibration(base, depth4, color4, a, w, h, cfactor)) { g_err = std::string("cannot write calibration ") + base; return -1; }
  return 0;
}
int bsh_load_calibration_arrays(const char* base, float* depth4, float* color4, float* a, int w, int h, float* cfactor) {
  if (!LoadCalibration(base, depth4, color4, a, w, h, cfactor)) { g_err = std::string("cannot read calibration ") + base; return -1; }
  return 0;
}
// SaveCalibration / LoadCalibration of BS/io.cc:570-700 on a DirectBA
int bsh_save_calibration(void* ba_, void* stream, const char* base) {
  BSH_TRY({
    DirectBA* ba = static_cast<DirectBA*>(ba_);
    const int w = ba->cfactor_buffer().width(), h = ba->cfactor_buffer().height();
    std::vector<float> cf(static_cast<size_t>(w) * h);
    ba->cfactor_buffer().Download(static_cast<hipStream_t>(stream), cf.data(), static_cast<size_t>(w) * sizeof(float));
    if (!SaveCalibration(base, ba->depth_camera().parameters(), ba->color_camera().parameters(), ba->a(), w, h, cf.data()))
      throw std::runtime_error(std::string("cannot write calibration ") + base);
  });
}
int bsh_load_calibration(void* ba_, void* stream, const char* base) {
  BSH_TRY({
    DirectBA* ba = static_cast<DirectBA*>(ba_);
    const int w = ba->cfactor_buffer().width(), h = ba->cfactor_buffer().height();
    std::vector<float> cf(static_cast<size_t>(w) * h);
    float d[4], c[4], a = 0.f;
    if (!LoadCalibration(base, d, c, &a, w, h, cf.data())) throw std::runtime_error(std::string("cannot read calibration ") + base);
    ba->SetDepthCamera(PinholeCamera4f(ba->depth_camera().width(), ba->depth_camera().height(), d));
    ba->SetColorCamera(PinholeCamera4f(ba->color_camera().width(), ba->color_camera().height(), c));
    ba->SetA(a);
    ba->UploadCFactor(static_cast<hipStream_t>(stream), cf.data());
  });
}

// ---- state file v1 ----
void* bsh_state_load(const char* path) {
  auto st = std::make_unique<StateV1>();
  std::string err;
  if (!LoadState(path, st.get(), &err)) { g_err = err; return nullptr; }
  return st.release();
}
void bsh_state_free(void* st) { delete static_cast<StateV1*>(st); }
int bsh_state_save(void* st, const char* path) {
  if (!SaveState(*static_cast<StateV1*>(st), path)) { g_err = std::string("cannot write ") + path; return -1; }
  return 0;
}
// ints: base_kf_id, last_frame_index, frame count, keyframe count, surfel_count, surfels_size, ba_iteration_count, cell size;
// floats: a, raw_to_float_depth, baseline_fx, depth camera (4), colour camera (4)
int bsh_state_summary(void* st_, int32_t* ints8, float* floats11) {
  const StateV1* st = static_cast<StateV1*>(st_);
  ints8[0] = st->base_kf_id; ints8[1] = st->last_frame_index; ints8[2] = static_cast<int32_t>(st->frame_global_T_frame.size());
  ints8[3] = static_cast<int32_t>(st->keyframes.size()); ints8[4] = st->surfel_count; ints8[5] = st->surfels_size; ints8[6] = st->ba_iteration_count;
  ints8[7] = st->sparse_surfel_cell_size;
  floats11[0] = st->a; floats11[1] = st->raw_to_float_depth; floats11[2] = st->baseline_fx;
  std::memcpy(floats11 + 3, st->depth_camera_parameters, 16);
  std::memcpy(floats11 + 7, st->color_camera_parameters, 16);
  return 0;
}
// The DirectBA part of SaveState (BS/io.cc:106-178): cameras, deformation, keyframe metadata, surfels, BA counters.  The SLAM
// front-end fields (config, motion model, queue) keep their defaults; frame poses = the keyframes' poses at their frame indices.
int bsh_state_save_from_ba(void* ba_, void* stream_, int frame_count, const char* path) {
  BSH_TRY({
    DirectBA* ba = static_cast<DirectBA*>(ba_);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StateV1 st;
    st.frame_global_T_frame.assign(static_cast<size_t>(frame_count), SE3f());
    st.config.raw_to_float_depth = ba->depth_params().raw_to_float_depth;
    st.config.baseline_fx = ba->depth_params().baseline_fx;
    st.config.sparse_surfel_cell_size = ba->depth_params().sparse_surfel_cell_size;
    st.config.use_geometric_residuals = ba->use_depth_residuals();
    st.config.use_photometric_residuals = ba->use_descriptor_residuals();
    st.color_camera_width = ba->color_camera().width(); st.color_camera_height = ba->color_camera().height();
    st.depth_camera_width = ba->depth_camera().width(); st.depth_camera_height = ba->depth_camera().height();
    std::memcpy(st.color_camera_parameters, ba->color_camera().parameters(), 16);
    std::memcpy(st.depth_camera_parameters, ba->depth_camera().parameters(), 16);
    st.cfactor_width = ba->cfactor_buffer().width(); st.cfactor_height = ba->cfactor_buffer().height();
    st.cfactor.resize(static_cast<size_t>(st.cfactor_width) * st.cfactor_height);
    ba->cfactor_buffer().Download(stream, st.cfactor.data(), static_cast<size_t>(st.cfactor_width) * sizeof(float));
    const bslam_depth_params dp = ba->depth_params();
    st.a = dp.a; st.raw_to_float_depth = dp.raw_to_float_depth; st.baseline_fx = dp.baseline_fx; st.sparse_surfel_cell_size = dp.sparse_surfel_cell_size;
    for (const auto& kf : ba->keyframes()) {
      StateKeyframeV1 k;
      if (kf) {
        k.id = kf->id(); k.frame_index = static_cast<int32_t>(kf->frame_index()); k.activation = static_cast<int32_t>(kf->activation());
        k.last_active_in_ba_iteration = kf->last_active_in_ba_iteration(); k.last_covis_in_ba_iteration = kf->last_covis_in_ba_iteration();
        if (k.frame_index < 0 || k.frame_index >= frame_count) throw std::invalid_argument("keyframe frame index outside the frame list");
        st.frame_global_T_frame[static_cast<size_t>(k.frame_index)] = kf->global_T_frame();
      }
      st.keyframes.push_back(k);
    }
    st.surfel_count = static_cast<int32_t>(ba->surfel_count()); st.surfels_size = static_cast<int32_t>(ba->surfels_size());
    st.surfels.resize(static_cast<size_t>(8) * st.surfels_size);
    if (st.surfels_size) ba->GetSurfels(stream, st.surfels.data(), static_cast<size_t>(st.surfels_size) * sizeof(float), 8);
    st.ba_iteration_count = ba->ba_iteration_count(); st.last_ba_iteration_count = ba->last_ba_iteration_count();
    st.use_depth_residuals = ba->use_depth_residuals(); st.use_descriptor_residuals = ba->use_descriptor_residuals();
    st.min_observation_count_while_bootstrapping_1 = ba->min_observation_count_while_bootstrapping_1();
    st.min_observation_count_while_bootstrapping_2 = ba->min_observation_count_while_bootstrapping_2();
    st.min_observation_count = ba->min_observation_count();
    st.surfel_merge_dist_factor = ba->surfel_merge_dist_factor();
    if (!SaveState(st, path)) throw std::runtime_error(std::string("cannot write ") + path);
  });
}
// The DirectBA part of LoadState (BS/io.cc:300-372, 446-475): the keyframes must already exist (the reference re-creates them
// from the dataset images); their poses come from the per-frame pose list.
int bsh_state_load_into_ba(void* ba_, void* stream_, const char* path) {
  BSH_TRY({
    DirectBA* ba = static_cast<DirectBA*>(ba_);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StateV1 st;
    std::string err;
    if (!LoadState(path, &st, &err)) throw std::runtime_error(err);
    if (st.cfactor_width != ba->cfactor_buffer().width() || st.cfactor_height != ba->cfactor_buffer().height())
      throw std::runtime_error("cfactor buffer size mismatch");                      // BS/io.cc:331-335
    if (st.keyframes.size() != ba->keyframes().size()) throw std::runtime_error("keyframe count mismatch");
    if (static_cast<u32>(st.surfels_size) > static_cast<u32>(ba->max_surfel_count())) throw std::runtime_error("surfel count exceeds max_surfel_count");   // :452-455
    ba->SetColorCamera(PinholeCamera4f(st.color_camera_width, st.color_camera_height, st.color_camera_parameters));
    ba->SetDepthCamera(PinholeCamera4f(st.depth_camera_width, st.depth_camera_height, st.depth_camera_parameters));
    ba->SetA(st.a);
    ba->UploadCFactor(stream, st.cfactor.data());
    for (size_t i = 0; i < st.keyframes.size(); ++i) {
      const StateKeyframeV1& k = st.keyframes[i];
      const auto& kf = ba->keyframes()[i];
      if ((k.id < 0) != (kf == nullptr)) throw std::runtime_error("keyframe list mismatch");
      if (!kf) continue;
      kf->SetActivation(static_cast<Keyframe::Activation>(k.activation));
      kf->SetLastActiveInBAIteration(k.last_active_in_ba_iteration);
      kf->SetLastCovisInBAIteration(k.last_covis_in_ba_iteration);
      kf->set_global_T_frame(st.frame_global_T_frame.at(static_cast<size_t>(k.frame_index)));
    }
    ba->SetSurfels(stream, st.surfels.data(), static_cast<size_t>(st.surfels_size) * sizeof(float), static_cast<u32>(st.surfels_size));
    ba->SetBAIterationCounts(st.ba_iteration_count, st.last_ba_iteration_count);
  });
}

// TrackFramePairwise on two keyframes of a DirectBA (base = the reference keyframe, tracked = the frame to localise):
// out = base_T_tracked.  iterations: num_scales ints (may be null).
int bsh_track_keyframe_pair_ex(void* ba_, void* stream, int tracked_id, int base_id, int num_scales, int test_different_initial_estimates,
                               const float* init1_pose7, const float* init2_pose7, float* out_pose7, int* iterations, int use_pyramid_level_0, int use_gradmag) {
  BSH_TRY({
    DirectBA* ba = static_cast<DirectBA*>(ba_);
    const auto& tracked = ba->keyframes().at(tracked_id);
    const auto& base = ba->keyframes().at(base_id);
    PairwiseFrameTrackingBuffers buffers(ba->depth_camera().width(), ba->depth_camera().height(), ba->color_camera().width(), ba->color_camera().height(),
                                         num_scales);
    SE3f out;
    const bslam_depth_params dp = ba->depth_params();
    for (int s = 0; s < num_scales; ++s) iterations[s] = 0;
    TrackFramePairwise(ba->context(), static_cast<hipStream_t>(stream), &buffers, ba->color_camera(), ba->depth_camera(), dp, ba->use_depth_residuals(),
                       ba->use_descriptor_residuals(), tracked->depth_buffer(), tracked->normals_buffer(), tracked->color_buffer(), base->depth_buffer(),
                       base->normals_buffer(), base->color_buffer(), test_different_initial_estimates != 0, pose_from7(init1_pose7),
                       pose_from7(init2_pose7 ? init2_pose7 : init1_pose7), &out, iterations, use_pyramid_level_0 != 0, use_gradmag != 0);
    pose_to7(out, out_pose7);
  });
}

int bsh_track_keyframe_pair(void* ba_, void* stream, int tracked_id, int base_id, int num_scales, int test_different_initial_estimates,
                            const float* init1_pose7, const float* init2_pose7, float* out_pose7, int* iterations) {
  BSH_TRY({
    DirectBA* ba = static_cast<DirectBA*>(ba_);
    const auto& tracked = ba->keyframes().at(tracked_id);
    const auto& base = ba->keyframes().at(base_id);
    PairwiseFrameTrackingBuffers buffers(ba->depth_camera().width(), ba->depth_camera().height(), ba->color_camera().width(), ba->color_camera().height(),
                                         num_scales);
    SE3f out;
    const bslam_depth_params dp = ba->depth_params();
    TrackFramePairwise(ba->context(), static_cast<hipStream_t>(stream), &buffers, ba->color_camera(), ba->depth_camera(), dp, ba->use_depth_residuals(),
                       ba->use_descriptor_residuals(), tracked->depth_buffer(), tracked->normals_buffer(), tracked->color_buffer(), base->depth_buffer(),
                       base->normals_buffer(), base->color_buffer(), test_different_initial_estimates != 0, pose_from7(init1_pose7),
                       pose_from7(init2_pose7 ? init2_pose7 : init1_pose7), &out, iterations);
    pose_to7(out, out_pose7);
  });
}

// ---- BadSlam front end (host/bad_slam.hpp) ----
// cfg: [keyframe_interval, max_num_ba_iterations_per_keyframe, num_scales, max_surfel_count, sparse_surfel_cell_size, use_motion_model,
//       use_geometric_residuals, use_photometric_residuals, do_surfel_updates, use_pcg, optimize_intrinsics, disable_deactivation, start_frame]
// fcfg: [raw_to_float_depth, max_depth, baseline_fx]
void* bsh_slam_create(const int* cfg, const float* fcfg, int color_width, int color_height, const float* color_params, int depth_width, int depth_height,
                      const float* depth_params, int device) {
  try {
    BadSlamConfigV1 c;
    c.keyframe_interval = cfg[0]; c.max_num_ba_iterations_per_keyframe = cfg[1]; c.num_scales = cfg[2]; c.max_surfel_count = cfg[3];
    c.sparse_surfel_cell_size = cfg[4]; c.use_motion_model = cfg[5] != 0; c.use_geometric_residuals = cfg[6] != 0;
    c.use_photometric_residuals = cfg[7] != 0; c.do_surfel_updates = cfg[8] != 0; c.use_pcg = cfg[9] != 0; c.optimize_intrinsics = cfg[10] != 0;
    c.disable_deactivation = cfg[11] != 0; c.start_frame = cfg[12];
    c.raw_to_float_depth = fcfg[0]; c.max_depth = fcfg[1]; c.baseline_fx = fcfg[2];
    return new BadSlam(c, PinholeCamera4f(color_width, color_height, color_params), PinholeCamera4f(depth_width, depth_height, depth_params), device);
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void bsh_slam_destroy(void* slam) { delete static_cast<BadSlam*>(slam); }
void* bsh_slam_direct_ba(void* slam) { return &static_cast<BadSlam*>(slam)->direct_ba(); }
int bsh_slam_process_frame(void* slam, int frame_index, const uint16_t* depth, const uint8_t* rgb, int force_keyframe) {
  BSH_TRY(static_cast<BadSlam*>(slam)->ProcessFrame(frame_index, depth, rgb, force_keyframe != 0));
}
int bsh_slam_run_bundle_adjustment(void* slam, int frame_index, int optimize_depth_intrinsics, int optimize_color_intrinsics, int optimize_poses,
                                   int optimize_geometry, int min_iterations, int max_iterations, int window_start, int window_end,
                                   int increase_ba_iteration_count, int* iterations_done, int* converged) {
  BSH_TRY({
    bool conv = false;
    int done = 0;
    static_cast<BadSlam*>(slam)->RunBundleAdjustment(static_cast<uint32_t>(frame_index), optimize_depth_intrinsics != 0, optimize_color_intrinsics != 0,
                                                     optimize_poses != 0, optimize_geometry != 0, min_iterations, max_iterations, window_start, window_end,
                                                     increase_ba_iteration_count != 0, &done, &conv);
    if (iterations_done) *iterations_done = done;
    if (converged) *converged = conv ? 1 : 0;
  });
}
int bsh_slam_frame_count(void* slam) { return static_cast<int>(static_cast<BadSlam*>(slam)->frame_poses().size()); }
int bsh_slam_get_frame_poses(void* slam, float* poses7, int capacity) {
  BSH_TRY({
    const auto& poses = static_cast<BadSlam*>(slam)->frame_poses();
    for (size_t i = 0; i < poses.size() && static_cast<int>(i) < capacity; ++i) pose_to7(poses[i], poses7 + 7 * i);
  });
}
// state: [keyframe_created, pose_estimated, num_planned_ba_iterations, base keyframe id or -1, motion model length]
int bsh_slam_state(void* slam, int* state) {
  BSH_TRY({
    BadSlam* s = static_cast<BadSlam*>(slam);
    state[0] = s->keyframe_created(); state[1] = s->pose_estimated(); state[2] = s->num_planned_ba_iterations();
    state[3] = s->base_kf() ? s->base_kf()->id() : -1;
    state[4] = static_cast<int>(s->motion_model_base_kf_tr_frame().size());
  });
}

}  // extern "C"
