// io.cpp -- see io.hpp.
#include "io.hpp"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <limits>
#include <sstream>

namespace bslam_host {

// ------------------------------------------------------------------------------------------------
// PNG (ISO/IEC 15948): signature, IHDR, concatenated IDAT -> zlib inflate -> per-scanline filters
// ------------------------------------------------------------------------------------------------
namespace {

struct PngRaw {
  PngInfo info;
  int color_type = 0;
  std::vector<uint8_t> pixels;   // unfiltered scanlines, height x (width * bytes per pixel)
};

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }

bool read_file(const std::string& path, std::vector<uint8_t>* out) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  out->resize(n > 0 ? static_cast<size_t>(n) : 0);
  const size_t got = out->empty() ? 0 : std::fread(out->data(), 1, out->size(), f);
  std::fclose(f);
  return got == out->size();
}

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

bool decode_png(const std::string& path, bool header_only, PngRaw* out) {
  std::vector<uint8_t> file;
  if (!read_file(path, &file) || file.size() < 33) return false;
  static const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (std::memcmp(file.data(), kSig, 8) != 0) return false;
  size_t pos = 8;
  std::vector<uint8_t> idat;
  bool have_ihdr = false;
  int interlace = 0;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const uint8_t* type = &file[pos + 4];
    if (pos + 12 + len > file.size()) return false;
    const uint8_t* data = &file[pos + 8];
    if (std::memcmp(type, "IHDR", 4) == 0 && len == 13) {
      out->info.width = static_cast<int>(be32(data));
      out->info.height = static_cast<int>(be32(data + 4));
      out->info.bit_depth = data[8];
      out->color_type = data[9];
      interlace = data[12];
      switch (out->color_type) {
        case 0: out->info.channels = 1; break;
        case 2: out->info.channels = 3; break;
        case 4: out->info.channels = 2; break;
        case 6: out->info.channels = 4; break;
        default: return false;   // palette images are not used by the datasets
      }
      have_ihdr = true;
      if (header_only) return true;
    } else if (std::memcmp(type, "IDAT", 4) == 0) {
      idat.insert(idat.end(), data, data + len);
    } else if (std::memcmp(type, "IEND", 4) == 0) {
      break;
    }
    pos += 12 + len;
  }
  if (!have_ihdr || interlace != 0 || (out->info.bit_depth != 8 && out->info.bit_depth != 16)) return false;
  const size_t bpp = static_cast<size_t>(out->info.channels) * (out->info.bit_depth / 8);
  const size_t stride = static_cast<size_t>(out->info.width) * bpp;
  std::vector<uint8_t> raw((stride + 1) * static_cast<size_t>(out->info.height));
  uLongf raw_len = static_cast<uLongf>(raw.size());
  if (uncompress(raw.data(), &raw_len, idat.data(), static_cast<uLong>(idat.size())) != Z_OK || raw_len != raw.size()) return false;
  out->pixels.assign(stride * static_cast<size_t>(out->info.height), 0);
  for (int y = 0; y < out->info.height; ++y) {
    const uint8_t filter = raw[(stride + 1) * y];
    const uint8_t* in = &raw[(stride + 1) * y + 1];
    uint8_t* cur = &out->pixels[stride * y];
    const uint8_t* up = y > 0 ? &out->pixels[stride * (y - 1)] : nullptr;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
      int v = in[i];
      switch (filter) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) / 2; break;
        case 4: v += paeth(a, b, c); break;
        default: return false;
      }
      cur[i] = static_cast<uint8_t>(v);
    }
  }
  return true;
}

}  // namespace

bool ReadPngInfo(const std::string& path, PngInfo* info) {
  PngRaw raw;
  if (!decode_png(path, true, &raw)) return false;
  *info = raw.info;
  return true;
}

bool ReadPngRgb8(const std::string& path, int* width, int* height, std::vector<uint8_t>* rgb) {
  PngRaw raw;
  if (!decode_png(path, false, &raw) || raw.info.bit_depth != 8) return false;
  *width = raw.info.width;
  *height = raw.info.height;
  const size_t n = static_cast<size_t>(raw.info.width) * raw.info.height;
  rgb->resize(n * 3);
  const int ch = raw.info.channels;
  for (size_t i = 0; i < n; ++i) {
    const uint8_t* p = &raw.pixels[i * ch];
    if (ch >= 3) { (*rgb)[3 * i] = p[0]; (*rgb)[3 * i + 1] = p[1]; (*rgb)[3 * i + 2] = p[2]; }
    else { (*rgb)[3 * i] = (*rgb)[3 * i + 1] = (*rgb)[3 * i + 2] = p[0]; }
  }
  return true;
}

bool ReadPngGray16(const std::string& path, int* width, int* height, std::vector<uint16_t>* gray) {
  PngRaw raw;
  if (!decode_png(path, false, &raw) || raw.info.channels != 1) return false;
  *width = raw.info.width;
  *height = raw.info.height;
  const size_t n = static_cast<size_t>(raw.info.width) * raw.info.height;
  gray->resize(n);
  for (size_t i = 0; i < n; ++i)
    (*gray)[i] = raw.info.bit_depth == 16 ? static_cast<uint16_t>((raw.pixels[2 * i] << 8) | raw.pixels[2 * i + 1]) : raw.pixels[i];
  return true;
}

// ------------------------------------------------------------------------------------------------
// TUM RGB-D
// ------------------------------------------------------------------------------------------------
namespace {
// Eigen QuaternionBase::slerp (Eigen/src/Geometry/Quaternion.h): shortest arc, lerp fallback near 1
void slerp(const SE3f& a, const SE3f& b, float t, float* q) {
  const float one = 1.0f - std::numeric_limits<float>::epsilon();
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
  q[0] = scale0 * a.qx + scale1 * b.qx;
  q[1] = scale0 * a.qy + scale1 * b.qy;
  q[2] = scale0 * a.qz + scale1 * b.qz;
  q[3] = scale0 * a.qw + scale1 * b.qw;
}
}  // namespace

bool InterpolatePose(double timestamp, const std::vector<double>& ts, const std::vector<SE3f>& poses, SE3f* pose) {
  if (ts.size() != poses.size() || ts.size() < 2) return false;
  if (timestamp <= ts[0]) { *pose = poses[0]; return true; }
  if (timestamp >= ts.back()) { *pose = poses.back(); return true; }
  for (size_t i = 0; i + 1 < ts.size(); ++i) {
    if (timestamp >= ts[i] && timestamp <= ts[i + 1]) {
      const double factor = (timestamp - ts[i]) / (ts[i + 1] - ts[i]);
      const SE3f& a = poses[i];
      const SE3f& b = poses[i + 1];
      float q[4];
      slerp(a, b, static_cast<float>(factor), q);
      const float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);   // the SE3(quaternion, t) constructor normalises
      SE3f r;
      r.qx = q[0] / n; r.qy = q[1] / n; r.qz = q[2] / n; r.qw = q[3] / n;
      const float f = static_cast<float>(factor);
      r.tx = a.tx + f * (b.tx - a.tx);
      r.ty = a.ty + f * (b.ty - a.ty);
      r.tz = a.tz + f * (b.tz - a.tz);
      *pose = r;
      return true;
    }
  }
  return false;
}

bool ReadTUMRGBDTrajectory(const std::string& path, std::vector<double>* pose_timestamps, std::vector<SE3f>* poses) {
  std::ifstream file(path);
  if (!file) return false;
  std::string line;
  std::getline(file, line);
  while (!line.empty()) {   // the reference stops at the first empty line (:87)
    if (line[0] != '#') {
      char time_string[128];
      double t[3], q[4];
      if (std::sscanf(line.c_str(), "%127s %lf %lf %lf %lf %lf %lf %lf", time_string, &t[0], &t[1], &t[2], &q[0], &q[1], &q[2], &q[3]) != 8) return false;
      const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      SE3f T;
      // Quaterniond::cast<float>() then the normalising SE3 constructor
      const float fq[4] = {static_cast<float>(q[0]), static_cast<float>(q[1]), static_cast<float>(q[2]), static_cast<float>(q[3])};
      const float fn = std::sqrt(fq[0] * fq[0] + fq[1] * fq[1] + fq[2] * fq[2] + fq[3] * fq[3]);
      (void)n;
      T.qx = fq[0] / fn; T.qy = fq[1] / fn; T.qz = fq[2] / fn; T.qw = fq[3] / fn;
      T.tx = static_cast<float>(t[0]); T.ty = static_cast<float>(t[1]); T.tz = static_cast<float>(t[2]);
      pose_timestamps->push_back(std::atof(time_string));
      poses->push_back(T);
    }
    line.clear();
    if (!std::getline(file, line)) break;
  }
  return true;
}

bool ReadTUMRGBDDatasetAssociatedAndCalibrated(const std::string& folder, const std::string& trajectory_filename, TumDataset* ds) {
  ds->frames.clear();
  std::ifstream calibration_file(folder + "/calibration.txt");
  if (!calibration_file) return false;
  std::string line;
  std::getline(calibration_file, line);
  double fx, fy, cx, cy;
  if (std::sscanf(line.c_str(), "%lf %lf %lf %lf", &fx, &fy, &cx, &cy) != 4) return false;
  std::vector<double> pose_timestamps;
  std::vector<SE3f> poses;
  if (!trajectory_filename.empty() && !ReadTUMRGBDTrajectory(folder + "/" + trajectory_filename, &pose_timestamps, &poses)) return false;
  std::ifstream associated_file(folder + "/associated.txt");
  if (!associated_file) return false;
  ds->width = ds->height = 0;
  while (std::getline(associated_file, line)) {
    if (line.empty() || line[0] == '#') continue;
    char rgb_time[128], rgb_file[128], depth_time[128], depth_file[128];
    if (std::sscanf(line.c_str(), "%127s %127s %127s %127s", rgb_time, rgb_file, depth_time, depth_file) != 4) return false;
    TumFrame f;
    f.rgb_timestamp_string = rgb_time;
    f.depth_timestamp_string = depth_time;
    f.rgb_timestamp = std::atof(rgb_time);
    f.depth_timestamp = std::atof(depth_time);
    if (!poses.empty()) {   // frames outside an unusable trajectory are dropped (:183-195)
      if (!InterpolatePose(f.rgb_timestamp, pose_timestamps, poses, &f.rgb_global_T_frame)) continue;
      if (!InterpolatePose(f.depth_timestamp, pose_timestamps, poses, &f.depth_global_T_frame)) continue;
    }
    f.rgb_path = folder + "/" + rgb_file;
    f.depth_path = folder + "/" + depth_file;
    if (ds->width == 0) {   // the first colour image defines the size (:211-221)
      PngInfo info;
      if (!ReadPngInfo(f.rgb_path, &info)) return false;
      ds->width = info.width;
      ds->height = info.height;
    }
    ds->frames.push_back(f);
  }
  ds->camera_parameters[0] = static_cast<float>(fx);
  ds->camera_parameters[1] = static_cast<float>(fy);
  ds->camera_parameters[2] = static_cast<float>(cx + 0.5);
  ds->camera_parameters[3] = static_cast<float>(cy + 0.5);
  return true;
}

// ------------------------------------------------------------------------------------------------
// exports
// ------------------------------------------------------------------------------------------------
bool SavePoses(const std::vector<std::string>& timestamp_strings, const std::vector<SE3f>& global_T_frame, int start_frame, const std::string& path) {
  if (timestamp_strings.size() != global_T_frame.size() || start_frame < 0 || start_frame >= static_cast<int>(global_T_frame.size())) return false;
  const SE3f start_frame_T_global = global_T_frame[static_cast<size_t>(start_frame)].Inverse();
  std::ofstream file(path, std::ios::out);
  if (!file) return false;
  file << std::setprecision(std::numeric_limits<double>::digits10 + 1);
  file << "# Format: Each line gives one global_T_frame pose with values: tx ty tz qx qy qz qw" << std::endl;
  for (size_t i = 0; i < global_T_frame.size(); ++i) {
    const SE3f T = start_frame_T_global * global_T_frame[i];
    file << timestamp_strings[i] << " " << T.tx << " " << T.ty << " " << T.tz << " " << T.qx << " " << T.qy << " " << T.qz << " " << T.qw << std::endl;
  }
  return static_cast<bool>(file);
}

bool SaveCalibration(const std::string& base, const float depth_camera[4], const float color_camera[4], float a, int w, int h, const float* cfactor) {
  const float* cams[2] = {depth_camera, color_camera};
  const char* names[2] = {".depth_intrinsics.txt", ".color_intrinsics.txt"};
  for (int i = 0; i < 2; ++i) {
    std::ofstream file(base + names[i], std::ios::out);
    if (!file) return false;
    file << cams[i][0] << " " << cams[i][1] << " " << (cams[i][2] - 0.5) << " " << (cams[i][3] - 0.5);
  }
  std::ofstream file(base + ".deformation.txt", std::ios::out);
  if (!file) return false;
  file.precision(8);
  file << w << " " << h << std::endl;
  file << a << std::endl;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) file << cfactor[static_cast<size_t>(y) * w + x] << std::endl;
  return static_cast<bool>(file);
}

bool LoadCalibration(const std::string& base, float depth_camera[4], float color_camera[4], float* a, int w, int h, float* cfactor) {
  float* cams[2] = {depth_camera, color_camera};
  const char* names[2] = {".depth_intrinsics.txt", ".color_intrinsics.txt"};
  for (int i = 0; i < 2; ++i) {
    std::ifstream file(base + names[i], std::ios::in);
    if (!file) return false;
    file >> cams[i][0] >> cams[i][1] >> cams[i][2] >> cams[i][3];
    if (!file) return false;
    cams[i][2] += 0.5;
    cams[i][3] += 0.5;
  }
  std::ifstream file(base + ".deformation.txt", std::ios::in);
  if (!file) return false;
  int fw = 0, fh = 0;
  file >> fw >> fh;
  if (fw != w || fh != h) return false;   // "cfactor buffer size mismatch" (:676-679)
  file >> *a;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) file >> cfactor[static_cast<size_t>(y) * w + x];
  return static_cast<bool>(file);
}

}  // namespace bslam_host

// ------------------------------------------------------------------------------------------------
// state file v1 (BS/io.cc:38-536)
// ------------------------------------------------------------------------------------------------
namespace bslam_host {
namespace {

struct Writer {
  FILE* f;
  void i32(int32_t v) { std::fwrite(&v, 4, 1, f); }
  void u32(uint32_t v) { std::fwrite(&v, 4, 1, f); }
  void f32(float v) { std::fwrite(&v, 4, 1, f); }
  void b(bool v) { const uint8_t u = v ? 1 : 0; std::fwrite(&u, 1, 1, f); }
  void se3(const SE3f& T) { const float d[7] = {T.qx, T.qy, T.qz, T.qw, T.tx, T.ty, T.tz}; std::fwrite(d, 4, 7, f); }   // Sophus SE3::data(): quaternion coeffs, translation
  void str(const std::string& s) { u32(static_cast<uint32_t>(s.size())); std::fwrite(s.data(), 1, s.size(), f); }
};
struct Reader {
  FILE* f;
  bool ok = true;
  template <class T> T get() { T v{}; if (std::fread(&v, sizeof(T), 1, f) != 1) ok = false; return v; }
  int32_t i32() { return get<int32_t>(); }
  uint32_t u32() { return get<uint32_t>(); }
  float f32() { return get<float>(); }
  bool b() { return get<uint8_t>() != 0; }
  SE3f se3() {
    float d[7] = {0, 0, 0, 1, 0, 0, 0};
    if (std::fread(d, 4, 7, f) != 7) ok = false;
    SE3f T;
    T.qx = d[0]; T.qy = d[1]; T.qz = d[2]; T.qw = d[3]; T.tx = d[4]; T.ty = d[5]; T.tz = d[6];
    return T;
  }
  std::string str() {
    const uint32_t n = u32();
    if (!ok || n > (1u << 20)) { ok = false; return std::string(); }
    std::string s(n, '\0');
    if (n && std::fread(&s[0], 1, n, f) != n) ok = false;
    return s;
  }
};

void save_config(Writer& w, const BadSlamConfigV1& c) {   // BS/bad_slam_config.cc:47-96
  w.f32(c.raw_to_float_depth); w.i32(c.start_frame); w.i32(c.end_frame); w.f32(c.target_frame_rate); w.i32(c.fps_restriction);
  w.i32(c.pyramid_level_for_depth); w.i32(c.pyramid_level_for_color); w.f32(c.max_depth); w.f32(c.baseline_fx);
  w.i32(c.median_filter_and_densify_iterations); w.f32(c.bilateral_filter_sigma_xy); w.f32(c.bilateral_filter_radius_factor);
  w.f32(c.bilateral_filter_sigma_inv_depth); w.i32(c.max_surfel_count); w.i32(c.sparse_surfel_cell_size); w.f32(c.surfel_merge_dist_factor);
  w.i32(c.min_observation_count_while_bootstrapping_1); w.i32(c.min_observation_count_while_bootstrapping_2); w.i32(c.min_observation_count);
  w.i32(c.num_scales); w.b(c.use_motion_model); w.i32(c.keyframe_interval); w.i32(c.max_num_ba_iterations_per_keyframe);
  w.b(c.disable_deactivation); w.b(c.use_geometric_residuals); w.b(c.use_photometric_residuals); w.b(c.optimize_intrinsics);
  w.i32(c.intrinsics_optimization_interval); w.b(c.do_surfel_updates); w.b(c.parallel_ba); w.b(c.use_pcg); w.b(c.estimate_poses);
  w.i32(c.min_free_gpu_memory_mb); w.b(c.enable_loop_detection); w.b(c.parallel_loop_detection); w.str(c.loop_detection_vocabulary_path);
  w.str(c.loop_detection_pattern_path); w.f32(c.loop_detection_image_frequency); w.i32(c.loop_detection_images_width);
  w.i32(c.loop_detection_images_height);
}
void load_config(Reader& r, BadSlamConfigV1* c) {   // BS/bad_slam_config.cc:98-200
  c->raw_to_float_depth = r.f32(); c->start_frame = r.i32(); c->end_frame = r.i32(); c->target_frame_rate = r.f32(); c->fps_restriction = r.i32();
  c->pyramid_level_for_depth = r.i32(); c->pyramid_level_for_color = r.i32(); c->max_depth = r.f32(); c->baseline_fx = r.f32();
  c->median_filter_and_densify_iterations = r.i32(); c->bilateral_filter_sigma_xy = r.f32(); c->bilateral_filter_radius_factor = r.f32();
  c->bilateral_filter_sigma_inv_depth = r.f32(); c->max_surfel_count = r.i32(); c->sparse_surfel_cell_size = r.i32(); c->surfel_merge_dist_factor = r.f32();
  c->min_observation_count_while_bootstrapping_1 = r.i32(); c->min_observation_count_while_bootstrapping_2 = r.i32(); c->min_observation_count = r.i32();
  c->num_scales = r.i32(); c->use_motion_model = r.b(); c->keyframe_interval = r.i32(); c->max_num_ba_iterations_per_keyframe = r.i32();
  c->disable_deactivation = r.b(); c->use_geometric_residuals = r.b(); c->use_photometric_residuals = r.b(); c->optimize_intrinsics = r.b();
  c->intrinsics_optimization_interval = r.i32(); c->do_surfel_updates = r.b(); c->parallel_ba = r.b(); c->use_pcg = r.b(); c->estimate_poses = r.b();
  c->min_free_gpu_memory_mb = r.i32(); c->enable_loop_detection = r.b(); c->parallel_loop_detection = r.b(); c->loop_detection_vocabulary_path = r.str();
  c->loop_detection_pattern_path = r.str(); c->loop_detection_image_frequency = r.f32(); c->loop_detection_images_width = r.i32();
  c->loop_detection_images_height = r.i32();
}
constexpr int32_t kPinholeCamera4fType = 1;   // Camera::Type::kPinholeCamera4f (LV/camera.h:289)

}  // namespace

bool SaveState(const StateV1& s, const std::string& path) {
  if (s.motion_model_base_kf_tr_frame.size() > 1000 || s.queued_keyframes_frame_indices.size() != s.queued_keyframes_last_kf_tr_this_kf.size() ||
      s.cfactor.size() != static_cast<size_t>(s.cfactor_width) * s.cfactor_height || s.surfels.size() != static_cast<size_t>(8) * s.surfels_size)
    return false;
  FILE* file = std::fopen(path.c_str(), "wb");
  if (!file) return false;
  Writer w{file};
  std::fwrite("BADSLAM", 1, 7, file);
  const uint8_t version = 1;
  std::fwrite(&version, 1, 1, file);
  w.i32(s.base_kf_id);
  w.u32(static_cast<uint32_t>(s.motion_model_base_kf_tr_frame.size()));
  for (const SE3f& T : s.motion_model_base_kf_tr_frame) w.se3(T);
  w.u32(static_cast<uint32_t>(s.queued_keyframes_frame_indices.size()));
  for (size_t i = 0; i < s.queued_keyframes_frame_indices.size(); ++i) { w.i32(s.queued_keyframes_frame_indices[i]); w.se3(s.queued_keyframes_last_kf_tr_this_kf[i]); }
  w.i32(s.last_frame_index);
  save_config(w, s.config);
  w.u32(static_cast<uint32_t>(s.frame_global_T_frame.size()));
  for (const SE3f& T : s.frame_global_T_frame) w.se3(T);
  w.i32(kPinholeCamera4fType); w.i32(s.color_camera_width); w.i32(s.color_camera_height); w.i32(4);
  std::fwrite(s.color_camera_parameters, 4, 4, file);
  w.i32(s.pyramid_level_for_color);
  w.i32(kPinholeCamera4fType); w.i32(s.depth_camera_width); w.i32(s.depth_camera_height); w.i32(4);
  std::fwrite(s.depth_camera_parameters, 4, 4, file);
  w.i32(s.cfactor_width); w.i32(s.cfactor_height);
  w.i32(s.cfactor_width * static_cast<int32_t>(sizeof(float)));   // Image<float>::stride() in bytes; written unpadded
  std::fwrite(s.cfactor.data(), 4, s.cfactor.size(), file);
  w.f32(s.a); w.f32(s.raw_to_float_depth); w.f32(s.baseline_fx); w.i32(s.sparse_surfel_cell_size);
  w.i32(static_cast<int32_t>(s.keyframes.size()));
  for (const StateKeyframeV1& k : s.keyframes) {
    w.i32(k.id);
    if (k.id >= 0) { w.i32(k.frame_index); w.i32(k.activation); w.i32(k.last_active_in_ba_iteration); w.i32(k.last_covis_in_ba_iteration); }
  }
  w.i32(s.surfel_count); w.i32(s.surfels_size);
  std::fwrite(s.surfels.data(), 4, s.surfels.size(), file);
  w.i32(s.ba_iteration_count); w.i32(s.last_ba_iteration_count);
  w.b(s.use_depth_residuals); w.b(s.use_descriptor_residuals);
  w.i32(s.min_observation_count_while_bootstrapping_1); w.i32(s.min_observation_count_while_bootstrapping_2); w.i32(s.min_observation_count);
  w.f32(s.surfel_merge_dist_factor);
  const bool ok = !std::ferror(file);
  std::fclose(file);
  return ok;
}

bool LoadState(const std::string& path, StateV1* s, std::string* error) {
  auto fail = [&](const char* msg) { if (error) *error = msg; return false; };
  FILE* file = std::fopen(path.c_str(), "rb");
  if (!file) return fail("cannot open the state file");
  struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{file};
  char id[7];
  if (std::fread(id, 1, 7, file) != 7 || std::memcmp(id, "BADSLAM", 7) != 0) return fail("File identifier does not match.");
  uint8_t version = 0;
  if (std::fread(&version, 1, 1, file) != 1 || version != 1) return fail("Unknown file format version.");
  Reader r{file};
  s->base_kf_id = r.i32();
  uint32_t n = r.u32();
  if (!r.ok || n > 1000) return fail("Unexpected motion model size.");          // BS/io.cc:263-266
  s->motion_model_base_kf_tr_frame.resize(n);
  for (uint32_t i = 0; i < n; ++i) s->motion_model_base_kf_tr_frame[i] = r.se3();
  n = r.u32();
  if (!r.ok || n > 10000) return fail("Unexpected queued keyframe count.");     // :273-276
  s->queued_keyframes_frame_indices.resize(n);
  s->queued_keyframes_last_kf_tr_this_kf.resize(n);
  for (uint32_t i = 0; i < n; ++i) { s->queued_keyframes_frame_indices[i] = r.i32(); s->queued_keyframes_last_kf_tr_this_kf[i] = r.se3(); }
  s->last_frame_index = r.i32();
  load_config(r, &s->config);
  n = r.u32();
  if (!r.ok || n > (1u << 24)) return fail("Unexpected frame count.");
  s->frame_global_T_frame.resize(n);
  for (uint32_t i = 0; i < n; ++i) s->frame_global_T_frame[i] = r.se3();
  for (int cam = 0; cam < 2; ++cam) {
    const int32_t type = r.i32(), w = r.i32(), h = r.i32(), count = r.i32();
    if (!r.ok || type != kPinholeCamera4fType || count != 4) return fail("Only PinholeCamera4f cameras are supported.");   // :311-316
    float* p = cam == 0 ? s->color_camera_parameters : s->depth_camera_parameters;
    if (std::fread(p, 4, 4, file) != 4) return fail("Unexpected end of file.");
    if (cam == 0) { s->color_camera_width = w; s->color_camera_height = h; s->pyramid_level_for_color = r.i32(); }
    else { s->depth_camera_width = w; s->depth_camera_height = h; }
  }
  s->cfactor_width = r.i32();
  s->cfactor_height = r.i32();
  const int32_t stride = r.i32();
  if (!r.ok || s->cfactor_width <= 0 || s->cfactor_height <= 0 || stride < s->cfactor_width * 4 || stride > 100000 * 4) return fail("Unexpected cfactor buffer size.");
  s->cfactor.assign(static_cast<size_t>(s->cfactor_width) * s->cfactor_height, 0.f);
  std::vector<uint8_t> row(static_cast<size_t>(stride));
  for (int y = 0; y < s->cfactor_height; ++y) {
    if (std::fread(row.data(), 1, row.size(), file) != row.size()) return fail("Unexpected end of file.");
    std::memcpy(&s->cfactor[static_cast<size_t>(y) * s->cfactor_width], row.data(), static_cast<size_t>(s->cfactor_width) * 4);
  }
  s->a = r.f32(); s->raw_to_float_depth = r.f32(); s->baseline_fx = r.f32(); s->sparse_surfel_cell_size = r.i32();
  const int32_t kf_count = r.i32();
  if (!r.ok || kf_count < 0 || static_cast<size_t>(kf_count) > s->frame_global_T_frame.size()) return fail("Unexpected keyframe count.");   // :359-362
  s->keyframes.assign(static_cast<size_t>(kf_count), StateKeyframeV1());
  for (int32_t i = 0; i < kf_count; ++i) {
    StateKeyframeV1& k = s->keyframes[static_cast<size_t>(i)];
    k.id = r.i32();
    if (k.id >= 0) {
      if (k.id != i) return fail("Keyframe ids must equal their position.");     // :384-387
      k.frame_index = r.i32(); k.activation = r.i32(); k.last_active_in_ba_iteration = r.i32(); k.last_covis_in_ba_iteration = r.i32();
    }
  }
  s->surfel_count = r.i32();
  s->surfels_size = r.i32();
  if (!r.ok || s->surfels_size < 0 || s->surfel_count < 0 || s->surfel_count > s->surfels_size) return fail("Unexpected surfel counts.");
  s->surfels.resize(static_cast<size_t>(8) * s->surfels_size);
  if (!s->surfels.empty() && std::fread(s->surfels.data(), 4, s->surfels.size(), file) != s->surfels.size()) return fail("Unexpected end of file.");
  s->ba_iteration_count = r.i32(); s->last_ba_iteration_count = r.i32();
  s->use_depth_residuals = r.b(); s->use_descriptor_residuals = r.b();
  s->min_observation_count_while_bootstrapping_1 = r.i32(); s->min_observation_count_while_bootstrapping_2 = r.i32(); s->min_observation_count = r.i32();
  s->surfel_merge_dist_factor = r.f32();
  if (!r.ok) return fail("Unexpected end of file.");
  return true;
}

}  // namespace bslam_host
