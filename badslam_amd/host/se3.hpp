// se3.hpp -- host-side rigid transforms with the semantics of Sophus::SE3f as the reference's
// DirectBA uses it (libvis/third_party/sophus/sophus/{so3,se3}.hpp): unit quaternion +
// translation in fp32, tangent = [translation(3), rotation(3)], first-order re-normalisation
// after products.  Eigen is not available in this image, so the handful of Eigen quaternion
// routines Sophus relies on are written out.
#pragma once

#include <cmath>
#include <utility>

#include "../../include/badslam_hip.h"

namespace bslam_host {

struct Vec3f { float x, y, z; };

struct SE3f {
  // quaternion (x, y, z, w) and translation
  float qx = 0.f, qy = 0.f, qz = 0.f, qw = 1.f;
  float tx = 0.f, ty = 0.f, tz = 0.f;

  static constexpr float kEpsilon = 1e-5f;   // Sophus::Constants<float>::epsilon()

  static SE3f FromPod(const bslam_se3f& p) {
    SE3f T;
    T.qx = p.q[0]; T.qy = p.q[1]; T.qz = p.q[2]; T.qw = p.q[3];
    T.tx = p.t[0]; T.ty = p.t[1]; T.tz = p.t[2];
    return T;
  }
  bslam_se3f ToPod() const {
    bslam_se3f p;
    p.q[0] = qx; p.q[1] = qy; p.q[2] = qz; p.q[3] = qw;
    p.t[0] = tx; p.t[1] = ty; p.t[2] = tz;
    return p;
  }

  // Eigen QuaternionBase::_transformVector
  Vec3f Rotate(Vec3f v) const {
    Vec3f uv{qy * v.z - qz * v.y, qz * v.x - qx * v.z, qx * v.y - qy * v.x};
    uv = Vec3f{uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
    const Vec3f c{qy * uv.z - qz * uv.y, qz * uv.x - qx * uv.z, qx * uv.y - qy * uv.x};
    return Vec3f{v.x + qw * uv.x + c.x, v.y + qw * uv.y + c.y, v.z + qw * uv.z + c.z};
  }

  // Eigen QuaternionBase::toRotationMatrix, row-major 3x3
  void RotationMatrix(float* R) const {
    const float tx2 = 2.0f * qx, ty2 = 2.0f * qy, tz2 = 2.0f * qz;
    const float twx = tx2 * qw, twy = ty2 * qw, twz = tz2 * qw;
    const float txx = tx2 * qx, txy = ty2 * qx, txz = tz2 * qx;
    const float tyy = ty2 * qy, tyz = tz2 * qy, tzz = tz2 * qz;
    R[0] = 1.0f - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
    R[3] = txy + twz;          R[4] = 1.0f - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = 1.0f - (txx + tyy);
  }

  bslam_mat3x4 Matrix3x4() const {
    float R[9];
    RotationMatrix(R);
    bslam_mat3x4 M;
    const float t[3] = {tx, ty, tz};
    for (int r = 0; r < 3; ++r) {
      M.m[4 * r + 0] = R[3 * r + 0]; M.m[4 * r + 1] = R[3 * r + 1]; M.m[4 * r + 2] = R[3 * r + 2];
      M.m[4 * r + 3] = t[r];
    }
    return M;
  }
  bslam_mat3x3 Rotation3x3() const {
    bslam_mat3x3 M;
    RotationMatrix(M.m);
    return M;
  }

  // SE3Base::inverse
  SE3f Inverse() const {
    SE3f I;
    I.qx = -qx; I.qy = -qy; I.qz = -qz; I.qw = qw;
    const Vec3f t = I.Rotate(Vec3f{tx * -1.f, ty * -1.f, tz * -1.f});
    I.tx = t.x; I.ty = t.y; I.tz = t.z;
    return I;
  }

  // SE3Base::operator*  (translation += so3 * other.translation; so3 *= other.so3)
  SE3f operator*(const SE3f& o) const {
    SE3f r;
    const Vec3f rt = Rotate(Vec3f{o.tx, o.ty, o.tz});
    r.tx = tx + rt.x; r.ty = ty + rt.y; r.tz = tz + rt.z;
    r.qw = qw * o.qw - qx * o.qx - qy * o.qy - qz * o.qz;
    r.qx = qw * o.qx + qx * o.qw + qy * o.qz - qz * o.qy;
    r.qy = qw * o.qy + qy * o.qw + qz * o.qx - qx * o.qz;
    r.qz = qw * o.qz + qz * o.qw + qx * o.qy - qy * o.qx;
    const float sn = r.qx * r.qx + r.qy * r.qy + r.qz * r.qz + r.qw * r.qw;
    if (sn != 1.0f) {   // SO3Base::operator*= so3.hpp:215-232
      const float f = 2.0f / (1.0f + sn);
      r.qx *= f; r.qy *= f; r.qz *= f; r.qw *= f;
    }
    return r;
  }

  // SE3::exp se3.hpp:293-313 with SO3::expAndTheta so3.hpp:282-318
  static SE3f Exp(const float* a) {
    const float ox = a[3], oy = a[4], oz = a[5];
    const float theta_sq = ox * ox + oy * oy + oz * oz;
    const float theta = std::sqrt(theta_sq);
    const float half_theta = 0.5f * theta;
    float imag_factor, real_factor;
    if (theta < kEpsilon) {
      const float theta_po4 = theta_sq * theta_sq;
      imag_factor = 0.5f - static_cast<float>(1.0 / 48.0) * theta_sq + static_cast<float>(1.0 / 3840.0) * theta_po4;
      real_factor = 1.f - 0.5f * theta_sq + static_cast<float>(1.0 / 384.0) * theta_po4;
    } else {
      imag_factor = std::sin(half_theta) / theta;
      real_factor = std::cos(half_theta);
    }
    SE3f T;
    T.qx = imag_factor * ox; T.qy = imag_factor * oy; T.qz = imag_factor * oz; T.qw = real_factor;
    const float O[9] = {0.f, -oz, oy, oz, 0.f, -ox, -oy, ox, 0.f};
    float O2[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) O2[3 * r + c] = O[3 * r + 0] * O[0 + c] + O[3 * r + 1] * O[3 + c] + O[3 * r + 2] * O[6 + c];
    float V[9];
    if (theta < kEpsilon) {
      T.RotationMatrix(V);
    } else {
      const float c1 = (1.f - std::cos(theta)) / (theta_sq);
      const float c2 = (theta - std::sin(theta)) / (theta_sq * theta);
      for (int i = 0; i < 9; ++i) {
        const float id = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
        V[i] = (id + c1 * O[i]) + c2 * O2[i];
      }
    }
    T.tx = V[0] * a[0] + V[1] * a[1] + V[2] * a[2];
    T.ty = V[3] * a[0] + V[4] * a[1] + V[5] * a[2];
    T.tz = V[6] * a[0] + V[7] * a[1] + V[8] * a[2];
    return T;
  }

  // SE3::log se3.hpp:435-466 with SO3::logAndTheta so3.hpp:421-466
  void Log(float* out) const {
    const float squared_n = qx * qx + qy * qy + qz * qz;
    const float n = std::sqrt(squared_n);
    float two_atan_nbyw_by_n;
    if (n < kEpsilon) {
      const float squared_w = qw * qw;
      two_atan_nbyw_by_n = 2.f / qw - 2.f * (squared_n) / (qw * squared_w);
    } else if (std::fabs(qw) < kEpsilon) {
      two_atan_nbyw_by_n = (qw > 0.f ? 1.f : -1.f) * static_cast<float>(M_PI) / n;
    } else {
      two_atan_nbyw_by_n = 2.f * std::atan(n / qw) / n;
    }
    const float theta = two_atan_nbyw_by_n * n;
    const float ox = two_atan_nbyw_by_n * qx, oy = two_atan_nbyw_by_n * qy, oz = two_atan_nbyw_by_n * qz;
    const float O[9] = {0.f, -oz, oy, oz, 0.f, -ox, -oy, ox, 0.f};
    float O2[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) O2[3 * r + c] = O[3 * r + 0] * O[0 + c] + O[3 * r + 1] * O[3 + c] + O[3 * r + 2] * O[6 + c];
    float k;
    if (std::fabs(theta) < kEpsilon) {
      k = static_cast<float>(1. / 12.);
    } else {
      const float half_theta = 0.5f * theta;
      k = (1.f - theta * std::cos(half_theta) / (2.f * std::sin(half_theta))) / (theta * theta);
    }
    float Vi[9];
    for (int i = 0; i < 9; ++i) {
      const float id = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
      Vi[i] = (id - 0.5f * O[i]) + k * O2[i];
    }
    out[0] = Vi[0] * tx + Vi[1] * ty + Vi[2] * tz;
    out[1] = Vi[3] * tx + Vi[4] * ty + Vi[5] * tz;
    out[2] = Vi[6] * tx + Vi[7] * ty + Vi[8] * tz;
    out[3] = ox; out[4] = oy; out[5] = oz;
  }
};

// IsScale1PoseEstimationConverged (BS/convergence_analysis.h:45-52)
inline bool IsScale1PoseEstimationConverged(const float* x) {
  constexpr float translation_threshold = 1e-06f;
  constexpr float rotation_threshold = 1e-07f;
  const float s = translation_threshold / rotation_threshold;
  float n = 0.f;
  for (int i = 0; i < 6; ++i) {
    const float v = (i < 3) ? x[i] : x[i] * s;
    n += v * v;
  }
  return n < translation_threshold;
}

// x = H^-1 b for the packed upper triangle H (row-major) in double: LDL^T with diagonal pivoting,
// what Eigen's selfadjointView<Upper>().ldlt().solve() computes (BS/direct_ba_alternating.cc:206).
inline void SolveLDLTUpper(int n, const float* H_upper, const float* b, float* x) {
  double A[36];
  int idx = 0;
  for (int r = 0; r < n; ++r)
    for (int c = r; c < n; ++c) { A[r * n + c] = H_upper[idx]; A[c * n + r] = H_upper[idx]; ++idx; }
  int perm[6];
  for (int k = 0; k < n; ++k) {
    int p = k;
    double biggest = std::fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i) { const double v = std::fabs(A[i * n + i]); if (v > biggest) { biggest = v; p = i; } }
    perm[k] = p;
    if (p != k) {
      for (int c = 0; c < n; ++c) std::swap(A[k * n + c], A[p * n + c]);
      for (int r = 0; r < n; ++r) std::swap(A[r * n + k], A[r * n + p]);
    }
    double temp[6];
    for (int j = 0; j < k; ++j) temp[j] = A[j * n + j] * A[k * n + j];
    double acc = 0.0;
    for (int j = 0; j < k; ++j) acc += A[k * n + j] * temp[j];
    const double dk = A[k * n + k] - acc;
    A[k * n + k] = dk;
    for (int i = k + 1; i < n; ++i) {
      double s = 0.0;
      for (int j = 0; j < k; ++j) s += A[i * n + j] * temp[j];
      const double v = A[i * n + k] - s;
      A[i * n + k] = (std::fabs(dk) > 0.0) ? v / dk : 0.0;
    }
  }
  double y[6];
  for (int i = 0; i < n; ++i) y[i] = b[i];
  for (int k = 0; k < n; ++k) if (perm[k] != k) std::swap(y[k], y[perm[k]]);
  for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) y[i] -= A[i * n + j] * y[j];
  for (int i = 0; i < n; ++i) { const double d = A[i * n + i]; y[i] = (std::fabs(d) > 2.2250738585072014e-308) ? y[i] / d : 0.0; }
  for (int i = n - 1; i >= 0; --i) for (int j = i + 1; j < n; ++j) y[i] -= A[j * n + i] * y[j];
  for (int k = n - 1; k >= 0; --k) if (perm[k] != k) std::swap(y[k], y[perm[k]]);
  for (int i = 0; i < n; ++i) x[i] = static_cast<float>(y[i]);
}

}  // namespace bslam_host
