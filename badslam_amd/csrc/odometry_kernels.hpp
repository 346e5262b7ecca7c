// odometry_kernels.hpp -- pairwise frame tracking for gfx950: pyramid construction
// (BS/kernel_downsample.cu) and the image-pair Gauss-Newton coefficients / cost kernels of
// BS/kernel_opt_pose.cu:422-661, 939-1171 (GradientXY variant, the one BadSlam::RunOdometry uses).
// Naming follows the reference: "surfel" images = the BASE frame (whose pixels are projected), "frame" images = the
// TRACKED frame; estimate_frame_T_surfel_frame = (base_T_frame)^-1.
// The reduction is the pose kernel's: 32 accumulators per lane -> transposed wave butterfly -> one row per wave
// -> pose_reduce_kernel / pose_reduce_final_kernel (deterministic, no atomics).
#pragma once

#include "pose_kernels.hpp"
#include "preprocess_kernels.hpp"

namespace bslam {

// bilinear fetch from a single-channel u8 image (texture model of tex_filter / tex_footprint)
__device__ __forceinline__ float tex_u8_direct(const Img& img, float x, float y, int tex_mode) {
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fx = floorf(xb), fy = floorf(yb);
  float a = xb - fx, b = yb - fy;
  if (tex_mode == BSLAM_TEX_FIXED_POINT_1_8) {
    a = floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f);
    b = floorf(b * 256.0f + 0.5f) * (1.0f / 256.0f);
  }
  const int i = (int)fminf(fmaxf(fx, -1.0f), (float)(img.width - 1));
  const int j = (int)fminf(fmaxf(fy, -1.0f), (float)(img.height - 1));
  auto texel = [&](int ix, int iy) {
    ix = max(0, min(ix, img.width - 1));
    iy = max(0, min(iy, img.height - 1));
    return (float)img.at<uint8_t>(iy, ix) * (1.0f / 255.0f);
  };
  LumaQuad q;
  q.tl = texel(i, j); q.tr = texel(i + 1, j); q.bl = texel(i, j + 1); q.br = texel(i + 1, j + 1);
  return tex_filter(q, a, b);
}
__device__ __forceinline__ uint8_t sat_u8(float v) { return (uint8_t)min(255, max(0, f2i(v))); }   // cvt.rzi.u8.f32

// ComputeBrightnessKernel(texture) BS/cuda_image_processing.cu:196-205
__global__ __launch_bounds__(256) void brightness_from_color_kernel(Img color, Img out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= out.width || y >= out.height) return;
  const uint32_t luma = color.at<uint32_t>(y, x) >> 24;
  out.at<uint8_t>(y, x) = sat_u8(255.f * ((float)luma * (1.0f / 255.0f)));
}
// CUDABuffer_<u8>::SetToReadModeNormalized LV/cuda/cuda_buffer.cu:82-102
__global__ __launch_bounds__(256) void read_mode_normalized_kernel(Img in, Img out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= out.width || y >= out.height) return;
  out.at<uint8_t>(y, x) = sat_u8(255.f * ((float)in.at<uint8_t>(y, x) * (1.0f / 255.0f)));
}
// CalibrateDepthCUDAKernel / CalibrateDepthAndTransformColorToDepthCUDAKernel BS/kernel_downsample.cu:236-312
template <bool kTransformColor>
__global__ __launch_bounds__(256) void calibrate_depth_kernel(CamConsts c, Img depth_u16, Img color_u8, Img out_depth, Img out_color) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= out_depth.width || y >= out_depth.height) return;
  const uint32_t raw = depth_u16.at<uint16_t>(y, x);
  float depth = 0.f;
  if (!(raw & BSLAM_INVALID_DEPTH_BIT)) {
    const float cf = *(const float*)((const uint8_t*)c.cfactor + (size_t)(y / c.cell) * c.cfactor_pitch + 4 * (size_t)(x / c.cell));
    depth = raw_to_calibrated_depth(c.a, cf, c.raw_to_float_depth, raw);
  }
  if (kTransformColor) {
    f2 color_pxy;
    const bool in_bounds = depth_to_color_pxy(c, f2{x + 0.5f, y + 0.5f}, &color_pxy);
    out_depth.at<float>(y, x) = in_bounds ? depth : 0.f;
    out_color.at<uint8_t>(y, x) = sat_u8(255.f * tex_u8_direct(color_u8, color_pxy.x, color_pxy.y, c.tex_mode) + 0.5f);
  } else {
    out_depth.at<float>(y, x) = depth;
  }
}
// DownsampleImagesCUDAKernel BS/kernel_downsample.cu:105-152
__global__ __launch_bounds__(256) void downsample_kernel(Img depth, Img normals, Img color, int tex_mode, Img out_depth, Img out_normals, Img out_color) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= out_depth.width || y >= out_depth.height) return;
  float depths[4], depth_sum = 0.f;
  int depth_count = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    depths[i] = depth.at<float>(2 * y + (i >> 1), 2 * x + (i & 1));
    if (depths[i] > 0) { depth_sum += depths[i]; depth_count += 1; }
    else depths[i] = __uint_as_float(0x7f800000u);
  }
  if (depth_count == 0) {
    out_depth.at<float>(y, x) = 0.f;
  } else {
    const float average_depth = depth_sum / (float)depth_count;
    int closest_index = 0;
    float closest_distance = __uint_as_float(0x7f800000u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float distance = fabsf(depths[i] - average_depth);
      if (distance < closest_distance) { closest_index = i; closest_distance = distance; }
    }
    out_depth.at<float>(y, x) = depths[closest_index];
    out_normals.at<uint16_t>(y, x) = normals.at<uint16_t>(2 * y + (closest_index >> 1), 2 * x + (closest_index & 1));
  }
  out_color.at<uint8_t>(y, x) = sat_u8(255.f * tex_u8_direct(color, 2 * x + 1.0f, 2 * y + 1.0f, tex_mode) + 0.5f);
}
// ComputeSobelGradientMagnitudeKernel(texture, gradmag) BS/cuda_image_processing.cu:104-143: Sobel magnitude of the luma channel
// (.w of the uchar4 image, clamp addressing), normalised to 0..255 and truncated
__global__ __launch_bounds__(256) void sobel_gradient_magnitude_kernel(Img color, Img out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= out.width || y >= out.height) return;
  float I[3][3];
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int ix = max(0, min(x + dx, color.width - 1)), iy = max(0, min(y + dy, color.height - 1));
      I[dy + 1][dx + 1] = 255.f * ((float)(color.at<uint32_t>(iy, ix) >> 24) * (1.0f / 255.0f));   // 255 * tex2D<float4>(...).w at a texel centre
    }
  const float gx = 1 * I[0][2] - 1 * I[0][0] + 2 * I[1][2] - 2 * I[1][0] + 1 * I[2][2] - 1 * I[2][0];
  const float gy = 1 * I[2][0] - 1 * I[0][0] + 2 * I[2][1] - 2 * I[0][1] + 1 * I[2][2] - 1 * I[0][2];
  constexpr float kNormalizer = 255.99f / (1.41421356237309504880f * 4 * 255.f);
  out.at<uint8_t>(y, x) = sat_u8(kNormalizer * sqrtf(gx * gx + gy * gy));
}
// CalibrateAndDownsampleImagesCUDAKernel<downsample_color> BS/kernel_downsample.cu:40-105: first pyramid step of a tracked frame whose
// level 0 is not used (the cfactor cell is looked up with the DOWNSAMPLED pixel coordinates, as in the reference, :65-66)
template <bool kDownsampleColor>
__global__ __launch_bounds__(256) void calibrate_and_downsample_kernel(CamConsts c, Img depth_u16, Img normals, Img color, Img out_depth, Img out_normals, Img out_color) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= out_depth.width || y >= out_depth.height) return;
  float depths[4], depth_sum = 0.f;
  int depth_count = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t raw = depth_u16.at<uint16_t>(2 * y + (i >> 1), 2 * x + (i & 1));
    if (!(raw & BSLAM_INVALID_DEPTH_BIT)) {
      const float cf = *(const float*)((const uint8_t*)c.cfactor + (size_t)(y / c.cell) * c.cfactor_pitch + 4 * (size_t)(x / c.cell));
      depths[i] = raw_to_calibrated_depth(c.a, cf, c.raw_to_float_depth, raw);
      depth_sum += depths[i];
      depth_count += 1;
    } else {
      depths[i] = __uint_as_float(0x7f800000u);
    }
  }
  if (depth_count == 0) {
    out_depth.at<float>(y, x) = 0.f;
  } else {
    const float average_depth = depth_sum / (float)depth_count;
    int closest_index = 0;
    float closest_distance = __uint_as_float(0x7f800000u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float distance = fabsf(depths[i] - average_depth);
      if (distance < closest_distance) { closest_index = i; closest_distance = distance; }
    }
    out_depth.at<float>(y, x) = depths[closest_index];
    out_normals.at<uint16_t>(y, x) = normals.at<uint16_t>(2 * y + (closest_index >> 1), 2 * x + (closest_index & 1));
  }
  const float col = kDownsampleColor ? tex_u8_direct(color, 2 * x + 1.0f, 2 * y + 1.0f, c.tex_mode) : tex_u8_direct(color, x + 0.5f, y + 0.5f, c.tex_mode);
  out_color.at<uint8_t>(y, x) = sat_u8(255.f * col + 0.5f);
}
// luma quads of a u8 image (see KfDev::quads)
__global__ __launch_bounds__(256) void build_quads_u8_kernel(Img img, uint32_t* __restrict__ quads) {
  const int qx = blockIdx.x * blockDim.x + threadIdx.x, qy = blockIdx.y;
  const int w = img.width, h = img.height;
  if (qx > w) return;
  const int i0 = max(0, qx - 1), i1 = min(qx, w - 1), j0 = max(0, qy - 1), j1 = min(qy, h - 1);
  const uint32_t tl = img.at<uint8_t>(j0, i0), tr = img.at<uint8_t>(j0, i1), bl = img.at<uint8_t>(j1, i0), br = img.at<uint8_t>(j1, i1);
  quads[(size_t)qy * (size_t)(w + 1) + qx] = tl | (tr << 8) | (bl << 16) | (br << 24);
}

struct PairImages {
  Img surfel_depth, surfel_normals, surfel_color;   // base frame: f32 depth, u16 normals, u8 colour (depth intrinsics)
  Img frame_depth, frame_normals;                   // tracked frame: f32 depth, u16 normals
  const uint32_t* frame_quads;                      // tracked frame colour as luma quads (colour intrinsics)
  Img frame_color;                                  // the same image, direct (the gradient-magnitude variant reads it as the reference does)
};

// One thread per base pixel.  kCoeffs: H (21) + b (6) of AccumulatePoseEstimationCoeffsFromImagesCUDAKernel_GradientXY
// (column 28 = number of visible pixels); !kCoeffs: cost (column 27) and residual count (column 28) of
// ComputeCostAndResidualCountFromImagesCUDAKernel_GradientXY.
// kGradMag: ONE colour residual on gradient-magnitude images (ComputeRawColorResidualAndJacobian BS/kernel_opt_pose.cu:192-222,
// kernels :713-937 and :1173-1338) instead of the two descriptor residuals.
template <bool kDepth, bool kDesc, bool kCoeffs, bool kGradMag = false>
__global__ __launch_bounds__(256) void pair_accumulate_kernel(CamConsts c, M34 T, float threshold_factor, PairImages im, float* __restrict__ partials) {
  const int pixel = blockIdx.x * blockDim.x + threadIdx.x;
  const int w = im.surfel_depth.width, h = im.surfel_depth.height;
  float acc[kRow];
#pragma unroll
  for (int i = 0; i < kRow; ++i) acc[i] = 0.f;
  uint32_t count = 0;
  if (pixel < w * h) {
    const int y = pixel / w, x = pixel - y * w;
    const float surfel_depth = im.surfel_depth.at<float>(y, x);
    bool visible = false;
    f3 local = mk3(0, 0, 0), n_local = mk3(0, 0, 0);
    f2 pxy = f2{0, 0};
    int px = 0, py = 0;
    float pixel_depth = 0.f;
    if (surfel_depth > 0) {
      const f3 p = unproject(c, x, y, surfel_depth);
      local.z = tr_row(T.m[8], T.m[9], T.m[10], T.m[11], p);
      if (local.z > 0.f) {
        local.x = tr_row(T.m[0], T.m[1], T.m[2], T.m[3], p);
        local.y = tr_row(T.m[4], T.m[5], T.m[6], T.m[7], p);
        pxy = project(c.fx, c.fy, c.cx, c.cy, local);
        px = f2i(pxy.x); py = f2i(pxy.y);
        if (!(pxy.x < 0 || pxy.y < 0 || px >= c.width || py >= c.height)) {
          pixel_depth = im.frame_depth.at<float>(py, px);
          if (pixel_depth > 0) {
            // IsAssociatedWithPixel<false>(… image normals …) BS/surfel_projection_nvcc_only.cuh:168-215
            n_local = rot34(T, u16_to_image_space_normal(im.surfel_normals.at<uint16_t>(y, x)));
            const float stddev = depth_stddev(nx_of(c, (float)px), ny_of(c, (float)py), pixel_depth, n_local, c.inv_baseline_fx);
            visible = !(fabsf(local.z - pixel_depth) > (threshold_factor * kDepthTukey) * stddev) && !(dot(local, n_local) > 0) &&
                      !(dot(n_local, u16_to_image_space_normal(im.frame_normals.at<uint16_t>(py, px))) < kCosNormalCompat);
          }
        }
      }
    }
    float raw_depth = 0.f, Jd[6], r1 = 0.f, r2 = 0.f, J1[6], J2[6];
    if (visible && kDepth) {
      const float inv_stddev = depth_inv_stddev(nx_of(c, (float)px), ny_of(c, (float)py), pixel_depth, n_local, c.baseline_fx);
      const f3 lu = unproject(c, px, py, pixel_depth);
      raw_depth = depth_residual(inv_stddev, n_local, lu, local);
      depth_pose_jacobian(inv_stddev, n_local, lu, Jd);
    }
    if (visible && kDesc && kGradMag) {
      f2 cp;
      if (depth_to_color_pxy(c, pxy, &cp)) {
        r1 = 255.f * tex_u8_direct(im.frame_color, cp.x, cp.y, c.tex_mode) - (float)im.surfel_color.at<uint8_t>(y, x);   // BS/cost_function.cuh:324-331
        if (kCoeffs) {                                                                                                    // :335-352
          const GradFootprint g = grad_footprint(c, cp);
          auto texel = [&](int ix, int iy) {
            ix = max(0, min(ix, im.frame_color.width - 1));
            iy = max(0, min(iy, im.frame_color.height - 1));
            return 255.f * ((float)im.frame_color.at<uint8_t>(iy, ix) * (1.0f / 255.0f));
          };
          const LumaQuad q{texel(g.ix, g.iy), texel(g.ix + 1, g.iy), texel(g.ix, g.iy + 1), texel(g.ix + 1, g.iy + 1)};
          float gx, gy;
          grad_filter(q, g, &gx, &gy);
          gx *= c.cfx;
          gy *= c.cfy;
          descriptor_pose_jacobian<true>(gx, gy, local, J1);
        }
      } else {
        visible = false;
      }
    } else if (visible && kDesc) {
      if (x < w - 1 && y < h - 1) {
        const float intensity = 1 / 255.f * (float)im.surfel_color.at<uint8_t>(y, x);
        const float t1_intensity = 1 / 255.f * (float)im.surfel_color.at<uint8_t>(y, x + 1);
        const float t2_intensity = 1 / 255.f * (float)im.surfel_color.at<uint8_t>(y + 1, x);
        const float d1 = (180.f * (t1_intensity - intensity)), d2 = (180.f * (t2_intensity - intensity));
        const f3 sn = u16_to_image_space_normal(im.surfel_normals.at<uint16_t>(y, x));
        const float plane_d = (nx_of(c, (float)x) * surfel_depth) * sn.x + (ny_of(c, (float)y) * surfel_depth) * sn.y + surfel_depth * sn.z;
        const float x1_depth = plane_d / (nx_of(c, (float)(x + 1)) * sn.x + ny_of(c, (float)y) * sn.y + sn.z);
        const f3 x1_local = mul34(T, unproject(c, x + 1, y, x1_depth));
        const f2 pxy_t1 = project(c.fx, c.fy, c.cx, c.cy, x1_local);
        if (pxy_t1.x < 0 || pxy_t1.y < 0 || f2i(pxy_t1.x) >= c.width || f2i(pxy_t1.y) >= c.height) visible = false;
        const float y1_depth = plane_d / (nx_of(c, (float)x) * sn.x + ny_of(c, (float)(y + 1)) * sn.y + sn.z);
        const f3 y1_local = mul34(T, unproject(c, x, y + 1, y1_depth));
        const f2 pxy_t2 = project(c.fx, c.fy, c.cx, c.cy, y1_local);
        if (pxy_t2.x < 0 || pxy_t2.y < 0 || f2i(pxy_t2.x) >= c.width || f2i(pxy_t2.y) >= c.height) visible = false;
        f2 c0, c1, c2;
        if (visible && x1_local.z > 0 && y1_local.z > 0 && depth_to_color_pxy(c, pxy, &c0) && depth_to_color_pxy(c, pxy_t1, &c1) && depth_to_color_pxy(c, pxy_t2, &c2)) {
          KfDev kf;
          kf.quads = im.frame_quads;
          float gx1, gy1, gx2, gy2;
          descriptor_residual_and_jacobian(kf, c, c0, c1, c2, d1, d2, &r1, &r2, &gx1, &gy1, &gx2, &gy2);
          if (kCoeffs) {
            gx1 *= c.cfx; gx2 *= c.cfx;
            gy1 *= c.cfy; gy2 *= c.cfy;
            // exact reciprocal here (ComputeRawDescriptorResidualAndJacobianWithFloatTexture BS/kernel_opt_pose.cu:168-190): odometry is not the hot path
            descriptor_pose_jacobian<true>(gx1, gy1, local, J1);
            descriptor_pose_jacobian<true>(gx2, gy2, local, J2);
          }
        } else {
          visible = false;
        }
      } else {
        visible = false;
      }
    }
    if (visible) {
      if (kCoeffs) {
        count = 1;
        if (kDepth) accumulate_h_b(raw_depth, 1.f * tukey_weight(raw_depth, threshold_factor * kDepthTukey), Jd, acc);
        if (kDesc) {
          accumulate_h_b(r1, threshold_factor * kDescWeight * huber_weight(r1, kDescHuber), J1, acc);
          if (!kGradMag) accumulate_h_b(r2, threshold_factor * kDescWeight * huber_weight(r2, kDescHuber), J2, acc);
        }
      } else {
        if (kDepth) { count += 1; acc[kRowCost] += 1.f * tukey_residual(raw_depth, threshold_factor * kDepthTukey); }
        if (kDesc && kGradMag) {
          count += 1;
          acc[kRowCost] += threshold_factor * kDescWeight * huber_residual(r1, kDescHuber);
        } else if (kDesc) {
          count += 2;
          acc[kRowCost] += threshold_factor * kDescWeight * huber_residual(r1, kDescHuber);
          acc[kRowCost] += threshold_factor * kDescWeight * huber_residual(r2, kDescHuber);
        }
      }
    }
  }
  acc[kRowCount] = (float)count;
  const float total = wave_transpose_sum32(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((lane & 1) == 0) partials[((size_t)blockIdx.x * (blockDim.x / 64) + wave) * kRow + (lane >> 1)] = total;
}

}  // namespace bslam
