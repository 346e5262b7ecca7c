// lifecycle_kernels.hpp -- surfel lifecycle for gfx950: supporting surfels + merge,
// creation (with the observation-count filter), deletion + radius update, compaction.
//
// Replaces BS/kernel_supporting_surfels.cu, BS/kernel_create_surfels.cu, BS/kernel_delete_surfels.cu and
// BS/kernel_compact_surfels.cu.  Two things differ by design:
//   * the reference decides cell ownership with atomicCAS races (any interleaving is a valid outcome); here
//     the outcome is fixed to "lowest surfel index first" (supporting surfels: the three holders of a cell
//     are its three smallest projecting surfel indices, found with atomicMin) and "raster order first"
//     (creation: one thread per cell scans its pixels), so results are deterministic;
//   * cub::DeviceScan is replaced by a three-kernel wave64 scan (scan_* below), and the per-keyframe
//     observation-count launches by one kernel that walks the keyframe table.
#pragma once

#include <hip/hip_fp16.h>

#include "device_math.hpp"

namespace bslam {

constexpr uint32_t kInvalidIndex = 0xffffffffu;
constexpr uint32_t kNanBits = 0x7fffffffu;   // CUDART_NAN_F marks a deleted surfel (BS/kernel_delete_surfels.cu:145)
constexpr int kMergeBufferCount = 3;         // BS/kernels.cuh:51

// Raw images of one keyframe, passed by value to the single-keyframe kernels (no derived records needed).
struct KfImages {
  const uint8_t* depth;   uint32_t depth_pitch;
  const uint8_t* normals; uint32_t normals_pitch;
  const uint8_t* radius;  uint32_t radius_pitch;
  const uint8_t* color;   uint32_t color_pitch;
  M34 frame_T_global;
};

__device__ __forceinline__ uint32_t img_u16(const uint8_t* base, uint32_t pitch, int y, int x) {
  return gload((const uint16_t*)(base + (size_t)y * pitch) + x);
}
__device__ __forceinline__ float cfactor_at(const CamConsts& c, int px, int py) {
  return gload((const float*)((const uint8_t*)c.cfactor + (size_t)(py / c.cell) * c.cfactor_pitch) + (px / c.cell));
}
__device__ __forceinline__ float half_bits_to_float(uint32_t h) { return __half2float(__ushort_as_half((unsigned short)h)); }

// Shared tail of the association test (BS/surfel_projection_nvcc_only.cuh:76-126), kFreeSpace selects the
// <true> variant's depth test.  n_local: surfel normal in the keyframe's frame.
template <bool kFreeSpace>
__device__ __forceinline__ bool association_tail(const CamConsts& c, f3 local, f3 n_local, int px, int py, float pixel_depth,
                                                 uint32_t pixel_normal, bool* fsv) {
  const float stddev = depth_stddev(nx_of(c, (float)px), ny_of(c, (float)py), pixel_depth, n_local, c.inv_baseline_fx);
  const float thr = kDepthTukey * stddev;
  if (kFreeSpace) {
    const float diff = pixel_depth - local.z;
    if (diff > thr) { *fsv = true; return false; }
    else if (diff < -thr) return false;
  } else {
    if (fabsf(local.z - pixel_depth) > thr) return false;
  }
  if (dot(local, n_local) > 0) return false;   // sign of (1 / |local|) * dot, see project_and_associate
  const f3 pn = u16_to_image_space_normal(pixel_normal);
  if (dot(n_local, pn) < kCosNormalCompat) return false;
  return true;
}

// SurfelProjectsToAssociatedPixel on the raw images of one keyframe.
template <bool kFreeSpace>
__device__ __forceinline__ bool associate_direct(const CamConsts& c, const KfImages& kf, f3 gp, f3 gn, int* px, int* py, bool* fsv) {
  const M34& T = kf.frame_T_global;
  f3 local;
  local.z = tr_row(T.m[8], T.m[9], T.m[10], T.m[11], gp);
  if (local.z <= 0.f) return false;
  local.x = tr_row(T.m[0], T.m[1], T.m[2], T.m[3], gp);
  local.y = tr_row(T.m[4], T.m[5], T.m[6], T.m[7], gp);
  const f2 pxy = project(c.fx, c.fy, c.cx, c.cy, local);
  *px = f2i(pxy.x);
  *py = f2i(pxy.y);
  if (pxy.x < 0 || pxy.y < 0 || *px >= c.width || *py >= c.height) return false;
  const uint32_t measured = img_u16(kf.depth, kf.depth_pitch, *py, *px);
  if (measured & BSLAM_INVALID_DEPTH_BIT) return false;
  const float depth = raw_to_calibrated_depth(c.a, cfactor_at(c, *px, *py), c.raw_to_float_depth, measured);
  return association_tail<kFreeSpace>(c, local, rot34(T, gn), *px, *py, depth, img_u16(kf.normals, kf.normals_pitch, *py, *px), fsv);
}

// Same test against the derived records of the keyframe table (free-space variant for deletion).
__device__ __forceinline__ bool associate_records_fs(const CamConsts& c, const KfDev& kf, f3 gp, f3 gn, int* px, int* py, bool* fsv) {
  const M34& T = kf.frame_T_global;
  f3 local;
  local.z = tr_row(T.m[8], T.m[9], T.m[10], T.m[11], gp);
  if (local.z <= 0.f) return false;
  local.x = tr_row(T.m[0], T.m[1], T.m[2], T.m[3], gp);
  local.y = tr_row(T.m[4], T.m[5], T.m[6], T.m[7], gp);
  const f2 pxy = project(c.fx, c.fy, c.cx, c.cy, local);
  *px = f2i(pxy.x);
  *py = f2i(pxy.y);
  if (pxy.x < 0 || pxy.y < 0 || *px >= c.width || *py >= c.height) return false;
  const PixelRecord rec = gload_record(kf.records + ((size_t)*py * c.width + *px));
  if (rec.depth == 0.f) return false;
  return association_tail<true>(c, local, rot34(T, gn), *px, *py, rec.depth, img_u16(kf.normals, kf.normals_pitch, *py, *px), fsv);
}

struct SurfelRowsAll {   // the eight persistent rows, writable
  float* x; float* y; float* z; uint32_t* normal; float* radius_squared; uint32_t* color; float* d1; float* d2;
};

// ---------------------------------------------------------------------------------------------
// supporting surfels (+ merge): BS/kernel_supporting_surfels.cu:45-97
// ---------------------------------------------------------------------------------------------
// Pass 0: association -> cell_of[i]; holder 0 of a cell = smallest projecting index.
__global__ __launch_bounds__(256) void support_claim0_kernel(CamConsts c, KfImages kf, SurfelRowsAll s, uint32_t size, int cells_w,
                                                            uint32_t* __restrict__ cell_of, uint32_t* __restrict__ sup0) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= size) return;
  int px, py;
  bool fsv;
  uint32_t cell = kInvalidIndex;
  if (associate_direct<false>(c, kf, mk3(s.x[i], s.y[i], s.z[i]), unpack_normal(s.normal[i]), &px, &py, &fsv)) {
    cell = (uint32_t)((py / c.cell) * cells_w + (px / c.cell));
    atomicMin(&sup0[cell], i);
  }
  cell_of[i] = cell;
}
// Pass 1 / 2: the next smallest index that is not an earlier holder.
__global__ __launch_bounds__(256) void support_claim_next_kernel(uint32_t size, const uint32_t* __restrict__ cell_of, const uint32_t* __restrict__ sup0,
                                                                const uint32_t* __restrict__ sup1, uint32_t* __restrict__ sup_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= size) return;
  const uint32_t cell = cell_of[i];
  if (cell == kInvalidIndex || sup0[cell] == i) return;
  if (sup1 != nullptr && sup1[cell] == i) return;
  atomicMin(&sup_out[cell], i);
}

__device__ __forceinline__ bool merge_close(const SurfelRowsAll& s, uint32_t sup, uint32_t self, float cell_merge_dist_squared, float cos_thr) {
  if (!(dot(unpack_normal(s.normal[sup]), unpack_normal(s.normal[self])) > cos_thr)) return false;
  const f3 d = sub3(mk3(s.x[sup], s.y[sup], s.z[sup]), mk3(s.x[self], s.y[self], s.z[self]));
  const float min_r = fminf(s.radius_squared[sup], s.radius_squared[self]);
  return sqlen(d) < min_r * cell_merge_dist_squared;   // false when either x is NaN (already merged away)
}

// Holders 1 and 2 of every cell, in arrival order (holder 1 meets holder 0; holder 2 meets holder 0 and then
// holder 1 as it is after its own test).
__global__ __launch_bounds__(256) void merge_holders_kernel(int cells, const uint32_t* __restrict__ sup0, const uint32_t* __restrict__ sup1,
                                                           const uint32_t* __restrict__ sup2, SurfelRowsAll s, float cell_merge_dist_squared,
                                                           float cos_thr, uint32_t* __restrict__ deleted_count) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= cells) return;
  const uint32_t h0 = sup0[cell], h1 = sup1[cell], h2 = sup2[cell];
  if (h1 == kInvalidIndex) return;
  uint32_t deleted = 0;
  const bool del1 = merge_close(s, h0, h1, cell_merge_dist_squared, cos_thr);
  bool del2 = false;
  if (h2 != kInvalidIndex) {
    del2 = merge_close(s, h0, h2, cell_merge_dist_squared, cos_thr);
    // the reference keeps walking the buffers after a deletion and counts `deleted = 1` once per surfel
    if (!del1 && merge_close(s, h1, h2, cell_merge_dist_squared, cos_thr)) del2 = true;
  }
  if (del1) { s.x[h1] = __uint_as_float(kNanBits); ++deleted; }
  if (del2) { s.x[h2] = __uint_as_float(kNanBits); ++deleted; }
  if (deleted) atomicAdd(deleted_count, deleted);
}

// Every other surfel of the cell arrives after the three holders.
__global__ __launch_bounds__(256) void merge_others_kernel(uint32_t size, const uint32_t* __restrict__ cell_of, const uint32_t* __restrict__ sup0,
                                                          const uint32_t* __restrict__ sup1, const uint32_t* __restrict__ sup2, SurfelRowsAll s,
                                                          float cell_merge_dist_squared, float cos_thr, uint32_t* __restrict__ deleted_count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool deleted = false;
  if (i < size) {
    const uint32_t cell = cell_of[i];
    if (cell != kInvalidIndex) {
      const uint32_t h0 = sup0[cell], h1 = sup1[cell], h2 = sup2[cell];
      if (i != h0 && i != h1 && i != h2) {
        deleted = merge_close(s, h0, i, cell_merge_dist_squared, cos_thr) || merge_close(s, h1, i, cell_merge_dist_squared, cos_thr) ||
                  merge_close(s, h2, i, cell_merge_dist_squared, cos_thr);
        if (deleted) s.x[i] = __uint_as_float(kNanBits);
      }
    }
  }
  const uint32_t n = wave_sum_u32(deleted ? 1u : 0u);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(deleted_count, n);
}

// ---------------------------------------------------------------------------------------------
// wave64 scan (replaces cub::DeviceScan): tiles of 1024 values, 4 per thread
// ---------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanPer = 4;
constexpr int kScanTile = kScanThreads * kScanPer;

// kind 0: in8[i];  kind 1: in32[i];  kind 2: reversed + inverted flags 1 - in32[n - 1 - i] (valid flags from the back)
__device__ __forceinline__ uint32_t scan_load(int kind, const void* in, uint32_t n, uint32_t i) {
  if (i >= n) return 0;
  if (kind == 0) return ((const uint8_t*)in)[i];
  if (kind == 1) return ((const uint32_t*)in)[i];
  return 1u - ((const uint32_t*)in)[n - 1 - i];
}

// out[i] = inclusive (or exclusive) prefix within the tile; tile_sums[b] = tile total.
__global__ __launch_bounds__(kScanThreads) void scan_tiles_kernel(int kind, const void* in, uint32_t n, int exclusive, uint32_t* __restrict__ out,
                                                                 uint32_t* __restrict__ tile_sums) {
  __shared__ uint32_t wave_tot[kScanThreads / 64];
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
  uint32_t v[kScanPer];
  uint32_t sum = 0;
#pragma unroll
  for (int j = 0; j < kScanPer; ++j) { v[j] = scan_load(kind, in, n, base + j); sum += v[j]; }
  // inclusive scan of the per-thread sums across the wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0;
  for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
  uint32_t run = wave_off + incl - sum;   // exclusive prefix of this thread's first element
#pragma unroll
  for (int j = 0; j < kScanPer; ++j) {
    const uint32_t i = base + j;
    if (exclusive) { if (i < n) out[i] = run; run += v[j]; }
    else { run += v[j]; if (i < n) out[i] = run; }
  }
  if (threadIdx.x == kScanThreads - 1) tile_sums[blockIdx.x] = wave_off + incl;
}

// Exclusive scan of the tile totals in place (one block); total[0] = grand total.
__global__ __launch_bounds__(256) void scan_sums_kernel(uint32_t* __restrict__ tile_sums, int tiles, uint32_t* __restrict__ total) {
  __shared__ uint32_t sm[256];
  uint32_t carry = 0;
  for (int base = 0; base < tiles; base += 256) {
    const int i = base + (int)threadIdx.x;
    const uint32_t v = (i < tiles) ? tile_sums[i] : 0;
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {   // Hillis-Steele inclusive scan
      const uint32_t t = (threadIdx.x >= (unsigned)off) ? sm[threadIdx.x - off] : 0;
      __syncthreads();
      sm[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < tiles) tile_sums[i] = carry + sm[threadIdx.x] - v;
    const uint32_t block_total = sm[255];
    __syncthreads();
    carry += block_total;
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(kScanThreads) void scan_add_kernel(uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ tile_sums) {
  const uint32_t off = tile_sums[blockIdx.x];
  const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
#pragma unroll
  for (int j = 0; j < kScanPer; ++j) if (base + j < n) out[base + j] += off;
}

// ---------------------------------------------------------------------------------------------
// creation: BS/kernel_create_surfels.cu
// ---------------------------------------------------------------------------------------------
// CreateSurfelsForKeyframeCUDASerializingKernel (:41-72): one thread per free cell, first valid pixel in
// raster order becomes the new surfel.  flags must be zeroed by the caller.
__global__ __launch_bounds__(256) void create_flag_kernel(CamConsts c, KfImages kf, int cells_w, int cells_h, uint32_t* __restrict__ sup0,
                                                         uint8_t* __restrict__ flags) {
  const int cellx = blockIdx.x * blockDim.x + threadIdx.x;
  const int celly = blockIdx.y;
  if (cellx >= cells_w || celly >= cells_h) return;
  uint32_t* occ = &sup0[(size_t)celly * cells_w + cellx];
  if (*occ != kInvalidIndex) return;
  constexpr int kBorder = 1;
  for (int y = celly * c.cell; y < min(c.height, (celly + 1) * c.cell); ++y) {
    for (int x = cellx * c.cell; x < min(c.width, (cellx + 1) * c.cell); ++x) {
      if (!(x >= kBorder && y >= kBorder && x < c.width - kBorder && y < c.height - kBorder)) continue;
      if (img_u16(kf.depth, kf.depth_pitch, y, x) & BSLAM_INVALID_DEPTH_BIT) continue;
      *occ = 0;
      flags[(size_t)y * c.width + x] = 1;
      return;
    }
  }
}

// WriteNewSurfelIndexAndInitializeObservations + CountObservationsForNewSurfels (per co-visible keyframe) +
// FilterNewSurfels (:163-305) in one pass per candidate pixel.
__global__ __launch_bounds__(256) void create_filter_kernel(CamConsts c, KfImages kf, int covis_count, const KfImages* __restrict__ covis,
                                                           const M34* __restrict__ covis_T_frame, int min_observation_count,
                                                           uint8_t* __restrict__ flags) {
  const uint32_t seq = blockIdx.x * blockDim.x + threadIdx.x;
  if (seq >= (uint32_t)(c.width * c.height) || flags[seq] != 1) return;
  const int y = (int)(seq / (uint32_t)c.width), x = (int)(seq - (uint32_t)y * (uint32_t)c.width);
  uint32_t observations = 1, violations = 0;
  const float depth = raw_to_calibrated_depth(c.a, cfactor_at(c, x, y), c.raw_to_float_depth, img_u16(kf.depth, kf.depth_pitch, y, x));
  const f3 input_position = unproject(c, x, y, depth);
  const f3 image_normal = u16_to_image_space_normal(img_u16(kf.normals, kf.normals_pitch, y, x));
  for (int k = 0; k < covis_count; ++k) {
    const KfImages& ck = covis[k];
    const M34& T = covis_T_frame[k];
    f3 local;
    local.z = tr_row(T.m[8], T.m[9], T.m[10], T.m[11], input_position);
    if (local.z <= 0.f) continue;
    local.x = tr_row(T.m[0], T.m[1], T.m[2], T.m[3], input_position);
    local.y = tr_row(T.m[4], T.m[5], T.m[6], T.m[7], input_position);
    const f2 pxy = project(c.fx, c.fy, c.cx, c.cy, local);
    const int px = f2i(pxy.x), py = f2i(pxy.y);
    if (pxy.x < 0 || pxy.y < 0 || px >= c.width || py >= c.height) continue;
    const uint32_t measured = img_u16(ck.depth, ck.depth_pitch, py, px);
    if (measured & BSLAM_INVALID_DEPTH_BIT) continue;
    const float pixel_depth = raw_to_calibrated_depth(c.a, cfactor_at(c, px, py), c.raw_to_float_depth, measured);
    bool fsv = false;
    if (association_tail<true>(c, local, rot34(T, image_normal), px, py, pixel_depth, img_u16(ck.normals, ck.normals_pitch, py, px), &fsv)) ++observations;
    else if (fsv) ++violations;
  }
  if ((uint16_t)observations < (uint16_t)min_observation_count || (uint16_t)violations > (uint16_t)observations) flags[seq] = 0;
}

// bilinear fetch of one colour channel from the uchar4 image (same filter model as tex_filter)
__device__ __forceinline__ float tex_channel_direct(const CamConsts& c, const KfImages& kf, float x, float y, int chn) {
  const TexFootprint f = tex_footprint(c, x, y);
  auto texel = [&](int ix, int iy) {
    ix = max(0, min(ix, c.color_width - 1));
    iy = max(0, min(iy, c.color_height - 1));
    return (float)gload(kf.color + (size_t)iy * kf.color_pitch + 4 * (size_t)ix + chn) * (1.0f / 255.0f);
  };
  LumaQuad q;
  q.tl = texel(f.i, f.j); q.tr = texel(f.i + 1, f.j); q.bl = texel(f.i, f.j + 1); q.br = texel(f.i + 1, f.j + 1);
  return tex_filter(q, f.a, f.b);
}

// the luma channel in byte units through the residual path's filter (device_math.hpp: bilinear_bytes): the initial descriptor
// is ComputeRawDescriptorResidual with a zero descriptor (:141-152), i.e. the same arithmetic as every later residual
__device__ __forceinline__ float tex_luma_bytes_direct(const CamConsts& c, const KfImages& kf, float x, float y) {
  const TexFootprint f = tex_footprint(c, x, y);
  auto texel = [&](int ix, int iy) {
    ix = max(0, min(ix, c.color_width - 1));
    iy = max(0, min(iy, c.color_height - 1));
    return (float)gload(kf.color + (size_t)iy * kf.color_pitch + 4 * (size_t)ix + 3);
  };
  LumaQuad q;
  q.tl = texel(f.i, f.j); q.tr = texel(f.i + 1, f.j); q.bl = texel(f.i, f.j + 1); q.br = texel(f.i + 1, f.j + 1);
  return bilinear_bytes(q, f.a, f.b);
}

// CreateSurfelsForKeyframeCUDACreationAppendKernel + CreateNewSurfel (:96-161, 357-385)
__global__ __launch_bounds__(256) void create_append_kernel(CamConsts c, KfImages kf, M34 global_T_frame, const uint8_t* __restrict__ flags,
                                                           const uint32_t* __restrict__ indices, uint32_t surfels_size, SurfelRowsAll s) {
  const uint32_t seq = blockIdx.x * blockDim.x + threadIdx.x;
  if (seq >= (uint32_t)(c.width * c.height) || flags[seq] != 1) return;
  const int y = (int)(seq / (uint32_t)c.width), x = (int)(seq - (uint32_t)y * (uint32_t)c.width);
  const uint32_t si = surfels_size + indices[seq] - 1;
  const float depth = raw_to_calibrated_depth(c.a, cfactor_at(c, x, y), c.raw_to_float_depth, img_u16(kf.depth, kf.depth_pitch, y, x));
  const f3 gp = mul34(global_T_frame, unproject(c, x, y, depth));
  s.x[si] = gp.x; s.y[si] = gp.y; s.z[si] = gp.z;
  const f3 gn = rot34(global_T_frame, u16_to_image_space_normal(img_u16(kf.normals, kf.normals_pitch, y, x)));
  s.normal[si] = pack_normal(gn);
  const float radius_squared = half_bits_to_float(img_u16(kf.radius, kf.radius_pitch, y, x));
  s.radius_squared[si] = radius_squared;
  f2 color_pxy;
  depth_to_color_pxy(c, f2{x + 0.5f, y + 0.5f}, &color_pxy);
  uint32_t col = 0;
#pragma unroll
  for (int chn = 0; chn < 3; ++chn) col |= ((uint32_t)f2i(255.f * tex_channel_direct(c, kf, color_pxy.x, color_pxy.y, chn)) & 0xffu) << (8 * chn);
  s.color[si] = col;
  f2 t1, t2;
  tangent_projections(gp, gn, radius_squared, kf.frame_T_global, c, &t1, &t2);   // the unquantised normal, as in the reference (:124-131)
  const float b0 = tex_luma_bytes_direct(c, kf, color_pxy.x, color_pxy.y);
  const float b1 = tex_luma_bytes_direct(c, kf, t1.x, t1.y);
  const float b2 = tex_luma_bytes_direct(c, kf, t2.x, t2.y);
  s.d1[si] = __builtin_fmaf(kDescScale, b1 - b0, -0.f);
  s.d2[si] = __builtin_fmaf(kDescScale, b2 - b0, -0.f);
}

// ---------------------------------------------------------------------------------------------
// deletion + radius update: BS/kernel_delete_surfels.cu (reset + K count launches + mark, fused)
// ---------------------------------------------------------------------------------------------
// Work order as in the BA kernels (make_schedule): block -> slot of an XCD-contiguous range, position -> surfel column through
// the per-surfel Morton permutation `perm` (nullptr: identity).  K = 300, S = 5.76 M: 30.7 ms -> about 5 ms (PCG BA iteration 370 -> 344 ms).
__global__ __launch_bounds__(256) void delete_and_update_radii_kernel(CamConsts c, const KfDev* __restrict__ kfs, int kf_count, int min_observation_count,
                                                                     uint32_t size, Schedule sc, const uint32_t* __restrict__ perm, SurfelRows sorted, SurfelRowsAll s,
                                                                     uint32_t* __restrict__ deleted_count) {
  // `sorted`: with a per-surfel order, the library's sorted copy of position and normal (position `at` of it is the caller's
  // column perm[at]); the keyframes a work slot cannot see are skipped (block-level frustum culling: a surfel that does not
  // project into a keyframe is neither an observation nor a free-space violation there)
  uint32_t slot;
  if (!slot_of_block(sc, blockIdx.x, &slot)) return;
  const uint32_t at = surfel_of_slot(sc, slot, 0, 1);
  const uint32_t i = (perm != nullptr && at < size) ? perm[at] : at;
  bool deleted = false;
  const bool valid = i < size;
  const uint32_t j = valid ? at : 0;
  const f3 gp = mk3(sorted.x[j], sorted.y[j], sorted.z[j]);
  const f3 gn = unpack_normal(sorted.normal[j]);
  float observations = 0.f, violations = 0.f, min_radius = __uint_as_float(0x7f800000u);
  BSLAM_FOR_VISITED_KEYFRAMES_IF(k, 0, kf_count, 1, true) {
    int px, py;
    bool fsv = false;
    if (!valid) continue;
    if (associate_records_fs(c, kfs[k], gp, gn, &px, &py, &fsv)) {
      observations += 1.f;
      min_radius = fminf(min_radius, half_bits_to_float(img_u16(kfs[k].radius, kfs[k].radius_pitch, py, px)));
    } else if (fsv) {
      violations += 1.f;
    }
  }
  if (valid) {
    if (observations < (float)min_observation_count || violations > observations) {
      if (__float_as_uint(gp.x) != kNanBits) { s.x[i] = __uint_as_float(kNanBits); deleted = true; }
    } else {
      s.radius_squared[i] = min_radius;
    }
  }
  const uint32_t n = wave_sum_u32(deleted ? 1u : 0u);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(deleted_count, n);
}

// ---------------------------------------------------------------------------------------------
// compaction: BS/kernel_compact_surfels.cu:97-281
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void compact_flag_kernel(uint32_t size, const float* __restrict__ x, uint32_t* __restrict__ invalid) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < size) invalid[i] = (__float_as_uint(x[i]) == kNanBits) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void compact_free_list_kernel(uint32_t size, uint32_t free_spot_count, const uint32_t* __restrict__ invalid,
                                                               const uint32_t* __restrict__ free_ordinal, uint32_t* __restrict__ free_list) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < size && invalid[i] && free_ordinal[i] < free_spot_count) free_list[free_ordinal[i]] = i;
}
// reverse_index_rev[n - 1 - i] = number of valid surfels behind surfel i
__global__ __launch_bounds__(256) void compact_move_kernel(uint32_t size, uint32_t free_spot_count, const uint32_t* __restrict__ invalid,
                                                          const uint32_t* __restrict__ reverse_index_rev, const uint32_t* __restrict__ free_list,
                                                          uint8_t* surfels_base, size_t pitch, uint8_t* active) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= size || invalid[i]) return;
  const uint32_t r = reverse_index_rev[size - 1 - i];
  if (r >= free_spot_count) return;
  const uint32_t spot = free_list[r];
  if (spot >= i) return;
#pragma unroll
  for (int row = 0; row < BSLAM_SURFEL_DATA_ATTRIBUTE_COUNT; ++row) {
    float* p = (float*)(surfels_base + (size_t)row * pitch);
    p[spot] = p[i];
  }
  if (active) active[spot] = active[i];
}

}  // namespace bslam
