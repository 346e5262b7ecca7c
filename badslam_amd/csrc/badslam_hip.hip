// badslam_hip.hip -- C ABI entry points (include/badslam_hip.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see build.py).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "context.hpp"
#include "geometry_kernels.hpp"
#include "intrinsics_kernels.hpp"
#include "pcg_kernels.hpp"
#include "lifecycle_kernels.hpp"
#include "preprocess_kernels.hpp"
#include "odometry_kernels.hpp"
#include "pose_kernels.hpp"

namespace bslam {

thread_local std::string g_last_error;

// ---- host-side construction of the kernel constants -------------------------------------------
// Same formulas, in fp32, as the reference's factories: CreatePixelCornerProjector,
// CreatePixelCenterUnprojector, CreateDepthToColorPixelCorner (BS/surfel_projection.h:42-124).
// Compiled with -ffp-contract=off like everything else in this library.
static CamConsts make_cam_consts(const bslam_context* ctx, const bslam_camera4f* color, const bslam_camera4f* depth,
                                 const bslam_depth_params* dp) {
  CamConsts c;
  std::memset(&c, 0, sizeof(c));
  c.fx = depth->fx; c.fy = depth->fy; c.cx = depth->cx; c.cy = depth->cy;
  c.width = depth->width; c.height = depth->height;
  c.fx_inv = 1.0f / depth->fx;
  c.fy_inv = 1.0f / depth->fy;
  const float cx_pixel_center = depth->cx - 0.5f;
  const float cy_pixel_center = depth->cy - 0.5f;
  c.cx_inv = -cx_pixel_center * c.fx_inv;
  c.cy_inv = -cy_pixel_center * c.fy_inv;
  if (color) {
    c.d2c_fx = color->fx / depth->fx;
    c.d2c_cx = -1 * color->fx * depth->cx / depth->fx + color->cx;
    c.d2c_fy = color->fy / depth->fy;
    c.d2c_cy = -1 * color->fy * depth->cy / depth->fy + color->cy;
    c.color_width = color->width; c.color_height = color->height;
    c.cfx = color->fx; c.cfy = color->fy; c.ccx = color->cx; c.ccy = color->cy;
  }
  c.a = dp->a;
  c.raw_to_float_depth = dp->raw_to_float_depth;
  c.baseline_fx = dp->baseline_fx;
  c.inv_baseline_fx = 1.0f / dp->baseline_fx;   // IEEE single division on the host, as in the CPU oracle
  c.cell = dp->sparse_surfel_cell_size;
  c.cfactor = (const float*)dp->cfactor_buffer.address;
  c.cfactor_pitch = (uint32_t)dp->cfactor_buffer.pitch;
  c.cfactor_width = dp->cfactor_buffer.width;
  c.tex_mode = ctx->tex_mode;
  c.d2c_identity = color && c.d2c_fx == 1.0f && c.d2c_fy == 1.0f && c.d2c_cx == 0.0f && c.d2c_cy == 0.0f && c.color_width == c.width && c.color_height == c.height;
  return c;
}

static void fill_kf(KfDev* d, const bslam_buffer2d* depth, const bslam_buffer2d* normals, const bslam_buffer2d* color,
                    const bslam_mat3x4* frame_T_global, const bslam_mat3x3* global_R_frame, int activation, int id,
                    const bslam_buffer2d* radius = nullptr) {
  std::memset(d, 0, sizeof(*d));
  d->depth = (const uint8_t*)depth->address;     d->depth_pitch = (uint32_t)depth->pitch;
  d->normals = (const uint8_t*)normals->address; d->normals_pitch = (uint32_t)normals->pitch;
  if (color) { d->color = (const uint8_t*)color->address; d->color_pitch = (uint32_t)color->pitch; }
  if (radius) { d->radius = (const uint8_t*)radius->address; d->radius_pitch = (uint32_t)radius->pitch; }
  std::memcpy(d->frame_T_global.m, frame_T_global->m, sizeof(float) * 12);
  if (global_R_frame) std::memcpy(d->global_R_frame, global_R_frame->m, sizeof(float) * 9);
  d->activation = activation;
  d->id = id;
}

static int check_common(const bslam_context* ctx, const bslam_camera4f* depth_camera, const bslam_depth_params* dp,
                        const bslam_buffer2d* surfels) {
  if (!ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "context is null");
  if (!depth_camera || !dp || !surfels) return fail(BSLAM_ERR_INVALID_ARGUMENT, "null argument");
  if (dp->sparse_surfel_cell_size < 1) return fail(BSLAM_ERR_INVALID_ARGUMENT, "sparse_surfel_cell_size must be >= 1");
  if (!dp->cfactor_buffer.address) return fail(BSLAM_ERR_INVALID_ARGUMENT, "cfactor_buffer is null");
  const int need_w = (depth_camera->width - 1) / dp->sparse_surfel_cell_size + 1;
  const int need_h = (depth_camera->height - 1) / dp->sparse_surfel_cell_size + 1;
  if (dp->cfactor_buffer.width < need_w || dp->cfactor_buffer.height < need_h)
    return fail(BSLAM_ERR_INVALID_ARGUMENT, "cfactor_buffer is %dx%d, camera %dx%d at cell %d needs %dx%d",
                dp->cfactor_buffer.width, dp->cfactor_buffer.height, depth_camera->width, depth_camera->height,
                dp->sparse_surfel_cell_size, need_w, need_h);
  if (surfels->height < BSLAM_SURFEL_DATA_ATTRIBUTE_COUNT) return fail(BSLAM_ERR_INVALID_ARGUMENT, "surfel buffer has %d rows, need >= %d", surfels->height, BSLAM_SURFEL_DATA_ATTRIBUTE_COUNT);
  return BSLAM_OK;
}

static int check_image(const bslam_buffer2d* b, const bslam_camera4f* cam, size_t elem, const char* name) {
  if (!b || !b->address) return fail(BSLAM_ERR_INVALID_ARGUMENT, "%s buffer is null", name);
  if (b->width != cam->width || b->height != cam->height)
    return fail(BSLAM_ERR_INVALID_ARGUMENT, "%s buffer is %dx%d but the camera is %dx%d", name, b->width, b->height, cam->width, cam->height);
  if (b->pitch < (size_t)b->width * elem) return fail(BSLAM_ERR_INVALID_ARGUMENT, "%s pitch %zu too small", name, b->pitch);
  return BSLAM_OK;
}

static SurfelRows surfel_rows(const bslam_buffer2d* s, uint32_t size) {
  auto row = [&](int r) { return (const float*)((const uint8_t*)s->address + (size_t)r * s->pitch); };
  SurfelRows o;
  o.x = row(BSLAM_SURFEL_X); o.y = row(BSLAM_SURFEL_Y); o.z = row(BSLAM_SURFEL_Z);
  o.normal = (const uint32_t*)row(BSLAM_SURFEL_NORMAL);
  o.radius_squared = row(BSLAM_SURFEL_RADIUS_SQUARED);
  o.d1 = row(BSLAM_SURFEL_DESCRIPTOR1); o.d2 = row(BSLAM_SURFEL_DESCRIPTOR2);
  o.size = size;
  return o;
}

static SurfelRowsRW surfel_rows_rw(const bslam_buffer2d* s, const bslam_buffer2d* active, uint32_t size) {
  auto row = [&](int r) { return (float*)((uint8_t*)s->address + (size_t)r * s->pitch); };
  SurfelRowsRW o;
  o.x = row(BSLAM_SURFEL_X); o.y = row(BSLAM_SURFEL_Y); o.z = row(BSLAM_SURFEL_Z);
  o.normal = (uint32_t*)row(BSLAM_SURFEL_NORMAL);
  o.radius_squared = row(BSLAM_SURFEL_RADIUS_SQUARED);
  o.d1 = row(BSLAM_SURFEL_DESCRIPTOR1); o.d2 = row(BSLAM_SURFEL_DESCRIPTOR2);
  o.active = active ? (uint8_t*)active->address : nullptr;
  o.size = size;
  o.perm = nullptr;
  o.ox = o.x; o.oy = o.y; o.oz = o.z; o.onormal = o.normal; o.od1 = o.d1; o.od2 = o.d2;
  return o;
}

// Uploads a keyframe table through pinned staging.  The staging buffer is only rewritten after
// the previous upload has been consumed (stream-ordered, so we wait for the stream first if the
// same staging is still in flight -- uploads are a few KB and the wait is normally a no-op).
static int upload_kf_table(bslam_context* ctx, hipStream_t stream, std::vector<KfDev>& table, const CamConsts& c) {
  const size_t bytes = table.size() * sizeof(KfDev);
  int rc = ctx->kf_table.reserve(bytes);
  if (rc) return rc;
  const size_t rec_per_kf = (size_t)c.width * c.height;
  if ((rc = ctx->records.reserve(table.size() * rec_per_kf * sizeof(PixelRecord)))) return rc;
  for (size_t k = 0; k < table.size(); ++k) table[k].records = (const PixelRecord*)ctx->records.ptr + k * rec_per_kf;
  const bool with_color = !table.empty() && table[0].color != nullptr;
  const size_t quads_per_kf = (size_t)(c.color_width + 1) * (size_t)(c.color_height + 1);
  if (with_color) {
    if ((rc = ctx->quads.reserve(table.size() * quads_per_kf * sizeof(uint32_t)))) return rc;
    for (size_t k = 0; k < table.size(); ++k) table[k].quads = (const uint32_t*)ctx->quads.ptr + k * quads_per_kf;
  } else {
    for (size_t k = 0; k < table.size(); ++k) table[k].quads = nullptr;
  }
  // The calls of one BA iteration (activation, geometry, poses) hand in the same list with the same poses: an upload that
  // would not change a byte of the device copy is skipped (each costs a copy node and two launch gaps on the stream, about
  // 10 us -- 2 % of a K = 50 iteration).  Stream order makes this safe across calls on one stream, the only use the boundary
  // allows for one context.
  const bool same = ctx->kf_table_host_ptr == ctx->kf_table.ptr && ctx->kf_table_host.size() == bytes &&
                    std::memcmp(ctx->kf_table_host.data(), table.data(), bytes) == 0;
  if (!same) {
    void* stage = nullptr;
    if ((rc = ctx->upload_ring.acquire(bytes, &stage))) return rc;
    std::memcpy(stage, table.data(), bytes);
    BSLAM_HIP_TRY(hipMemcpyAsync(ctx->kf_table.ptr, stage, bytes, hipMemcpyHostToDevice, stream));
    if ((rc = ctx->upload_ring.commit(stream))) return rc;
    ctx->kf_table_host.assign((const uint8_t*)table.data(), (const uint8_t*)table.data() + bytes);
    ctx->kf_table_host_ptr = ctx->kf_table.ptr;
  }
  // derived pixel records / luma quads: rebuilt on every call because the caller owns (and may have rewritten)
  // the depth / normal / colour / cfactor images between calls -- unless the caller promised otherwise
  // (bslam_set_keyframe_cache) and nothing they depend on has changed since they were built
  bool rebuild_records = true, rebuild_quads = with_color;
  if (ctx->keyframe_cache) {
    std::vector<uint64_t> sig;
    sig.reserve(table.size() * 3 + 8);
    for (const KfDev& kf : table) { sig.push_back((uint64_t)kf.depth); sig.push_back((uint64_t)kf.normals); sig.push_back(((uint64_t)kf.depth_pitch << 32) | kf.normals_pitch); }
    uint32_t fa, fr; std::memcpy(&fa, &c.a, 4); std::memcpy(&fr, &c.raw_to_float_depth, 4);
    sig.push_back((uint64_t)c.cfactor); sig.push_back(((uint64_t)c.cfactor_pitch << 32) | (uint32_t)c.cell);
    sig.push_back(((uint64_t)fa << 32) | fr); sig.push_back(((uint64_t)c.width << 32) | (uint32_t)c.height);
    sig.push_back((uint64_t)ctx->records.ptr);
    if (sig == ctx->records_signature) rebuild_records = false; else ctx->records_signature.swap(sig);
    if (with_color) {
      std::vector<uint64_t> qs;
      qs.reserve(table.size() * 2 + 2);
      for (const KfDev& kf : table) { qs.push_back((uint64_t)kf.color); qs.push_back((uint64_t)kf.color_pitch); }
      qs.push_back((uint64_t)ctx->quads.ptr); qs.push_back(((uint64_t)c.color_width << 32) | (uint32_t)c.color_height);
      if (qs == ctx->quads_signature) rebuild_quads = false; else ctx->quads_signature.swap(qs);
    }
  }
  if (rebuild_records && !table.empty() && table[0].depth != nullptr) {
    hipLaunchKernelGGL(build_records_kernel, dim3((unsigned)((c.width + 255) / 256), (unsigned)c.height, (unsigned)table.size()), dim3(256), 0, stream,
                       c, (const KfDev*)ctx->kf_table.ptr, (PixelRecord*)ctx->records.ptr);
    BSLAM_HIP_TRY(hipGetLastError());
  }
  if (rebuild_quads) {
    hipLaunchKernelGGL(build_quads_kernel, dim3((unsigned)((c.color_width + 1 + 255) / 256), (unsigned)(c.color_height + 1), (unsigned)table.size()), dim3(256), 0, stream,
                       c, (const KfDev*)ctx->kf_table.ptr, (uint32_t*)ctx->quads.ptr);
    BSLAM_HIP_TRY(hipGetLastError());
  }
  return BSLAM_OK;
}

static int build_kf_table(const bslam_camera4f* depth_camera, const bslam_camera4f* color_camera, bool need_color,
                          int keyframe_count, const bslam_keyframe_view* keyframes, std::vector<KfDev>* table, bool need_radius = false) {
  table->resize((size_t)keyframe_count);
  for (int k = 0; k < keyframe_count; ++k) {
    const bslam_keyframe_view& v = keyframes[k];
    int rc = check_image(&v.depth, depth_camera, 2, "keyframe depth");
    if (rc) return rc;
    rc = check_image(&v.normals, depth_camera, 2, "keyframe normals");
    if (rc) return rc;
    if (need_color) {
      rc = check_image(&v.color, color_camera, 4, "keyframe color");
      if (rc) return rc;
    }
    if (need_radius) {
      rc = check_image(&v.radius, depth_camera, 2, "keyframe radius");
      if (rc) return rc;
    }
    fill_kf(&(*table)[k], &v.depth, &v.normals, need_color ? &v.color : nullptr, &v.frame_T_global, &v.global_R_frame, v.activation, v.id,
            need_radius ? &v.radius : nullptr);
  }
  return BSLAM_OK;
}

// Builds (or re-uses) the XCD-aware granule order for a surfel buffer: centroid per granule on the
// device, Morton sort on the host (a few thousand keys), uploaded once and cached until the buffer
// or its size changes.  The order only steers locality, never results' validity.
static uint32_t morton10(uint32_t v) {
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// Keyframe lists of at least this length walk the surfels in per-surfel Morton order (`perm`); shorter ones (the per-keyframe
// entry points) keep coalesced surfel columns and only order the 256-column granules.
constexpr int kPermMinKeyframes = 4;

// Work order of the surfel kernels, cached per surfel buffer (address, size, pitch).  Long keyframe lists: a permutation of
// the surfel columns sorted by the 30-bit Morton code of their positions (device radix sort, rocPRIM); the kernels then read a
// sorted COPY of the surfel rows (prepare_surfels), so that their loads stay coalesced while the 64 surfels of a wave -- and the
// R x 256 of a workgroup -- form a compact blob that projects onto a few cache lines in every keyframe.
// Surfels are created in cell-raster order (BS/kernel_create_surfels.cu), i.e. 256 consecutive columns are a 1-pixel-high
// strip hundreds of pixels long: K = 50, geometry iteration 350 -> 306 us (geometry-only), 701 -> 640 us (photometric), pose
// kernels -4 ... -5 %.  Any permutation gives the same results up to the order of the per-keyframe sums; a cached order that has
// gone stale (surfels moved) only costs locality.
static int make_schedule(bslam_context* ctx, hipStream_t stream, const bslam_buffer2d* surfels, uint32_t surfels_size, int R, Schedule* out,
                         int keyframe_count = 1, const uint32_t** perm_out = nullptr) {
  const uint32_t G = (surfels_size + kGranule - 1) / kGranule;
  out->granules = G;
  out->slots = (G + (uint32_t)R - 1) / (uint32_t)R;
  out->slots_per_xcd = (out->slots + 7) / 8;
  out->order = nullptr;
  out->bounds = nullptr;
  if (perm_out) *perm_out = nullptr;
  if (!ctx->use_schedule || G < 64) return BSLAM_OK;   // tiny problems: identity order
  const bool want_perm = perm_out != nullptr && keyframe_count >= kPermMinKeyframes;
  if (want_perm && ctx->perm_key_ptr == surfels->address && ctx->perm_key_size == surfels_size && ctx->perm_key_pitch == surfels->pitch) {
    // the permutation was sorted asynchronously on the stream of the call that built it: a call on another stream waits for it
    if (stream != ctx->perm_stream && ctx->perm_ready) BSLAM_HIP_TRY(hipStreamWaitEvent(stream, ctx->perm_ready, 0));
    *perm_out = (const uint32_t*)ctx->perm.ptr;
    return BSLAM_OK;
  }
  if (!want_perm && ctx->order_key_ptr == surfels->address && ctx->order_key_size == surfels_size && ctx->order_key_pitch == surfels->pitch) {
    out->order = (const uint32_t*)ctx->order.ptr;
    return BSLAM_OK;
  }
  // granule centroids live in their own slab: the per-surfel path below must not disturb the cached granule order
  // (ctx->order) of another buffer, whose key stays valid
  int rc = ctx->centroids.reserve((size_t)G * sizeof(float4));
  if (rc) return rc;
  float4* d_cent = (float4*)ctx->centroids.ptr;
  auto row = [&](int r) { return (const float*)((const uint8_t*)surfels->address + (size_t)r * surfels->pitch); };
  hipLaunchKernelGGL(granule_centroid_kernel, dim3(G), dim3(kGranule), 0, stream, row(BSLAM_SURFEL_X), row(BSLAM_SURFEL_Y), row(BSLAM_SURFEL_Z), surfels_size, d_cent);
  BSLAM_HIP_TRY(hipGetLastError());
  std::vector<float4> cent(G);
  BSLAM_HIP_TRY(hipMemcpyAsync(cent.data(), d_cent, (size_t)G * sizeof(float4), hipMemcpyDeviceToHost, stream));
  BSLAM_HIP_TRY(hipStreamSynchronize(stream));
  float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
  for (uint32_t g = 0; g < G; ++g) {
    if (cent[g].w <= 0.f) continue;
    const float v[3] = {cent[g].x, cent[g].y, cent[g].z};
    for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], v[d]); hi[d] = std::max(hi[d], v[d]); }
  }
  if (want_perm) {
    // keys + identity -> radix sort by key -> perm.  Layout of ctx->perm: perm[S] | keys[S] | keys_sorted[S] | ids[S] | sort scratch
    const size_t n = surfels_size, words = (n + 63) & ~(size_t)63;
    size_t temp_bytes = 0;
    ctx->perm_key_ptr = nullptr;   // as above
    BSLAM_HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 30, stream));
    if ((rc = ctx->perm.reserve(4 * words * sizeof(uint32_t) + temp_bytes + 256))) return rc;
    uint32_t* d_perm = (uint32_t*)ctx->perm.ptr;
    uint32_t* d_keys = d_perm + words;
    uint32_t* d_keys_sorted = d_keys + words;
    uint32_t* d_ids = d_keys_sorted + words;
    void* d_temp = (void*)(d_ids + words);
    f3 lo3{lo[0], lo[1], lo[2]}, inv3;
    inv3.x = hi[0] > lo[0] ? 1.f / (hi[0] - lo[0]) : 0.f;
    inv3.y = hi[1] > lo[1] ? 1.f / (hi[1] - lo[1]) : 0.f;
    inv3.z = hi[2] > lo[2] ? 1.f / (hi[2] - lo[2]) : 0.f;
    hipLaunchKernelGGL(surfel_morton_key_kernel, dim3((surfels_size + 255) / 256), dim3(256), 0, stream, row(BSLAM_SURFEL_X), row(BSLAM_SURFEL_Y), row(BSLAM_SURFEL_Z),
                       surfels_size, lo3, inv3, d_keys, d_ids);
    BSLAM_HIP_TRY(hipGetLastError());
    BSLAM_HIP_TRY(rocprim::radix_sort_pairs(d_temp, temp_bytes, (const uint32_t*)d_keys, d_keys_sorted, (const uint32_t*)d_ids, d_perm, n, 0, 30, stream));
    if (!ctx->perm_ready) BSLAM_HIP_TRY(hipEventCreateWithFlags(&ctx->perm_ready, hipEventDisableTiming));
    BSLAM_HIP_TRY(hipEventRecord(ctx->perm_ready, stream));
    ctx->perm_stream = stream;
    ++ctx->perm_serial;
    ctx->perm_key_ptr = surfels->address;
    ctx->perm_key_size = surfels_size;
    ctx->perm_key_pitch = surfels->pitch;
    *perm_out = d_perm;
    return BSLAM_OK;
  }
  ctx->order_key_ptr = nullptr;   // the slab may move and is rewritten below: no stale key survives an early return
  if ((rc = ctx->order.reserve((size_t)G * sizeof(uint32_t)))) return rc;
  uint32_t* d_order = (uint32_t*)ctx->order.ptr;
  std::vector<std::pair<uint32_t, uint32_t>> keyed(G);
  for (uint32_t g = 0; g < G; ++g) {
    uint32_t key = 0x3fffffffu;   // empty granules last
    if (cent[g].w > 0.f) {
      const float v[3] = {cent[g].x, cent[g].y, cent[g].z};
      uint32_t q[3];
      for (int d = 0; d < 3; ++d) {
        const float span = hi[d] - lo[d];
        const float t = span > 0.f ? (v[d] - lo[d]) / span : 0.f;
        q[d] = (uint32_t)std::min(1023.f, std::max(0.f, t * 1023.f));
      }
      key = morton10(q[0]) | (morton10(q[1]) << 1) | (morton10(q[2]) << 2);
    }
    keyed[g] = std::make_pair(key, g);
  }
  std::sort(keyed.begin(), keyed.end());
  std::vector<uint32_t> order(G);
  for (uint32_t g = 0; g < G; ++g) order[g] = keyed[g].second;
  BSLAM_HIP_TRY(hipMemcpyAsync(d_order, order.data(), (size_t)G * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  BSLAM_HIP_TRY(hipStreamSynchronize(stream));   // `order` is a stack-local vector
  ctx->order_key_ptr = surfels->address;
  ctx->order_key_size = surfels_size;
  ctx->order_key_pitch = surfels->pitch;
  out->order = d_order;
  return BSLAM_OK;
}

// Schedule + the rows the surfel kernels of one API call read: the caller's rows, or (per-surfel order) the library's sorted
// copy of them, rebuilt here because the caller may have changed the surfels since the last call (7 rows: 28 B per surfel).
struct SurfelWork {
  Schedule sc;
  SurfelRows rows;          // what the kernels read
  const uint32_t* perm;     // position in `rows` -> caller's column, or nullptr
};
static int prepare_surfels(bslam_context* ctx, hipStream_t stream, const bslam_buffer2d* surfels, uint32_t surfels_size, int R, int keyframe_count, SurfelWork* w,
                           bool need_descriptor_rows = true, bool same_surfels_as_last_call = false) {
  int rc = make_schedule(ctx, stream, surfels, surfels_size, R, &w->sc, keyframe_count, &w->perm);
  if (rc) return rc;
  w->rows = surfel_rows(surfels, surfels_size);
  if (!w->perm) { ctx->sorted_key_ptr = nullptr; return BSLAM_OK; }
  const size_t pitch = ((size_t)surfels_size + 63) & ~(size_t)63;
  if ((rc = ctx->sorted_rows.reserve(7 * pitch * sizeof(float)))) return rc;
  float* out = (float*)ctx->sorted_rows.ptr;
  // bounding boxes of the granules of the copy, from the very values the kernels will read (frustum culling)
  float4* bounds = nullptr;
  if (ctx->culling) {
    if ((rc = ctx->bounds.reserve(2 * (size_t)w->sc.granules * sizeof(float4)))) return rc;
    bounds = (float4*)ctx->bounds.ptr;
  }
  // the copy costs a scattered 4-byte read per row and surfel: rows the call's kernels never read (radius, descriptors in a
  // geometry-only call) are left out
  const int copy_rows = need_descriptor_rows ? 7 : 4;
  const bool reuse = same_surfels_as_last_call && ctx->sorted_key_ptr == surfels->address && ctx->sorted_key_size == surfels_size &&
                     ctx->sorted_key_pitch == surfels->pitch && ctx->sorted_key_rows >= copy_rows && ctx->sorted_key_bounds == (bounds != nullptr) &&
                     ctx->sorted_key_perm_serial == ctx->perm_serial;
  if (reuse) {
    // K = 300 photometric PCG: 0.47 ms per PCGStep1 call, 8.5 of the 325 ms of a BA iteration
  } else if (need_descriptor_rows)
    hipLaunchKernelGGL(permute_surfel_rows_kernel<7>, dim3(w->sc.granules), dim3(256), 0, stream, w->perm, surfels_size, (const float*)surfels->address,
                       surfels->pitch / sizeof(float), out, pitch, bounds);
  else
    hipLaunchKernelGGL(permute_surfel_rows_kernel<4>, dim3(w->sc.granules), dim3(256), 0, stream, w->perm, surfels_size, (const float*)surfels->address,
                       surfels->pitch / sizeof(float), out, pitch, bounds);
  BSLAM_HIP_TRY(hipGetLastError());
  ctx->sorted_key_ptr = surfels->address; ctx->sorted_key_size = surfels_size; ctx->sorted_key_pitch = surfels->pitch;
  if (!reuse) { ctx->sorted_key_rows = copy_rows; ctx->sorted_key_bounds = bounds != nullptr; ctx->sorted_key_perm_serial = ctx->perm_serial; }
  w->sc.bounds = bounds;
  w->rows.x = out; w->rows.y = out + pitch; w->rows.z = out + 2 * pitch;
  w->rows.normal = (const uint32_t*)(out + 3 * pitch);
  w->rows.radius_squared = out + 4 * pitch;
  w->rows.d1 = out + 5 * pitch; w->rows.d2 = out + 6 * pitch;
  return BSLAM_OK;
}
// The same for kernels that also update surfels: reads (and write-through) on w.rows, updates to the caller's rows as well.
static SurfelRowsRW surfel_rows_rw(const SurfelWork& w, const bslam_buffer2d* surfels, const bslam_buffer2d* active, uint32_t size) {
  SurfelRowsRW o = surfel_rows_rw(surfels, active, size);
  o.perm = w.perm;
  if (w.perm) {
    o.x = (float*)w.rows.x; o.y = (float*)w.rows.y; o.z = (float*)w.rows.z;
    o.normal = (uint32_t*)w.rows.normal;
    o.radius_squared = w.rows.radius_squared;
    o.d1 = (float*)w.rows.d1; o.d2 = (float*)w.rows.d2;
  }
  return o;
}

// Chooses how many keyframes one block walks: enough blocks to fill 256 CUs several times over.
static int choose_kfs_per_block(int tiles, int kf_count) {
#ifndef BSLAM_POSE_TARGET_BLOCKS
#define BSLAM_POSE_TARGET_BLOCKS 8192
#endif
  const int target_blocks = BSLAM_POSE_TARGET_BLOCKS;
  int chunks = (target_blocks + tiles - 1) / tiles;
  if (chunks < 1) chunks = 1;
  if (chunks > kf_count) chunks = kf_count;
  int per_block = (kf_count + chunks - 1) / chunks;
  // Large surfel counts give enough blocks with one chunk, but then every block walks the whole keyframe list and the
  // resident blocks spread over all of it; a cap keeps the blocks in flight (chunk-major order) on a few keyframes.
  // (Measured flat optimum 12 ... 32 keyframes with one visit bit per keyframe in a 32-bit word: K = 200: -7 %, K = 300 photometric:
  // -13 % kernel time; with the per-surfel order 32: 15.5 ms, 16: 15.8, 8: 16.2.)  Round 3, 64-bit visit words: where the surfels
  // alone give enough workgroups (>= 8192 work slots) a chunk of up to 64 keyframes halves the number of (slot, chunk) workgroups
  // that only start, test and leave on stacks with most pairs out of view, and costs nothing on the dense one -- K = 300
  // photometric dense 13.38 vs 13.36 ms, trajectory stack 1316 -> 1244 us, survey-range stack 4.59 -> 4.36 ms, K = 1000 geometry-
  // only 26.1 -> 25.4 ms; with fewer slots the longer chunks leave too few workgroups for an even tail (K = 200, 2500 slots:
  // 1149 -> 1187 us), so those keep 32.
#ifndef BSLAM_POSE_MAX_KFS_PER_BLOCK
#define BSLAM_POSE_MAX_KFS_PER_BLOCK 0   /* 0: 64 with >= 8192 work slots, else 32 */
#endif
  const int cap = BSLAM_POSE_MAX_KFS_PER_BLOCK > 0 ? BSLAM_POSE_MAX_KFS_PER_BLOCK : (tiles >= 8192 ? 64 : 32);
  if (per_block > cap) per_block = cap;
  return per_block;
}

// Device counters of the cull statistics (bslam_debug_cull_stats), maintained while profiling is on.
static unsigned long long* cull_stats_ptr(bslam_context* ctx) {
  return ctx->profiling ? (unsigned long long*)((uint8_t*)ctx->misc.ptr + kMiscCullStats) : nullptr;
}

// `work`: the prepared schedule + rows of this API call (prepare_surfels; the batched Gauss-Newton loop prepares them once for
// all its iterations), or nullptr to prepare them here.
static int launch_pose_accumulate(bslam_context* ctx, hipStream_t stream, int use_depth, int use_desc, const CamConsts& c,
                                  int kf_count, uint32_t surfels_size, const bslam_buffer2d* surfels, const PoseState* states,
                                  int* tiles_out, bool reduce_rows = true, const SurfelWork* work = nullptr, int* kfs_per_block_out = nullptr,
                                  bool want_cost = false, const int* kf_list = nullptr, int kf_list_bound = 0) {
  SurfelWork local;
  int rc = BSLAM_OK;
  if (!work) {
    if ((rc = prepare_surfels(ctx, stream, surfels, surfels_size, pose_surfels_per_thread(use_desc != 0, surfels_size), kf_count, &local, use_desc != 0))) return rc;
    work = &local;
  }
  Schedule sc = work->sc;
  // A block of this kernel walks a chunk of <= 32 keyframes of one work slot; on short keyframe lists the chunks are so short
  // (6 keyframes at K = 50) that the culling test at the head of every block costs more than it saves on any stack we have
  // (K = 50 dense, 12 % of the blocks culled: 96 -> 102 us per launch).  The geometry / PCG / intrinsics kernels walk the whole
  // list per block and keep it.
#ifndef BSLAM_POSE_CULL_MIN_KEYFRAMES
#define BSLAM_POSE_CULL_MIN_KEYFRAMES 64
#endif
  if (kf_count < BSLAM_POSE_CULL_MIN_KEYFRAMES) sc.bounds = nullptr;
  const int tiles = (int)sc.slots;
  *tiles_out = tiles;
  const int rows_per_kf = tiles * kPoseRowsPerSlot;   // one partial row per (slot, wave), or per slot
  const size_t partial_floats = (size_t)rows_per_kf * kf_count * kRow;
  rc = ctx->partials.reserve((partial_floats + (size_t)kf_count * kReduceParts * kRow) * sizeof(float));
  if (rc) return rc;
  rc = ctx->coeffs.reserve((size_t)kf_count * kRow * sizeof(float));
  if (rc) return rc;
  const int per_block = choose_kfs_per_block(tiles, kf_count);   // <= 64: one bit per keyframe of a chunk in a visit word
  const unsigned chunks = (unsigned)((kf_count + per_block - 1) / per_block);
  if (kfs_per_block_out) *kfs_per_block_out = per_block;
  if ((rc = ctx->vis.reserve((size_t)chunks * sc.slots * sizeof(VisWord)))) return rc;
  VisWord* vis = (VisWord*)ctx->vis.ptr;
  // kf_list: the launch walks the device-side list of unconverged keyframes (batched loop); kf_list_bound = an upper bound of its
  // length known to the host (the count of an earlier iteration), which sizes the grid -- the visit words keep room for all chunks
  const unsigned launch_chunks = kf_list ? (unsigned)std::max(1, (std::min(kf_list_bound, kf_count) + per_block - 1) / per_block) : chunks;
  dim3 grid(8u * sc.slots_per_xcd * launch_chunks);
  const SurfelRows rows = work->rows;
  const KfDev* kfs = (const KfDev*)ctx->kf_table.ptr;
  float* partials = (float*)ctx->partials.ptr;
  // tuning aid: BSLAM_DEBUG_POSE_LDS = bytes of (unused) dynamic LDS per block, to cap the blocks per CU without touching the code
  static const unsigned debug_lds = getenv("BSLAM_DEBUG_POSE_LDS") ? (unsigned)atoi(getenv("BSLAM_DEBUG_POSE_LDS")) : 0u;
  {
  ProfScope prof(ctx, stream);
  const bool cost = want_cost;
#define BSLAM_LAUNCH_POSE(DEPTH, DESC, R)                                                                                                        \
  do {                                                                                                                                           \
    if (cost) hipLaunchKernelGGL((pose_accumulate_kernel<DEPTH, DESC, R, true>), grid, dim3(kPoseThreads), debug_lds, stream, c, kfs, kf_count, per_block, sc, rows, partials, rows_per_kf, states, vis, kf_list);  \
    else hipLaunchKernelGGL((pose_accumulate_kernel<DEPTH, DESC, R, false>), grid, dim3(kPoseThreads), debug_lds, stream, c, kfs, kf_count, per_block, sc, rows, partials, rows_per_kf, states, vis, kf_list); \
  } while (0)
  if (use_depth && use_desc) BSLAM_LAUNCH_POSE(true, true, kPoseRDesc);
  else if (use_depth && pose_surfels_per_thread(false, surfels_size) == kPoseRGeoLarge) BSLAM_LAUNCH_POSE(true, false, kPoseRGeoLarge);
  else if (use_depth) BSLAM_LAUNCH_POSE(true, false, kPoseRGeo);
  else BSLAM_LAUNCH_POSE(false, true, kPoseRDesc);
#undef BSLAM_LAUNCH_POSE
  }
  BSLAM_HIP_TRY(hipGetLastError());
  if (!reduce_rows) return BSLAM_OK;   // the caller's pose_reduce_solve_kernel sums the rows itself
  // sums of counts are formed per row as floats: a (slot, wave) row holds <= 64 * kPoseR residuals, a thread's
  // share at most rows / 32 * 256 -- exact in fp32 up to 2^24
  hipLaunchKernelGGL(pose_reduce_rows_kernel, dim3((unsigned)kf_count), dim3(1024), (unsigned)visit_map_bytes(tiles), stream, (const float*)partials, rows_per_kf, kf_count,
                     (float*)ctx->coeffs.ptr, states, (const VisWord*)vis, per_block, cull_stats_ptr(ctx), kf_list ? kf_list + 1 + kf_count : nullptr);
  BSLAM_HIP_TRY(hipGetLastError());
  return BSLAM_OK;
}

}  // namespace bslam

// ---- native exchange: an RCCL communicator owned by the context ------------------------------------------------------
// librccl is opened at run time (dlopen) the first time a communicator is asked for: the kernel library itself has no link
// dependency on it, a single-GPU user never loads it, and a process that already holds an RCCL (e.g. the one PyTorch ships)
// shares that copy -- dlopen by soname returns the loaded library.
namespace {
using bslam::fail;
struct RcclId128 { char b[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
constexpr int kRcclFloat32 = 7, kRcclSum = 0;
}  // namespace
// The prototypes below are declared by hand so that the library carries no build dependency on RCCL; where its header is
// installed the assumptions are checked against it at compile time.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert(sizeof(ncclUniqueId) == sizeof(RcclId128), "ncclUniqueId is not 128 bytes");
static_assert((int)ncclFloat32 == kRcclFloat32 && (int)ncclSum == kRcclSum, "RCCL enum values changed");
static_assert(sizeof(ncclComm_t) == sizeof(void*), "ncclComm_t is not a pointer");
#endif
namespace {
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId128, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommUserRank)(void*, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int load_rccl() {
  if (g_rccl.handle) return BSLAM_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
  if (!h) return fail(BSLAM_ERR_HIP, "RCCL not found (dlopen librccl.so.1: %s)", dlerror());
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
  g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))dlsym(h, "ncclCommUserRank");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce) {
    dlclose(h);
    return fail(BSLAM_ERR_HIP, "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
  }
  g_rccl.handle = h;
  return BSLAM_OK;
}
const char* rccl_error(int e) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?"; }
}  // namespace

namespace bslam {
int rccl_allreduce_sum(bslam_context* ctx, hipStream_t stream, float* device_buffer, size_t count) {
  if (count == 0) return BSLAM_OK;
  // in place, on the caller's stream: ordered after the producers the library enqueued
  const int e = g_rccl.AllReduce(device_buffer, device_buffer, count, kRcclFloat32, kRcclSum, ctx->comm, stream);
  if (e != 0) return fail(BSLAM_ERR_HIP, "ncclAllReduce(%zu floats) failed: %s", count, rccl_error(e));
  return BSLAM_OK;
}
}  // namespace bslam

using namespace bslam;

extern "C" {

int bslam_abi_version(void) { return 1; }

const char* bslam_last_error(void) { return g_last_error.c_str(); }

int bslam_create(int device, bslam_context** out_ctx) {
  if (!out_ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "out_ctx is null");
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) return fail(BSLAM_ERR_NO_DEVICE, "no HIP device visible (%s); this library has no CPU fallback", hipGetErrorString(e));
  if (device < 0 || device >= count) return fail(BSLAM_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device, count);
  BSLAM_HIP_TRY(hipSetDevice(device));
  bslam_context* ctx = new (std::nothrow) bslam_context();
  if (!ctx) return fail(BSLAM_ERR_OUT_OF_MEMORY, "out of host memory");
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
  int rc = ctx->misc.reserve(256);
  if (rc) { delete ctx; return rc; }
  if (hipMemset(ctx->misc.ptr, 0, 256) != hipSuccess) { delete ctx; return fail(BSLAM_ERR_HIP, "hipMemset failed"); }
  if (const char* e = getenv("BSLAM_CULLING")) ctx->culling = atoi(e) != 0;   // A/B runs of whole programs (bslam_set_culling otherwise)
  *out_ctx = ctx;
  return BSLAM_OK;
}

int bslam_destroy(bslam_context* ctx) {
  if (!ctx) return BSLAM_OK;
  hipError_t e = hipSetDevice(ctx->device); (void)e;
  bslam_comm_destroy(ctx);
  ctx->kf_table.release(); ctx->partials.release(); ctx->coeffs.release(); ctx->pose_state.release(); ctx->misc.release(); ctx->records.release(); ctx->quads.release(); ctx->exchange.release(); ctx->lifecycle.release(); ctx->quads_aux.release(); ctx->order.release(); ctx->centroids.release(); ctx->perm.release(); ctx->sorted_rows.release(); ctx->bounds.release(); ctx->vis.release(); ctx->intr_cells.release(); ctx->prof_counters.release();
  ctx->staging.release(); ctx->staging2.release(); ctx->upload_ring.release();
  for (hipEvent_t& e : ctx->iter_done) if (e) { hipError_t err = hipEventDestroy(e); (void)err; e = nullptr; }
  for (hipEvent_t& e : ctx->solve_done) if (e) { hipError_t err = hipEventDestroy(e); (void)err; e = nullptr; }
  if (ctx->copy_stream) { hipError_t err = hipStreamDestroy(ctx->copy_stream); (void)err; ctx->copy_stream = nullptr; }
  if (ctx->perm_ready) { hipError_t err = hipEventDestroy(ctx->perm_ready); (void)err; ctx->perm_ready = nullptr; }
  for (auto& ev : ctx->prof_pending) ctx->prof_pool.push_back(std::make_pair(ev.start, ev.stop));
  for (auto& ev : ctx->prof_pool) { hipError_t e1 = hipEventDestroy(ev.first); e1 = hipEventDestroy(ev.second); (void)e1; }
  delete ctx;
  return BSLAM_OK;
}

int bslam_set_texture_mode(bslam_context* ctx, int mode) {
  if (!ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "context is null");
  if (mode != BSLAM_TEX_FIXED_POINT_1_8 && mode != BSLAM_TEX_EXACT_FLOAT) return fail(BSLAM_ERR_INVALID_ARGUMENT, "unknown texture mode %d", mode);
  ctx->tex_mode = mode;
  return BSLAM_OK;
}

int bslam_set_allreduce(bslam_context* ctx, bslam_allreduce_fn allreduce, void* allreduce_user) {
  if (!ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "context is null");
  ctx->allreduce = allreduce;
  ctx->allreduce_user = allreduce_user;
  return BSLAM_OK;
}

int bslam_comm_get_unique_id(void* out_id, size_t bytes) {
  if (!out_id || bytes < BSLAM_COMM_UNIQUE_ID_BYTES) return fail(BSLAM_ERR_INVALID_ARGUMENT, "need a buffer of %d bytes", BSLAM_COMM_UNIQUE_ID_BYTES);
  int rc = load_rccl();
  if (rc) return rc;
  const int e = g_rccl.GetUniqueId(out_id);
  if (e != 0) return fail(BSLAM_ERR_HIP, "ncclGetUniqueId failed: %s", rccl_error(e));
  return BSLAM_OK;
}

int bslam_comm_init(bslam_context* ctx, const void* unique_id, int rank, int world_size) {
  if (!ctx || !unique_id) return fail(BSLAM_ERR_INVALID_ARGUMENT, "null argument");
  if (world_size < 1 || rank < 0 || rank >= world_size) return fail(BSLAM_ERR_INVALID_ARGUMENT, "rank %d of %d", rank, world_size);
  if (ctx->comm) return fail(BSLAM_ERR_INVALID_ARGUMENT, "the context already has a communicator (bslam_comm_destroy first)");
  int rc = load_rccl();
  if (rc) return rc;
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));
  RcclId128 id;
  std::memcpy(id.b, unique_id, sizeof(id.b));
  void* comm = nullptr;
  const int e = g_rccl.CommInitRank(&comm, world_size, id, rank);
  if (e != 0) return fail(BSLAM_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world_size, rccl_error(e));
  ctx->comm = comm;
  ctx->comm_rank = rank;
  ctx->comm_world = world_size;
  return BSLAM_OK;
}

int bslam_comm_query(bslam_context* ctx, int* rank, int* world_size) {
  if (!ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "context is null");
  int r = 0, w = 1;
  if (ctx->comm) {
    // read back from the communicator itself (ncclCommUserRank / ncclCommCount), not from what bslam_comm_init was told
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(BSLAM_ERR_HIP, "librccl lacks ncclCommCount / ncclCommUserRank");
    int e = g_rccl.CommCount(ctx->comm, &w);
    if (e == 0) e = g_rccl.CommUserRank(ctx->comm, &r);
    if (e != 0) return fail(BSLAM_ERR_HIP, "ncclCommCount / ncclCommUserRank failed: %s", rccl_error(e));
  }
  if (rank) *rank = r;
  if (world_size) *world_size = w;
  return BSLAM_OK;
}

int bslam_comm_destroy(bslam_context* ctx) {
  if (!ctx) return BSLAM_OK;
  if (ctx->comm) {
    hipError_t he = hipSetDevice(ctx->device); (void)he;
    const int e = g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 1; ctx->comm_rank = 0;
    if (e != 0) return fail(BSLAM_ERR_HIP, "ncclCommDestroy failed: %s", rccl_error(e));
  }
  return BSLAM_OK;
}

int bslam_set_keyframe_cache(bslam_context* ctx, int enable) {
  if (!ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "context is null");
  ctx->keyframe_cache = enable != 0;
  ctx->records_signature.clear();
  ctx->quads_signature.clear();
  return BSLAM_OK;
}

int bslam_invalidate_keyframe_cache(bslam_context* ctx) {
  if (!ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "context is null");
  ctx->records_signature.clear();
  ctx->quads_signature.clear();
  return BSLAM_OK;
}

int bslam_set_xcd_schedule(bslam_context* ctx, int enable) {
  if (!ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "context is null");
  ctx->use_schedule = enable != 0;
  ctx->order_key_ptr = nullptr;
  ctx->perm_key_ptr = nullptr;
  return BSLAM_OK;
}

int bslam_set_culling(bslam_context* ctx, int enable) {
  if (!ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "context is null");
  ctx->culling = enable != 0;
  return BSLAM_OK;
}

int bslam_set_pose_keyframe_list(bslam_context* ctx, int min_keyframes) {
  if (!ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "null context");
  if (min_keyframes < 0) return fail(BSLAM_ERR_INVALID_ARGUMENT, "min_keyframes must be >= 0");
  ctx->pose_list_min_keyframes = min_keyframes;
  return BSLAM_OK;
}

int bslam_debug_cull_stats(bslam_context* ctx, uint64_t* tested, uint64_t* culled) {
  if (!ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "context is null");
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));
  BSLAM_HIP_TRY(hipDeviceSynchronize());
  unsigned long long h[2] = {0, 0};
  BSLAM_HIP_TRY(hipMemcpy(h, (uint8_t*)ctx->misc.ptr + kMiscCullStats, sizeof(h), hipMemcpyDeviceToHost));
  BSLAM_HIP_TRY(hipMemset((uint8_t*)ctx->misc.ptr + kMiscCullStats, 0, sizeof(h)));
  if (tested) *tested = h[0];
  if (culled) *culled = h[0] - h[1];
  return BSLAM_OK;
}

int bslam_profile_enable(bslam_context* ctx, int enable) {
  if (!ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "context is null");
  ctx->profiling = enable != 0;
  for (auto& ev : ctx->prof_pending) ctx->prof_pool.push_back(std::make_pair(ev.start, ev.stop));
  ctx->prof_pending.clear();
  for (int t = 0; t < 8; ++t) { ctx->prof_launches[t] = 0; ctx->prof_ms[t] = 0.f; }
  for (int i = 0; i < 8; ++i) ctx->prof_counter_base[i] = 0;
  if (ctx->prof_counters.ptr && ctx->prof_counter_slots) {
    BSLAM_HIP_TRY(hipSetDevice(ctx->device));
    BSLAM_HIP_TRY(hipDeviceSynchronize());
    BSLAM_HIP_TRY(hipMemset(ctx->prof_counters.ptr, 0, ctx->prof_counter_slots * sizeof(unsigned long long)));
  }
  return BSLAM_OK;
}

int bslam_profile_read_counters(bslam_context* ctx, uint64_t* counters8) {
  if (!ctx || !counters8) return fail(BSLAM_ERR_INVALID_ARGUMENT, "null argument");
  for (int i = 0; i < 8; ++i) counters8[i] = ctx->prof_counter_base[i];
  if (!ctx->prof_counters.ptr || !ctx->prof_counter_slots) return BSLAM_OK;
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));
  BSLAM_HIP_TRY(hipDeviceSynchronize());
  std::vector<unsigned long long> host(ctx->prof_counter_slots);   // [block][2]: summed here, not with contended atomics on the device
  BSLAM_HIP_TRY(hipMemcpy(host.data(), ctx->prof_counters.ptr, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < host.size(); ++i) counters8[i & 1] += host[i];
  return BSLAM_OK;
}

int bslam_profile_read(bslam_context* ctx, int kernel, int32_t* launches, float* total_ms) {
  if (!ctx) return fail(BSLAM_ERR_INVALID_ARGUMENT, "context is null");
  if (kernel < 0 || kernel >= 8) return fail(BSLAM_ERR_INVALID_ARGUMENT, "unknown kernel tag %d", kernel);
  for (auto& ev : ctx->prof_pending) {
    BSLAM_HIP_TRY(hipEventSynchronize(ev.stop));
    float ms = 0.f;
    BSLAM_HIP_TRY(hipEventElapsedTime(&ms, ev.start, ev.stop));
    ctx->prof_ms[ev.tag] += ms;
    ctx->prof_launches[ev.tag] += 1;
    ctx->prof_pool.push_back(std::make_pair(ev.start, ev.stop));
  }
  ctx->prof_pending.clear();
  if (launches) *launches = ctx->prof_launches[kernel];
  if (total_ms) *total_ms = ctx->prof_ms[kernel];
  ctx->prof_launches[kernel] = 0;
  ctx->prof_ms[kernel] = 0.f;
  return BSLAM_OK;
}

int bslam_set_geometry_keyframe_chunk(bslam_context* ctx, int keyframes_per_launch) {
  if (!ctx || keyframes_per_launch < -1) return fail(BSLAM_ERR_INVALID_ARGUMENT, "bad argument");
  ctx->geom_kf_chunk = keyframes_per_launch;
  return BSLAM_OK;
}

int bslam_assign_colors(
    bslam_context* ctx, void* stream_, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
    const bslam_depth_params* depth_params, int keyframe_count, const bslam_keyframe_view* keyframes, uint32_t surfels_size,
    const bslam_buffer2d* surfels) {
  hipStream_t stream = (hipStream_t)stream_;
  if (surfels_size == 0) return BSLAM_OK;   // BS/kernel_assign_colors.cc:49-51
  int rc = check_common(ctx, depth_camera, depth_params, surfels);
  if (rc) return rc;
  if (!color_camera) return fail(BSLAM_ERR_INVALID_ARGUMENT, "color_camera is null");
  if (keyframe_count < 0 || (keyframe_count > 0 && !keyframes)) return fail(BSLAM_ERR_INVALID_ARGUMENT, "bad keyframe list");
  if (surfels_size > (uint32_t)surfels->width) return fail(BSLAM_ERR_INVALID_ARGUMENT, "surfels_size %u exceeds the buffer width %d", surfels_size, surfels->width);
  if (surfels->height <= BSLAM_SURFEL_COLOR) return fail(BSLAM_ERR_INVALID_ARGUMENT, "the surfel buffer has no colour row");
  if (keyframe_count == 0) return BSLAM_OK;
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));
  std::vector<KfDev> table;
  if ((rc = build_kf_table(depth_camera, color_camera, true, keyframe_count, keyframes, &table))) return rc;
  const CamConsts c = make_cam_consts(ctx, color_camera, depth_camera, depth_params);
  if ((rc = upload_kf_table(ctx, stream, table, c))) return rc;
  uint32_t* color_row = (uint32_t*)((uint8_t*)surfels->address + (size_t)BSLAM_SURFEL_COLOR * surfels->pitch);
  hipLaunchKernelGGL(assign_colors_kernel, dim3((surfels_size + 255) / 256), dim3(256), 0, stream, c, (const KfDev*)ctx->kf_table.ptr, keyframe_count,
                     surfel_rows_rw(surfels, nullptr, surfels_size), color_row);
  BSLAM_HIP_TRY(hipGetLastError());
  return BSLAM_OK;
}

int bslam_debug_decode_normals(bslam_context* ctx, void* stream_, float* out_xyz) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ctx || !out_xyz) return fail(BSLAM_ERR_INVALID_ARGUMENT, "null argument");
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)65536 * 3 * sizeof(float);
  int rc = ctx->coeffs.reserve(bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(decode_normals_kernel, dim3(256), dim3(256), 0, stream, (float*)ctx->coeffs.ptr);
  BSLAM_HIP_TRY(hipGetLastError());
  BSLAM_HIP_TRY(hipMemcpyAsync(out_xyz, ctx->coeffs.ptr, bytes, hipMemcpyDeviceToHost, stream));
  BSLAM_HIP_TRY(hipStreamSynchronize(stream));
  return BSLAM_OK;
}

int bslam_debug_wave_column_sums(bslam_context* ctx, void* stream_, int live_columns, int columns_per_round, const float* in, float* out) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ctx || !in || !out) return fail(BSLAM_ERR_INVALID_ARGUMENT, "null argument");
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));
  const size_t in_bytes = 64 * kRow * sizeof(float), out_bytes = kRow * sizeof(float);
  int rc = ctx->coeffs.reserve(in_bytes + out_bytes);
  if (rc) return rc;
  float* d_in = (float*)ctx->coeffs.ptr;
  float* d_out = (float*)((uint8_t*)ctx->coeffs.ptr + in_bytes);
  BSLAM_HIP_TRY(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, stream));
  BSLAM_HIP_TRY(hipMemcpyAsync(d_out, out, out_bytes, hipMemcpyHostToDevice, stream));
#define BSLAM_PROBE(LIVE, COLS) \
  if (live_columns == LIVE && columns_per_round == COLS) hipLaunchKernelGGL((wave_column_sums_probe_kernel<LIVE, COLS>), dim3(1), dim3(64), 0, stream, (const float*)d_in, d_out); else
  // the configurations the kernels use: pose rows (27 live columns, 28 with the cost) in rounds of 4 (photometric) and 8
  // (geometry-only), the PCG kernels' 6 / 12 pose entries in rounds of 4
  BSLAM_PROBE(27, 4) BSLAM_PROBE(28, 4) BSLAM_PROBE(27, 8) BSLAM_PROBE(28, 8) BSLAM_PROBE(6, 4) BSLAM_PROBE(12, 4)
  return fail(BSLAM_ERR_INVALID_ARGUMENT, "no kernel sums %d columns in rounds of %d", live_columns, columns_per_round);
#undef BSLAM_PROBE
  BSLAM_HIP_TRY(hipGetLastError());
  BSLAM_HIP_TRY(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, stream));
  BSLAM_HIP_TRY(hipStreamSynchronize(stream));
  return BSLAM_OK;
}

int bslam_debug_jacobians(bslam_context* ctx, void* stream_, int kind, int count, const float* in, float* out) {
  hipStream_t stream = (hipStream_t)stream_;
  static const int kIn[8] = {10, 1, 21, 11, 14, 8, 10, 1}, kOut[8] = {7, 1, 7, 9, 1, 4, 7, 2};
  if (!ctx || !in || !out) return fail(BSLAM_ERR_INVALID_ARGUMENT, "null argument");
  if (kind < 0 || kind > 7) return fail(BSLAM_ERR_INVALID_ARGUMENT, "unknown probe kind %d", kind);
  if (count <= 0) return BSLAM_OK;
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));
  const size_t in_bytes = (size_t)count * kIn[kind] * sizeof(float), out_bytes = (size_t)count * kOut[kind] * sizeof(float);
  int rc = ctx->coeffs.reserve(in_bytes + out_bytes);
  if (rc) return rc;
  float* d_in = (float*)ctx->coeffs.ptr;
  float* d_out = (float*)((uint8_t*)ctx->coeffs.ptr + in_bytes);
  BSLAM_HIP_TRY(hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(jacobian_probe_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, kind, count, kIn[kind], kOut[kind], (const float*)d_in, d_out);
  BSLAM_HIP_TRY(hipGetLastError());
  BSLAM_HIP_TRY(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, stream));
  BSLAM_HIP_TRY(hipStreamSynchronize(stream));
  return BSLAM_OK;
}

int bslam_debug_count_pairs(
    bslam_context* ctx, void* stream_, const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes, uint32_t surfels_size, const bslam_buffer2d* surfels,
    uint64_t* in_bounds_pairs, uint64_t* associated_pairs) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common(ctx, depth_camera, depth_params, surfels);
  if (rc) return rc;
  if (keyframe_count <= 0 || !keyframes) return fail(BSLAM_ERR_INVALID_ARGUMENT, "need at least one keyframe");
  if (surfels_size == 0 || surfels_size > (uint32_t)surfels->width) return fail(BSLAM_ERR_INVALID_ARGUMENT, "bad surfels_size %u", surfels_size);
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));
  std::vector<KfDev> table;
  if ((rc = build_kf_table(depth_camera, nullptr, false, keyframe_count, keyframes, &table))) return rc;
  const CamConsts c = make_cam_consts(ctx, nullptr, depth_camera, depth_params);
  if ((rc = upload_kf_table(ctx, stream, table, c))) return rc;
  unsigned long long* d_out = (unsigned long long*)((uint8_t*)ctx->misc.ptr + 64);
  BSLAM_HIP_TRY(hipMemsetAsync(d_out, 0, 2 * sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(count_pairs_kernel, dim3((surfels_size + 255) / 256), dim3(256), 0, stream, c, (const KfDev*)ctx->kf_table.ptr, keyframe_count,
                     surfel_rows_rw(surfels, nullptr, surfels_size), d_out);
  BSLAM_HIP_TRY(hipGetLastError());
  if ((rc = ctx->staging2.reserve(64))) return rc;
  BSLAM_HIP_TRY(hipMemcpyAsync(ctx->staging2.ptr, d_out, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
  BSLAM_HIP_TRY(hipStreamSynchronize(stream));
  const unsigned long long* o = (const unsigned long long*)ctx->staging2.ptr;
  if (in_bounds_pairs) *in_bounds_pairs = o[0];
  if (associated_pairs) *associated_pairs = o[1];
  return BSLAM_OK;
}

int bslam_accumulate_pose_estimation_coeffs(
    bslam_context* ctx, void* stream_, int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    const bslam_buffer2d* depth_buffer, const bslam_buffer2d* normals_buffer, const bslam_buffer2d* color_buffer,
    const bslam_mat3x4* frame_T_global_estimate, uint32_t surfels_size, const bslam_buffer2d* surfels,
    int debug, uint32_t* residual_count, float* residual_sum, float* H, float* b) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common(ctx, depth_camera, depth_params, surfels);
  if (rc) return rc;
  // CHECK(use_depth_residuals || use_descriptor_residuals); CHECK_GT(surfels_size, 0)  (BS/kernel_opt_pose.cc:58-61)
  if (!use_depth_residuals && !use_descriptor_residuals) return fail(BSLAM_ERR_INVALID_ARGUMENT, "need depth and/or descriptor residuals");
  if (surfels_size == 0) return fail(BSLAM_ERR_INVALID_ARGUMENT, "surfels_size must be > 0");
  if (surfels_size > (uint32_t)surfels->width) return fail(BSLAM_ERR_INVALID_ARGUMENT, "surfels_size %u exceeds the buffer width %d", surfels_size, surfels->width);
  if (!H || !b || !frame_T_global_estimate || !color_camera) return fail(BSLAM_ERR_INVALID_ARGUMENT, "null argument");
  if ((rc = check_image(depth_buffer, depth_camera, 2, "depth"))) return rc;
  if ((rc = check_image(normals_buffer, depth_camera, 2, "normals"))) return rc;
  if (use_descriptor_residuals && (rc = check_image(color_buffer, color_camera, 4, "color"))) return rc;
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));

  std::vector<KfDev> table(1);
  fill_kf(&table[0], depth_buffer, normals_buffer, use_descriptor_residuals ? color_buffer : nullptr, frame_T_global_estimate, nullptr, BSLAM_KF_ACTIVE, 0);
  const CamConsts c = make_cam_consts(ctx, color_camera, depth_camera, depth_params);
  if ((rc = upload_kf_table(ctx, stream, table, c))) return rc;
  int tiles = 0;
  if ((rc = launch_pose_accumulate(ctx, stream, use_depth_residuals, use_descriptor_residuals, c, 1, surfels_size, surfels, nullptr, &tiles, true, nullptr, nullptr, debug != 0))) return rc;
  if ((rc = ctx->staging2.reserve(kRow * sizeof(float)))) return rc;
  BSLAM_HIP_TRY(hipMemcpyAsync(ctx->staging2.ptr, ctx->coeffs.ptr, kRow * sizeof(float), hipMemcpyDeviceToHost, stream));
  BSLAM_HIP_TRY(hipStreamSynchronize(stream));   // results valid on return, BS/kernel_opt_pose.cc:96
  const float* out = (const float*)ctx->staging2.ptr;
  std::memcpy(H, out, 21 * sizeof(float));
  std::memcpy(b, out + 21, 6 * sizeof(float));
  if (debug) {
    if (residual_sum) *residual_sum = out[kRowCost];
    if (residual_count) *residual_count = (uint32_t)out[kRowCount] + ((uint32_t)out[kRowCount + 1] << 16);
  }
  return BSLAM_OK;
}

int bslam_accumulate_pose_coeffs_batched(
    bslam_context* ctx, void* stream_, int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes, uint32_t surfels_size, const bslam_buffer2d* surfels,
    float* Hb, uint32_t* counts) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common(ctx, depth_camera, depth_params, surfels);
  if (rc) return rc;
  if (!use_depth_residuals && !use_descriptor_residuals) return fail(BSLAM_ERR_INVALID_ARGUMENT, "need depth and/or descriptor residuals");
  if (keyframe_count <= 0 || !keyframes || !color_camera) return fail(BSLAM_ERR_INVALID_ARGUMENT, "need at least one keyframe");
  if (surfels_size == 0) return fail(BSLAM_ERR_INVALID_ARGUMENT, "surfels_size must be > 0");
  if (surfels_size > (uint32_t)surfels->width) return fail(BSLAM_ERR_INVALID_ARGUMENT, "surfels_size %u exceeds the buffer width %d", surfels_size, surfels->width);
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));
  std::vector<KfDev> table;
  if ((rc = build_kf_table(depth_camera, color_camera, use_descriptor_residuals != 0, keyframe_count, keyframes, &table))) return rc;
  const CamConsts c = make_cam_consts(ctx, color_camera, depth_camera, depth_params);
  if ((rc = upload_kf_table(ctx, stream, table, c))) return rc;
  int tiles = 0;
  if ((rc = launch_pose_accumulate(ctx, stream, use_depth_residuals, use_descriptor_residuals, c, keyframe_count, surfels_size, surfels, nullptr, &tiles))) return rc;
  if (Hb || counts) {
    const size_t bytes = (size_t)keyframe_count * kRow * sizeof(float);
    if ((rc = ctx->staging2.reserve(bytes))) return rc;
    BSLAM_HIP_TRY(hipMemcpyAsync(ctx->staging2.ptr, ctx->coeffs.ptr, bytes, hipMemcpyDeviceToHost, stream));
    BSLAM_HIP_TRY(hipStreamSynchronize(stream));
    const float* out = (const float*)ctx->staging2.ptr;
    for (int k = 0; k < keyframe_count; ++k) {
      if (Hb) std::memcpy(Hb + 27 * (size_t)k, out + (size_t)k * kRow, 27 * sizeof(float));
      if (counts) counts[k] = (uint32_t)out[(size_t)k * kRow + kRowCount] + ((uint32_t)out[(size_t)k * kRow + kRowCount + 1] << 16);
    }
  }
  return BSLAM_OK;
}

int bslam_estimate_frame_poses_batched(
    bslam_context* ctx, void* stream_, int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes, uint32_t surfels_size, const bslam_buffer2d* surfels,
    int max_iterations, bslam_se3f* poses, int32_t* iterations_done, int32_t* converged,
    bslam_allreduce_fn allreduce, void* allreduce_user) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common(ctx, depth_camera, depth_params, surfels);
  if (rc) return rc;
  if (!use_depth_residuals && !use_descriptor_residuals) return fail(BSLAM_ERR_INVALID_ARGUMENT, "need depth and/or descriptor residuals");
  if (keyframe_count <= 0 || !keyframes || !poses || !color_camera) return fail(BSLAM_ERR_INVALID_ARGUMENT, "need at least one keyframe and its pose");
  if (surfels_size > (uint32_t)surfels->width) return fail(BSLAM_ERR_INVALID_ARGUMENT, "surfels_size %u exceeds the buffer width %d", surfels_size, surfels->width);
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));

  std::vector<KfDev> table;
  if ((rc = build_kf_table(depth_camera, color_camera, use_descriptor_residuals != 0, keyframe_count, keyframes, &table))) return rc;
  // pose state: keyframes that are INACTIVE are not optimised (BS/direct_ba_alternating.cc:549-552)
  std::vector<PoseState> states((size_t)keyframe_count);
  for (int k = 0; k < keyframe_count; ++k) {
    PoseState& st = states[k];
    std::memset(&st, 0, sizeof(st));
    std::memcpy(st.q, poses[k].q, sizeof(float) * 4);
    std::memcpy(st.t, poses[k].t, sizeof(float) * 3);
    st.converged = (keyframes[k].activation == BSLAM_KF_INACTIVE) ? 1 : 0;
  }
  const CamConsts c = make_cam_consts(ctx, color_camera, depth_camera, depth_params);
  if ((rc = upload_kf_table(ctx, stream, table, c))) return rc;
  const size_t state_bytes = states.size() * sizeof(PoseState);
  if ((rc = ctx->pose_state.reserve(state_bytes))) return rc;
  if ((rc = ctx->staging2.reserve(state_bytes + 64))) return rc;
  {
    void* stage = nullptr;
    if ((rc = ctx->upload_ring.acquire(state_bytes, &stage))) return rc;
    std::memcpy(stage, states.data(), state_bytes);
    BSLAM_HIP_TRY(hipMemcpyAsync(ctx->pose_state.ptr, stage, state_bytes, hipMemcpyHostToDevice, stream));
    if ((rc = ctx->upload_ring.commit(stream))) return rc;
  }

  PoseState* d_states = (PoseState*)ctx->pose_state.ptr;
  int* d_active = (int*)ctx->misc.ptr;                                     // [4]: one counter per in-flight iteration
  int* h_active = (int*)((uint8_t*)ctx->staging2.ptr + state_bytes);       // [4]
  for (hipEvent_t& e : ctx->iter_done)
    if (!e) BSLAM_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (hipEvent_t& e : ctx->solve_done)
    if (!e) BSLAM_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (!ctx->copy_stream) BSLAM_HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  ctx->kf_table_host_ptr = nullptr;   // the kernels below rewrite frame_T_global in the device copy of the table
  hipLaunchKernelGGL(pose_init_kernel, dim3((unsigned)((keyframe_count + 63) / 64)), dim3(64), 0, stream, keyframe_count,
                     (const PoseState*)d_states, (KfDev*)ctx->kf_table.ptr, d_active);
  BSLAM_HIP_TRY(hipGetLastError());

  SurfelWork work;   // schedule + (sorted) surfel rows: the surfels do not change during the loop
  if (surfels_size > 0 && (rc = prepare_surfels(ctx, stream, surfels, surfels_size, pose_surfels_per_thread(use_descriptor_residuals != 0, surfels_size), keyframe_count, &work, use_descriptor_residuals != 0))) return rc;

  // Long keyframe lists: the accumulation walks a device-side list of the keyframes that are still unconverged, rebuilt behind
  // every solve (pose_active_list_kernel: one small block between two launches that take milliseconds), so that its chunks -- and
  // with them the number of workgroups of a launch -- shrink with the work that is left.  On a sequence most keyframes converge
  // in two or three iterations and a few stragglers run to the cap: K = 1000 on the trajectory stack spends 28 of its 30 launches
  // on fewer than 50 keyframes.  Short lists (the per-iteration kernel would cost more than it saves) walk the table itself
  // (bslam_set_pose_keyframe_list: from 64 keyframes on by default).
  int* d_list = nullptr;
  if (surfels_size > 0 && ctx->pose_list_min_keyframes > 0 && keyframe_count >= ctx->pose_list_min_keyframes) {
    if ((rc = ctx->pose_list.reserve((size_t)(1 + 2 * keyframe_count) * sizeof(int)))) return rc;
    d_list = (int*)ctx->pose_list.ptr;
    hipLaunchKernelGGL(pose_active_list_kernel, dim3(1), dim3(kActiveListThreads), 0, stream, keyframe_count, (const PoseState*)d_states, d_list);
    BSLAM_HIP_TRY(hipGetLastError());
  }
  int known_active = keyframe_count;   // upper bound of the list's length: the count the host has read back last (lags the device)

  // One Gauss-Newton iteration of all unconverged keyframes, ending with the number of keyframes still
  // unconverged on its way to h_active[it % 4].
  auto enqueue_iteration = [&](int it) -> int {
    const int slot = it & 3;
    const bool exchange = allreduce != nullptr || has_exchange(ctx);   // the call's own hook, else the context's hook / RCCL communicator
    const bool fused = surfels_size > 0 && !exchange;
    // Slot 0 was zeroed by pose_init_kernel, later slots are zeroed by the previous iteration's solve kernel.
    if (fused) {
      int tiles = 0, per_block = 1;
      int r = launch_pose_accumulate(ctx, stream, use_depth_residuals, use_descriptor_residuals, c, keyframe_count, surfels_size, surfels, d_states, &tiles, false, &work, &per_block,
                                     false, d_list, known_active);
      if (r) return r;
      {
        ProfScope prof(ctx, stream, BSLAM_PROF_POSE_REDUCE);
        hipLaunchKernelGGL(pose_reduce_solve_kernel, dim3((unsigned)keyframe_count), dim3(kReduceSolveThreads), (unsigned)visit_map_bytes(tiles), stream, (const float*)ctx->partials.ptr,
                           tiles * kPoseRowsPerSlot, keyframe_count, d_states, (KfDev*)ctx->kf_table.ptr, d_active + slot, d_active + ((it + 1) & 3),
                           (const VisWord*)ctx->vis.ptr, per_block, cull_stats_ptr(ctx), d_list ? d_list + 1 + keyframe_count : nullptr);
      }
      BSLAM_HIP_TRY(hipGetLastError());
    } else {
      if (surfels_size > 0) {
        int tiles = 0;
        int r = launch_pose_accumulate(ctx, stream, use_depth_residuals, use_descriptor_residuals, c, keyframe_count, surfels_size, surfels, d_states, &tiles, true, &work, nullptr,
                                       false, d_list, known_active);
        if (r) return r;
      } else {
        int r = ctx->coeffs.reserve((size_t)keyframe_count * kRow * sizeof(float));
        if (r) return r;
        BSLAM_HIP_TRY(hipMemsetAsync(ctx->coeffs.ptr, 0, (size_t)keyframe_count * kRow * sizeof(float), stream));   // H.setZero(); b.setZero() (:147-149)
      }
      {
        ProfScope prof(ctx, stream, BSLAM_PROF_EXCHANGE);
        if (allreduce) {
          // rows of converged keyframes are zeros on every rank; the solve kernel ignores them.
          const int arc = allreduce(allreduce_user, ctx->coeffs.ptr, (size_t)keyframe_count * kRow, stream);
          if (arc) return fail(BSLAM_ERR_HIP, "allreduce callback failed with %d", arc);
        } else if (exchange) {
          const int arc = exchange_sum(ctx, stream, (float*)ctx->coeffs.ptr, (size_t)keyframe_count * kRow);
          if (arc) return arc;
        }
      }
      hipLaunchKernelGGL(pose_solve_kernel, dim3((unsigned)((keyframe_count + 63) / 64)), dim3(64), 0, stream,
                         (const float*)ctx->coeffs.ptr, keyframe_count, d_states, (KfDev*)ctx->kf_table.ptr, d_active + slot, d_active + ((it + 1) & 3));
      BSLAM_HIP_TRY(hipGetLastError());
    }
    // The 4-byte flag travels on a side stream behind the solve kernel: on the BA stream the copy node and its two launch gaps
    // (about 15 us) would sit between this iteration's solve and the next iteration's accumulation.  The slot is rewritten four
    // iterations later at the earliest, by which time the host has waited for this copy.
    BSLAM_HIP_TRY(hipEventRecord(ctx->solve_done[slot], stream));
    BSLAM_HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, ctx->solve_done[slot], 0));
    BSLAM_HIP_TRY(hipMemcpyAsync(h_active + slot, d_active + slot, sizeof(int), hipMemcpyDeviceToHost, ctx->copy_stream));
    BSLAM_HIP_TRY(hipEventRecord(ctx->iter_done[slot], ctx->copy_stream));
    if (d_list) {   // the list the NEXT iteration's accumulation walks (behind the event: the flag does not wait for it)
      hipLaunchKernelGGL(pose_active_list_kernel, dim3(1), dim3(kActiveListThreads), 0, stream, keyframe_count, (const PoseState*)d_states, d_list);
      BSLAM_HIP_TRY(hipGetLastError());
    }
    return BSLAM_OK;
  };
  // The host learns "all converged" one iteration late: iteration it + 1 is already enqueued while the flag of
  // iteration it travels back, so the GPU never idles on the round trip.  An iteration enqueued after
  // convergence is a no-op on the device (every kernel skips converged keyframes), so poses, iteration counts
  // and flags are the same as with a blocking check after every iteration.
  if (max_iterations > 0 && (rc = enqueue_iteration(0))) return rc;
  for (int it = 0; it < max_iterations; ++it) {
    if (it + 1 < max_iterations && (rc = enqueue_iteration(it + 1))) return rc;
    BSLAM_HIP_TRY(hipEventSynchronize(ctx->iter_done[it & 3]));
    if (h_active[it & 3] == 0) break;
    known_active = std::min(known_active, h_active[it & 3]);   // convergence is final: later lists are no longer than this
  }
  BSLAM_HIP_TRY(hipMemcpyAsync(ctx->staging2.ptr, d_states, state_bytes, hipMemcpyDeviceToHost, stream));
  BSLAM_HIP_TRY(hipStreamSynchronize(stream));
  BSLAM_HIP_TRY(hipStreamSynchronize(ctx->copy_stream));   // the flag copy of the last (no-op) iteration writes into staging2 as well
  const PoseState* out = (const PoseState*)ctx->staging2.ptr;
  for (int k = 0; k < keyframe_count; ++k) {
    std::memcpy(poses[k].q, out[k].q, sizeof(float) * 4);
    std::memcpy(poses[k].t, out[k].t, sizeof(float) * 3);
    if (iterations_done) iterations_done[k] = out[k].iterations;
    if (converged) converged[k] = (keyframes[k].activation == BSLAM_KF_INACTIVE) ? 1 : out[k].converged;
  }
  return BSLAM_OK;
}

// ---- activation / geometry ------------------------------------------------------------------

static int geometry_common(bslam_context* ctx, hipStream_t stream, const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera,
                           const bslam_depth_params* dp, bool need_color, int keyframe_count, const bslam_keyframe_view* keyframes,
                           uint32_t surfels_size, const bslam_buffer2d* surfels, const bslam_buffer2d* active) {
  int rc = check_common(ctx, depth_camera, dp, surfels);
  if (rc) return rc;
  if (keyframe_count < 0 || (keyframe_count > 0 && !keyframes)) return fail(BSLAM_ERR_INVALID_ARGUMENT, "bad keyframe list");
  if (!active || !active->address) return fail(BSLAM_ERR_INVALID_ARGUMENT, "active_surfels is null");
  if (surfels_size > (uint32_t)surfels->width || surfels_size > (uint32_t)active->width) return fail(BSLAM_ERR_INVALID_ARGUMENT, "surfels_size %u exceeds a buffer width", surfels_size);
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));
  std::vector<KfDev> table;
  if ((rc = build_kf_table(depth_camera, color_camera, need_color, keyframe_count, keyframes, &table))) return rc;
  if (table.empty()) { table.resize(1); std::memset(&table[0], 0, sizeof(KfDev)); }   // keep the table pointer valid; kf_count = 0 makes every loop empty
  const CamConsts c = make_cam_consts(ctx, color_camera, depth_camera, dp);
  return upload_kf_table(ctx, stream, table, c);
}

int bslam_update_surfel_activation(
    bslam_context* ctx, void* stream_, const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes, uint32_t surfels_size, const bslam_buffer2d* surfels,
    const bslam_buffer2d* active_surfels) {
  hipStream_t stream = (hipStream_t)stream_;
  if (surfels_size == 0) return BSLAM_OK;   // BS/kernel_surfel_activation.cc:48-50
  int rc = geometry_common(ctx, stream, nullptr, depth_camera, depth_params, false, keyframe_count, keyframes, surfels_size, surfels, active_surfels);
  if (rc) return rc;
  const CamConsts c = make_cam_consts(ctx, nullptr, depth_camera, depth_params);
  // A surfel stops at its first associated keyframe.  On a stack where every keyframe sees everything that is about 3 of K
  // visits, which does not pay for a sorted copy of the rows: granule order.  On a trajectory the first keyframes of the list
  // do not see the surfel at all and the walk is long; from kActivationPermMinKeyframes keyframes on the pass therefore runs
  // in the per-surfel order with the block-level frustum culling, which skips the keyframes a work slot cannot see
  // (K = 300 trajectory stack: 1.67 ms -> see DESIGN.md; dense stack: unchanged within noise).
#ifndef BSLAM_ACTIVATION_PERM_MIN_KEYFRAMES
#define BSLAM_ACTIVATION_PERM_MIN_KEYFRAMES 64
#endif
  SurfelWork work;
  const bool per_surfel_order = ctx->culling && keyframe_count >= BSLAM_ACTIVATION_PERM_MIN_KEYFRAMES;
  if ((rc = prepare_surfels(ctx, stream, surfels, surfels_size, 1, per_surfel_order ? keyframe_count : 1, &work, false))) return rc;
  const Schedule sc = work.sc;
  {
    ProfScope prof(ctx, stream, BSLAM_PROF_ACTIVATION);
    const size_t counter_slots = 2 * (size_t)(8u * sc.slots_per_xcd);
    if (ctx->profiling && ctx->prof_counter_slots < counter_slots) {   // (re)size the per-block counter array, keeping what was counted so far
      uint64_t keep[8];
      if ((rc = bslam_profile_read_counters(ctx, keep))) return rc;
      for (int i = 0; i < 8; ++i) ctx->prof_counter_base[i] = keep[i];
      if ((rc = ctx->prof_counters.reserve(counter_slots * sizeof(unsigned long long)))) return rc;
      BSLAM_HIP_TRY(hipMemsetAsync(ctx->prof_counters.ptr, 0, counter_slots * sizeof(unsigned long long), stream));
      ctx->prof_counter_slots = counter_slots;
    }
    if (ctx->profiling && ctx->prof_counters.ptr)
      hipLaunchKernelGGL(activation_kernel<true>, dim3(8u * sc.slots_per_xcd), dim3(256), 0, stream, c, (const KfDev*)ctx->kf_table.ptr, keyframe_count, sc,
                         surfel_rows_rw(work, surfels, active_surfels, surfels_size), (unsigned long long*)ctx->prof_counters.ptr);
    else
      hipLaunchKernelGGL(activation_kernel<false>, dim3(8u * sc.slots_per_xcd), dim3(256), 0, stream, c, (const KfDev*)ctx->kf_table.ptr, keyframe_count, sc,
                         surfel_rows_rw(work, surfels, active_surfels, surfels_size), (unsigned long long*)nullptr);
  }
  BSLAM_HIP_TRY(hipGetLastError());
  return BSLAM_OK;
}

int bslam_update_surfel_normals(
    bslam_context* ctx, void* stream_, const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes, uint32_t surfels_size, const bslam_buffer2d* surfels,
    const bslam_buffer2d* active_surfels) {
  hipStream_t stream = (hipStream_t)stream_;
  if (surfels_size == 0) return BSLAM_OK;   // BS/kernel_opt_geometry.cc:48-50
  int rc = geometry_common(ctx, stream, nullptr, depth_camera, depth_params, false, keyframe_count, keyframes, surfels_size, surfels, active_surfels);
  if (rc) return rc;
  const CamConsts c = make_cam_consts(ctx, nullptr, depth_camera, depth_params);
  SurfelWork work;
  if ((rc = prepare_surfels(ctx, stream, surfels, surfels_size, 1, keyframe_count, &work, false))) return rc;
  const Schedule sc = work.sc;
  // the normals pass of the geometry iteration on its own: one launch over the whole keyframe list (first and last chunk)
  hipLaunchKernelGGL((geometry_chunk_kernel<1, 0>), dim3(8u * sc.slots_per_xcd), dim3(256), 0, stream, c, (const KfDev*)ctx->kf_table.ptr, 0, keyframe_count, 1, 1, sc, 0u,
                     surfel_rows_rw(work, surfels, active_surfels, surfels_size), (float*)nullptr, 0u);
  BSLAM_HIP_TRY(hipGetLastError());
  return BSLAM_OK;
}

// Launches of (nearly) equal length instead of full chunks and a short tail: ceil(K / chunk) launches of ceil(K / launches)
// keyframes.  0 stays 0 (one launch for the whole list).
static int equal_keyframe_chunks(int keyframe_count, int chunk) {
  if (chunk <= 0 || keyframe_count <= chunk) return chunk;
  const int launches = (keyframe_count + chunk - 1) / chunk;
  return (keyframe_count + launches - 1) / launches;
}

int bslam_optimize_geometry_iteration(
    bslam_context* ctx, void* stream_, int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    int keyframe_count, const bslam_keyframe_view* keyframes, uint32_t surfels_size, const bslam_buffer2d* surfels,
    const bslam_buffer2d* active_surfels) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!use_depth_residuals && !use_descriptor_residuals) return fail(BSLAM_ERR_INVALID_ARGUMENT, "need depth and/or descriptor residuals");   // BS/kernel_opt_geometry.cc:91
  if (surfels_size == 0) return BSLAM_OK;   // BS/kernel_opt_geometry.cc:93-95
  if (!color_camera) return fail(BSLAM_ERR_INVALID_ARGUMENT, "color_camera is null");
  int rc = geometry_common(ctx, stream, color_camera, depth_camera, depth_params, use_descriptor_residuals != 0, keyframe_count, keyframes, surfels_size, surfels, active_surfels);
  if (rc) return rc;
  const CamConsts c = make_cam_consts(ctx, color_camera, depth_camera, depth_params);
#ifndef BSLAM_GEOM_R
#define BSLAM_GEOM_R 3
#endif
#ifndef BSLAM_GEOM_R_DESC
#define BSLAM_GEOM_R_DESC 2
#endif
  SurfelWork work;
  if ((rc = prepare_surfels(ctx, stream, surfels, surfels_size, use_descriptor_residuals ? BSLAM_GEOM_R_DESC : BSLAM_GEOM_R, keyframe_count, &work, use_descriptor_residuals != 0))) return rc;
  const Schedule sc = work.sc;
  const dim3 grid(8u * sc.slots_per_xcd), block(256);
  const KfDev* kfs = (const KfDev*)ctx->kf_table.ptr;
  const SurfelRowsRW rows = surfel_rows_rw(work, surfels, active_surfels, surfels_size);
  {
  ProfScope prof(ctx, stream, 1);
  if (!use_descriptor_residuals) {
    // One launch over all surfels and keyframes.  (Round 1 split the work into resident grids and keyframe chunks of 128 to keep
    // the workgroups in lockstep on the keyframe table; with the per-surfel work order that costs 24 % at K = 200 and nothing is
    // gained by chunks, K = 1000 included.  bslam_set_geometry_keyframe_chunk still selects chunked launches, per-surfel sums
    // carried in library scratch, bit-identical results.)
    const int kf_chunk = equal_keyframe_chunks(keyframe_count, ctx->geom_kf_chunk < 0 ? 0 : ctx->geom_kf_chunk);
    if (kf_chunk <= 0 || keyframe_count <= kf_chunk) {
      hipLaunchKernelGGL((geometry_position_kernel<BSLAM_GEOM_R>), grid, block, 0, stream, c, kfs, keyframe_count, sc, 0u, rows);
    } else {
      const uint32_t acc_pitch = (surfels_size + 63u) & ~63u;
      if ((rc = ctx->exchange.reserve((size_t)acc_pitch * 4 * sizeof(float)))) return rc;
      float* acc = (float*)ctx->exchange.ptr;
      for (int pass = 0; pass < 2; ++pass) {
        for (int k0 = 0; k0 < keyframe_count; k0 += kf_chunk) {
          const int k1 = std::min(keyframe_count, k0 + kf_chunk);
          if (pass == 0) hipLaunchKernelGGL((geometry_chunk_kernel<BSLAM_GEOM_R, 0>), grid, block, 0, stream, c, kfs, k0, k1, k0 == 0, k1 == keyframe_count, sc, 0u, rows, acc, acc_pitch);
          else hipLaunchKernelGGL((geometry_chunk_kernel<BSLAM_GEOM_R, 1>), grid, block, 0, stream, c, kfs, k0, k1, k0 == 0, k1 == keyframe_count, sc, 0u, rows, acc, acc_pitch);
        }
      }
    }
  }
  else {
    // photometric iteration: BSLAM_GEOM_R_DESC surfels per thread, one launch per pass over all surfels and keyframes (K = 300:
    // 22.0 ms per iteration with resident-grid launches, 16.5 ms with one); optional keyframe chunks with the per-surfel sums
    // carried in scratch (4 floats for the normals pass, 8 for the joint position + descriptor pass)
    const int kf_chunk_set = equal_keyframe_chunks(keyframe_count, ctx->geom_kf_chunk < 0 ? 0 : ctx->geom_kf_chunk);
    const int kf_chunk = kf_chunk_set > 0 ? kf_chunk_set : keyframe_count;
    float* acc = nullptr;
    uint32_t acc_pitch = 0;
    if (keyframe_count > kf_chunk) {
      acc_pitch = (surfels_size + 63u) & ~63u;
      if ((rc = ctx->exchange.reserve((size_t)acc_pitch * 8 * sizeof(float)))) return rc;
      acc = (float*)ctx->exchange.ptr;
    }
    for (int pass = 0; pass < 2; ++pass) {
      for (int k0 = 0; k0 < keyframe_count; k0 += kf_chunk) {
        const int k1 = std::min(keyframe_count, k0 + kf_chunk);
        const int fc = k0 == 0, lc = k1 == keyframe_count;
        if (pass == 0) hipLaunchKernelGGL((geometry_desc_chunk_kernel<BSLAM_GEOM_R_DESC, 0, true>), grid, block, 0, stream, c, kfs, k0, k1, fc, lc, sc, 0u, rows, acc, acc_pitch);
        else if (use_depth_residuals) hipLaunchKernelGGL((geometry_desc_chunk_kernel<BSLAM_GEOM_R_DESC, 1, true>), grid, block, 0, stream, c, kfs, k0, k1, fc, lc, sc, 0u, rows, acc, acc_pitch);
        else hipLaunchKernelGGL((geometry_desc_chunk_kernel<BSLAM_GEOM_R_DESC, 1, false>), grid, block, 0, stream, c, kfs, k0, k1, fc, lc, sc, 0u, rows, acc, acc_pitch);
      }
    }
  }
  }
  BSLAM_HIP_TRY(hipGetLastError());
  return BSLAM_OK;
}

int bslam_debug_association(
    bslam_context* ctx, void* stream_, const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    const bslam_keyframe_view* keyframe, uint32_t surfels_size, const bslam_buffer2d* surfels, uint32_t* out_pixel) {
  hipStream_t stream = (hipStream_t)stream_;
  if (surfels_size == 0) return BSLAM_OK;
  int rc = check_common(ctx, depth_camera, depth_params, surfels);
  if (rc) return rc;
  if (!keyframe || !out_pixel) return fail(BSLAM_ERR_INVALID_ARGUMENT, "null argument");
  if (surfels_size > (uint32_t)surfels->width) return fail(BSLAM_ERR_INVALID_ARGUMENT, "surfels_size %u exceeds the buffer width %d", surfels_size, surfels->width);
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));
  std::vector<KfDev> table;
  if ((rc = build_kf_table(depth_camera, nullptr, false, 1, keyframe, &table))) return rc;
  const CamConsts c = make_cam_consts(ctx, nullptr, depth_camera, depth_params);
  if ((rc = upload_kf_table(ctx, stream, table, c))) return rc;
  hipLaunchKernelGGL(association_kernel, dim3((surfels_size + 255) / 256), dim3(256), 0, stream, c, (const KfDev*)ctx->kf_table.ptr,
                     surfel_rows_rw(surfels, nullptr, surfels_size), out_pixel);
  BSLAM_HIP_TRY(hipGetLastError());
  return BSLAM_OK;
}

int bslam_debug_pose_residuals(
    bslam_context* ctx, void* stream_, int use_depth_residuals, int use_descriptor_residuals,
    const bslam_camera4f* color_camera, const bslam_camera4f* depth_camera, const bslam_depth_params* depth_params,
    const bslam_keyframe_view* keyframe, uint32_t surfels_size, const bslam_buffer2d* surfels, float* out) {
  hipStream_t stream = (hipStream_t)stream_;
  if (surfels_size == 0) return BSLAM_OK;
  int rc = check_common(ctx, depth_camera, depth_params, surfels);
  if (rc) return rc;
  if (!keyframe || !out || !color_camera) return fail(BSLAM_ERR_INVALID_ARGUMENT, "null argument");
  if (surfels_size > (uint32_t)surfels->width) return fail(BSLAM_ERR_INVALID_ARGUMENT, "surfels_size %u exceeds the buffer width %d", surfels_size, surfels->width);
  BSLAM_HIP_TRY(hipSetDevice(ctx->device));
  std::vector<KfDev> table;
  if ((rc = build_kf_table(depth_camera, color_camera, use_descriptor_residuals != 0, 1, keyframe, &table))) return rc;
  const CamConsts c = make_cam_consts(ctx, color_camera, depth_camera, depth_params);
  if ((rc = upload_kf_table(ctx, stream, table, c))) return rc;
  const dim3 grid((surfels_size + 255) / 256), block(256);
  const KfDev* kfs = (const KfDev*)ctx->kf_table.ptr;
  const SurfelRowsRW rows = surfel_rows_rw(surfels, nullptr, surfels_size);
  if (use_depth_residuals && use_descriptor_residuals) hipLaunchKernelGGL((residual_probe_kernel<true, true>), grid, block, 0, stream, c, kfs, rows, out);
  else if (use_depth_residuals) hipLaunchKernelGGL((residual_probe_kernel<true, false>), grid, block, 0, stream, c, kfs, rows, out);
  else hipLaunchKernelGGL((residual_probe_kernel<false, true>), grid, block, 0, stream, c, kfs, rows, out);
  BSLAM_HIP_TRY(hipGetLastError());
  return BSLAM_OK;
}

}  // extern "C"

#include "intrinsics_abi.inc"
#include "pcg_abi.inc"
#include "lifecycle_abi.inc"
#include "preprocess_abi.inc"
#include "odometry_abi.inc"
