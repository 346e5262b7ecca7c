// context.hpp -- host-side context of libbadslam_hip: scratch slabs, keyframe table upload,
// error reporting.  Plays the role of the reference's PoseEstimationHelperBuffers /
// IntrinsicsOptimizationHelperBuffers (BS/kernels.h:46-89): scratch is allocated once and
// re-used, launch paths never allocate unless a slab has to grow.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "device_math.hpp"

namespace bslam {

extern thread_local std::string g_last_error;

inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define BSLAM_HIP_TRY(expr)                                                                         \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return ::bslam::fail(BSLAM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Byte offsets of the small device scalars in bslam_context::misc (256 bytes, zeroed at creation): [0, 16) active-keyframe
// counters of the batched pose loop, [64, 80) pair census, [128, 192) PCG scalars, [192, 224) lifecycle / preprocessing counters.
constexpr size_t kMiscCullStats = 224;   // unsigned long long[2]: (work slot, keyframe) pairs tested / visited by the pose kernel

// A device slab that only ever grows.
struct Slab {
  void* ptr = nullptr;
  size_t bytes = 0;
  int reserve(size_t need) {
    if (need <= bytes) return BSLAM_OK;
    if (ptr) { hipError_t e = hipFree(ptr); (void)e; ptr = nullptr; bytes = 0; }
    size_t want = need + need / 4 + 256;
    hipError_t e = hipMalloc(&ptr, want);
    if (e != hipSuccess) return fail(BSLAM_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    bytes = want;
    return BSLAM_OK;
  }
  void release() { if (ptr) { hipError_t e = hipFree(ptr); (void)e; } ptr = nullptr; bytes = 0; }
};

struct PinnedSlab {
  void* ptr = nullptr;
  size_t bytes = 0;
  int reserve(size_t need) {
    if (need <= bytes) return BSLAM_OK;
    if (ptr) { hipError_t e = hipHostFree(ptr); (void)e; ptr = nullptr; bytes = 0; }
    size_t want = need + need / 4 + 256;
    hipError_t e = hipHostMalloc(&ptr, want, hipHostMallocDefault);
    if (e != hipSuccess) return fail(BSLAM_ERR_OUT_OF_MEMORY, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    bytes = want;
    return BSLAM_OK;
  }
  void release() { if (ptr) { hipError_t e = hipHostFree(ptr); (void)e; } ptr = nullptr; bytes = 0; }
};

// Ring of pinned upload buffers: an API call takes the next slot for its small host->device uploads and
// records an event after enqueuing the copy, so calls do not have to drain the stream before re-using
// staging memory (the wait only happens if the ring wraps onto a copy that has not finished yet).
struct StagingRing {
  static constexpr int kSlots = 8;
  PinnedSlab slot[kSlots];
  hipEvent_t done[kSlots] = {};
  bool pending[kSlots] = {};
  int next = 0;
  int cur = 0;
  int acquire(size_t bytes, void** out) {
    cur = next;
    next = (next + 1) % kSlots;
    if (pending[cur]) {
      hipError_t e = hipEventSynchronize(done[cur]);
      if (e != hipSuccess) return fail(BSLAM_ERR_HIP, "hipEventSynchronize failed: %s", hipGetErrorString(e));
      pending[cur] = false;
    }
    int rc = slot[cur].reserve(bytes);
    if (rc) return rc;
    *out = slot[cur].ptr;
    return BSLAM_OK;
  }
  int commit(hipStream_t stream) {   // after the copies out of the acquired slot were enqueued
    if (!done[cur]) {
      hipError_t e = hipEventCreateWithFlags(&done[cur], hipEventDisableTiming);
      if (e != hipSuccess) return fail(BSLAM_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
    }
    hipError_t e = hipEventRecord(done[cur], stream);
    if (e != hipSuccess) return fail(BSLAM_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(e));
    pending[cur] = true;
    return BSLAM_OK;
  }
  void release() {
    for (int i = 0; i < kSlots; ++i) {
      if (done[i]) { hipError_t e = hipEventDestroy(done[i]); (void)e; done[i] = nullptr; }
      slot[i].release();
      pending[i] = false;
    }
  }
};

}  // namespace bslam

struct bslam_context {
  int device = 0;
  int tex_mode = BSLAM_TEX_FIXED_POINT_1_8;
  int cu_count = 256;
  bslam::Slab kf_table;      // KfDev[K]
  bslam::Slab records;       // PixelRecord[K][h][w] derived pixel records (16 B)
  bslam::Slab quads;         // uint32[K][ch+1][cw+1] derived luma quads (only when colour images are given)
  bslam::Slab partials;      // float[tiles][K][32]
  bslam::Slab coeffs;        // float[K][32]
  bslam::Slab pose_state;    // PoseState[K]
  bslam::Slab misc;          // small device scalars
  bslam::Slab intr_cells;    // B[5][cells], D, b2, obs of the intrinsics step; or the PCG path's fp64 cell sums
  int geom_kf_chunk = -1;    // geometry iteration: keyframes per launch (0: one launch for the whole list; -1: default = one launch)
  int intr_cells_owner = 0;  // 0: intrinsics step (zeroes per call), 1: PCG (kept zero between calls)
  size_t intr_cells_cells = 0;
  bslam::PinnedSlab staging; // pinned host staging for tiny up/downloads
  bslam::PinnedSlab staging2;
  bslam::StagingRing upload_ring;   // keyframe table / pose state uploads
  hipEvent_t iter_done[4] = {};     // batched pose loop: one event per in-flight iteration (recorded behind its flag copy)
  hipEvent_t solve_done[4] = {};    // ... recorded on the BA stream behind the iteration's solve kernel
  hipStream_t copy_stream = nullptr;   // side stream that carries the 4-byte convergence flags to the host, off the BA stream's critical path
  // what the device copy of the keyframe table holds (upload_kf_table skips an identical upload; the pose kernels rewrite
  // frame_T_global on the device and invalidate it)
  std::vector<uint8_t> kf_table_host;
  const void* kf_table_host_ptr = nullptr;
  // derived-record cache (bslam_set_keyframe_cache)
  bool keyframe_cache = false;
  std::vector<uint64_t> records_signature;   // what the depth records were built from
  std::vector<uint64_t> quads_signature;     // what the luma quads were built from
  // XCD-aware schedule: granule order (short keyframe lists) and per-surfel order, each cached per surfel buffer
  bslam::Slab quads_aux;     // luma quads of the tracked frame's colour pyramid level (odometry)
  bslam::Slab lifecycle;     // supporting-surfel cell images, scan buffers, flags of the surfel lifecycle calls
  bslam::Slab exchange;      // staging of the multi-rank exchanges (PCG shared unknowns, intrinsics sums)
  bslam_allreduce_fn allreduce = nullptr;   // bslam_set_allreduce: sum across the ranks of a surfel-sharded run
  void* allreduce_user = nullptr;
  void* comm = nullptr;                     // bslam_comm_init: RCCL communicator (ncclComm_t) of the surfel-sharded run
  int comm_rank = 0, comm_world = 1;
  bslam::Slab order;
  bslam::Slab centroids;     // granule centroids (scratch of make_schedule)
  bslam::Slab perm;          // per-surfel Morton order (+ sort scratch), cached like `order`
  bslam::Slab sorted_rows;   // the seven persistent surfel rows in that order, rebuilt by every call that uses it
  bslam::Slab bounds;        // float4[2 * granules]: bounding box of every granule of the sorted copy, rebuilt with it
  // what the sorted copy holds (prepare_surfels): PCGStep1 re-uses the copy PCGInit / the previous PCGStep1 of the same solve
  // made -- the surfels do not change inside a solve (BS/direct_ba_pcg.cc:339-425) -- every other call rebuilds it
  const void* sorted_key_ptr = nullptr;
  uint32_t sorted_key_size = 0;
  size_t sorted_key_pitch = 0;
  int sorted_key_rows = 0;
  bool sorted_key_bounds = false;
  uint64_t sorted_key_perm_serial = 0;   // the permutation the copy was made with
  uint64_t perm_serial = 0;              // counts rebuilds of `perm`
  bslam::Slab pose_list;     // batched pose loop: int n | int list[K] (the unconverged keyframes, ascending) | int pos[K] (place in the list, or -1)
  bslam::Slab vis;           // uint64[chunks][slots]: keyframes of a chunk (<= 64) a work slot visited in the last pose_accumulate launch
  bool culling = true;       // block-level frustum culling in the pair kernels (bslam_set_culling)
  int pose_list_min_keyframes = 64;   // batched pose loop: walk the list of unconverged keyframes from this many keyframes on (bslam_set_pose_keyframe_list; 0: never)
  const void* perm_key_ptr = nullptr;
  uint32_t perm_key_size = 0;
  size_t perm_key_pitch = 0;
  hipEvent_t perm_ready = nullptr;   // recorded behind the sort that produced `perm`
  hipStream_t perm_stream = nullptr; // ... on this stream
  bool use_schedule = true;
  const void* order_key_ptr = nullptr;
  uint32_t order_key_size = 0;
  size_t order_key_pitch = 0;
  // kernel timing (bslam_profile_*)
  bool profiling = false;
  struct ProfEntry { hipEvent_t start, stop; int tag; };
  std::vector<ProfEntry> prof_pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
  int prof_launches[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float prof_ms[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bslam::Slab prof_counters; // unsigned long long[blocks][2]: work counters filled by the counting kernel variants while profiling
  size_t prof_counter_slots = 0;
  uint64_t prof_counter_base[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // counts carried over a resize of the per-block array
};

namespace bslam {
// The exchange of a surfel-sharded run: an in-place float sum across ranks, ordered on `stream`.  A caller-supplied hook
// (bslam_set_allreduce) takes precedence over the library's own RCCL communicator (bslam_comm_init).
int rccl_allreduce_sum(bslam_context* ctx, hipStream_t stream, float* device_buffer, size_t count);   // badslam_hip.hip
inline bool has_exchange(const bslam_context* ctx) { return ctx->allreduce != nullptr || ctx->comm != nullptr; }
inline int exchange_sum(bslam_context* ctx, hipStream_t stream, float* device_buffer, size_t count) {
  if (ctx->allreduce) {
    const int arc = ctx->allreduce(ctx->allreduce_user, device_buffer, count, stream);
    if (arc) return fail(BSLAM_ERR_HIP, "allreduce callback failed with %d", arc);
    return BSLAM_OK;
  }
  if (ctx->comm) return rccl_allreduce_sum(ctx, stream, device_buffer, count);
  return BSLAM_OK;
}

// Brackets one kernel launch with events when profiling is on.
struct ProfScope {
  bslam_context* ctx; hipStream_t stream; std::pair<hipEvent_t, hipEvent_t> ev; bool on; int tag;
  ProfScope(bslam_context* c, hipStream_t s, int tag_ = 0) : ctx(c), stream(s), on(c->profiling), tag(tag_) {
    if (!on) return;
    if (!ctx->prof_pool.empty()) { ev = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); }
    else { hipError_t e = hipEventCreate(&ev.first); e = hipEventCreate(&ev.second); (void)e; }
    hipError_t e = hipEventRecord(ev.first, stream); (void)e;
  }
  ~ProfScope() {
    if (!on) return;
    hipError_t e = hipEventRecord(ev.second, stream); (void)e;
    ctx->prof_pending.push_back(bslam_context::ProfEntry{ev.first, ev.second, tag});
  }
};
}  // namespace bslam
