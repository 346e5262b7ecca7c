// microbench_valu.hip -- what does one SIMD of gfx950 sustain for plain (non-packed) fp32 VALU?
// Inline asm keeps hipcc from SLP-packing the streams.  8 independent chains, 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o tools/microbench_valu
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kIters = 4096;

#define CHAIN8(OP)                                                                                         \
  asm volatile(OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n"  \
               OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n"  \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b))
#define CHAIN8_2(OP)                                                                          \
  asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n"     \
               OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"     \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a))
#define CHAIN8_1(OP)                                                                  \
  asm volatile(OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n"             \
               OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n"             \
               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7))

#define KERNEL(NAME, BODY)                                                                             \
  __global__ __launch_bounds__(256) void NAME(float* out, float a, float b) {                          \
    float x0 = threadIdx.x + 1, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    for (int i = 0; i < kIters; ++i) { BODY; }                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;               \
  }

KERNEL(k_fma, CHAIN8("v_fma_f32"))
KERNEL(k_mul, CHAIN8_2("v_mul_f32"))
KERNEL(k_add, CHAIN8_2("v_add_f32"))
KERNEL(k_rcp, CHAIN8_1("v_rcp_f32"))
KERNEL(k_sqrt, CHAIN8_1("v_sqrt_f32"))
KERNEL(k_cvt, CHAIN8_1("v_cvt_i32_f32"))

typedef float float2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pkfma(float* out, float a, float b) {
  float2_t x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  const float2_t va = {a, a}, vb = {b, b};
  for (int i = 0; i < kIters; ++i) {
    asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\n"
                 "v_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va), "v"(vb));
  }
  const float2_t s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <typename K>
static void run(const char* name, K kernel, float* d_out) {
  const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  const double per_simd = (double)blocks * 4 * kIters * 8 / 1024.0;
  printf("%-14s %8.3f ms  -> %.2f cycles per wave64 instruction at 2.4 GHz (%.2f at 2.0 GHz)\n", name, ms, (ms * 1e6 * 2.4) / per_simd,
         (ms * 1e6 * 2.0) / per_simd);
}

int main() {
  float* d_out;
  (void)hipMalloc(&d_out, 256 * 8 * 256 * sizeof(float));
  run("v_fma_f32", k_fma, d_out);
  run("v_mul_f32", k_mul, d_out);
  run("v_add_f32", k_add, d_out);
  run("v_pk_fma_f32", k_pkfma, d_out);
  run("v_rcp_f32", k_rcp, d_out);
  run("v_sqrt_f32", k_sqrt, d_out);
  run("v_cvt_i32_f32", k_cvt, d_out);
  return 0;
}
