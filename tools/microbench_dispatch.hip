// microbench_dispatch.hip -- how fast does gfx950 start workgroups that do (almost) nothing?
// A launch of the pose kernel on a short keyframe list starts 12 504 workgroups of which nine in ten run ~150 instructions and
// leave; it takes 110 - 210 us (lead i).  This measures the floor: N workgroups of 256 threads that load one
// word and exit, with no LDS / 12 KB / 30 KB of LDS and few / ~96 VGPRs.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_dispatch.hip -o tools/microbench_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>

template <int kLdsBytes, int kRegs>
__global__ __launch_bounds__(256) void leave_kernel(const int* __restrict__ flag, float* __restrict__ out) {
  __shared__ float lds[kLdsBytes > 0 ? kLdsBytes / 4 : 1];
  if (flag[blockIdx.x & 1023] == 0) return;   // always taken (flag is all zeros): one dependent load, then the workgroup leaves
  float v[kRegs];                             // never executed; keeps the register and LDS allocation of the kernel up
#pragma unroll
  for (int i = 0; i < kRegs; ++i) v[i] = out[threadIdx.x + 256 * i];
  lds[threadIdx.x] = v[0];
  __syncthreads();
  float s = lds[(threadIdx.x * 7) & 255];
#pragma unroll
  for (int i = 0; i < kRegs; ++i) s = s * v[i] + v[(i + 1) % kRegs];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K>
static void run(const char* name, K kernel, const int* flag, float* out, int blocks) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, flag, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  const int reps = 20;
  for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, flag, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("%-34s %8d workgroups  %8.1f us per launch  %6.2f ns per workgroup\n", name, blocks, us, us * 1e3 / blocks);
}

int main() {
  int* flag;
  float* out;
  (void)hipMalloc(&flag, 1024 * sizeof(int));
  (void)hipMemset(flag, 0, 1024 * sizeof(int));
  (void)hipMalloc(&out, (size_t)256 * 256 * 128 * sizeof(float));
  for (int blocks : {12504, 50016, 200064}) {
    run("no LDS, few registers", leave_kernel<0, 4>, flag, out, blocks);
    run("12 KB LDS, few registers", leave_kernel<12288, 4>, flag, out, blocks);
    run("12 KB LDS, ~96 registers", leave_kernel<12288, 80>, flag, out, blocks);
    run("30 KB LDS, ~96 registers", leave_kernel<30720, 80>, flag, out, blocks);
  }
  return 0;
}
