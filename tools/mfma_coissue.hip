// Diagnostic microbenchmark (not part of the product): what does a wave pay for dependent fp32 MFMAs
// (CH accumulators in rotation) and for VALU / transcendental work placed between MFMAs, at one wave
// per SIMD (the situation of the latency-form GRU recurrences)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH, int NV, int NT, int NTHR = 256>
__global__ __launch_bounds__(NTHR, 1) void k(float* out, unsigned long long* stamps, int iters) {
  const int tid = threadIdx.x;
  f32x4 acc[CH];
  for (int j = 0; j < CH; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a[12], b[12], v[8], t[8];
  for (int j = 0; j < 12; ++j) { a[j] = (float)(tid + j) * 1e-3f; b[j] = (float)(tid * 3 + j) * 1e-3f; }
  for (int j = 0; j < 8; ++j) { v[j] = (float)(tid + j) * 1e-4f; t[j] = 1.0f + (float)(tid + j) * 1e-3f; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 48; ++i) {
      acc[i % CH] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i % 12], b[(i * 5) % 12], acc[i % CH], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) v[(i * NV + q) % 8] = fmaf(v[(i * NV + q) % 8], 0.999f, 1e-3f);
#pragma unroll
      for (int q = 0; q < NT; ++q) t[(i * NT + q) % 8] = __builtin_amdgcn_rcpf(t[(i * NT + q) % 8]);
    }
    if (NV + NT > 0) {
#pragma unroll
      for (int i = 0; i < 48; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV + NT, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int j = 0; j < CH; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  for (int j = 0; j < 8; ++j) s += v[j] + t[j];
  out[blockIdx.x * NTHR + tid] = s;
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int CH, int NV, int NT, int NTHR = 256> void run(int iters) {
  float* out; unsigned long long* st;
  hipMalloc(&out, 256 * NTHR * 4); hipMalloc(&st, 256 * 16);
  k<CH, NV, NT, NTHR><<<256, NTHR>>>(out, st, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<CH, NV, NT, NTHR><<<256, NTHR>>>(out, st, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  const double tflops = 256.0 * (NTHR / 64) * (double)iters * 48 * 2048.0 / (ms * 1e-3) / 1e12;
  unsigned long long h[512]; hipMemcpy(h, st, 256 * 16, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0; for (int i = 0; i < 256; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
  printf("waves/SIMD %d  chains %d  valu/MFMA %d  trans/MFMA %d : %.2f cycles per MFMA per wave  (clock %.2f GHz)  wall %.3f ms = %.1f TFLOP/s\n", NTHR / 256, CH, NV, NT, cyc / 256 / ((double)iters * 48),
         cyc / rt * 0.1, ms, tflops);
  hipFree(out); hipFree(st);
}
int main() {
  run<1, 0, 0>(2000); run<2, 0, 0>(2000); run<3, 0, 0>(2000); run<4, 0, 0>(2000); run<6, 0, 0>(2000);
  run<4, 1, 0>(2000); run<4, 2, 0>(2000); run<4, 4, 0>(2000); run<4, 6, 0>(2000); run<4, 8, 0>(2000);
  run<4, 0, 1>(2000); run<4, 0, 2>(2000); run<4, 2, 1>(2000); run<4, 3, 2>(2000);
  run<2, 2, 0>(2000); run<2, 0, 1>(2000);
  // two waves per SIMD: does one wave's VALU work hide under the other's MFMAs?
  run<4, 0, 0, 512>(2000); run<4, 2, 0, 512>(2000); run<4, 4, 0, 512>(2000); run<4, 8, 0, 512>(2000); run<4, 0, 2, 512>(2000);
  run<4, 0, 0, 1024>(2000); run<4, 4, 0, 1024>(2000); run<4, 8, 0, 1024>(2000);
  return 0;
}
