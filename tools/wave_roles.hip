// Diagnostic microbenchmark (not part of the product): do the vector instructions of ONE wave run beside the bf16 MFMAs of
// ANOTHER wave of the same SIMD?  512-thread workgroups = two waves per SIMD; waves 0-3 run a pure v_mfma_f32_16x16x32_bf16
// stream, waves 4-7 a pure VALU (or transcendental, or ds_read) stream.  Timed: each role alone, then both together.
// If the two pipes are independent, together ~ max(alone); if they share issue, together ~ sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode bit 0: waves 0-3 run MFMAs; bit 1: waves 4-7 run their stream.  KIND 0: v_fma_f32, 1: v_rcp_f32 (transcendental), 2: ds_read_b128
template <int KIND>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* stamps, int iters, int mode) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 8];
  const int tid = threadIdx.x, w = tid >> 6;
  for (int i = tid; i < 64 * 4 * 8; i += 512) lds[i] = (float)i;
  __syncthreads();
  float s = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (w < 4) {
    if (mode & 1) {
      f32x4 acc[4];
      for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bf16x8 a[4], b[4];
      for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 8; ++e) { a[j][e] = (__bf16)((float)((tid + j + e) % 13) * 0.125f); b[j][e] = (__bf16)((float)((tid * 3 + j + e) % 11) * 0.25f); }
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) acc[i % 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i % 4], b[(i * 3) % 4], acc[i % 4], 0, 0, 0);
      }
      for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    }
  } else if (mode & 2) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 1.0f + (float)(tid + j) * 1e-3f;
    float4 dq[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 256; ++i) {
        if constexpr (KIND == 0) v[i % 8] = fmaf(v[i % 8], 0.999f, 1e-3f);
        if constexpr (KIND == 1) v[i % 8] = __builtin_amdgcn_rcpf(v[i % 8]);
        if constexpr (KIND == 2) { if ((i & 3) == 0) dq[(i >> 2) % 4] = *(const float4*)&lds[((tid & 63) * 4 + ((i >> 2) % 8) * 256)]; }
      }
      if constexpr (KIND == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); for (int j = 0; j < 4; ++j) v[j] += dq[j].x; }
    }
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + tid] = s;
  if ((tid & 63) == 0) stamps[blockIdx.x * 8 + w] = t1 - t0;
}

template <int KIND> void run(const char* name, int iters) {
  float* out; unsigned long long* st;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&st, 256 * 8 * 8);
  double ms[4] = {0, 0, 0, 0}, cyc_m[4] = {0, 0, 0, 0}, cyc_v[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    k<KIND><<<256, 512>>>(out, st, iters, mode);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<KIND><<<256, 512>>>(out, st, iters, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float t = 0.f; (void)hipEventElapsedTime(&t, e0, e1); ms[mode] = t;
    unsigned long long h[256 * 8]; (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? cyc_m[mode] : cyc_v[mode]) += (double)h[b * 8 + w] / (256 * 4);
  }
  const double nm = (double)iters * 64, nv = (double)iters * (KIND == 2 ? 64 : 256);
  printf("%-14s MFMA wave alone: %6.2f cyc/MFMA (%.3f ms) | %s wave alone: %6.2f cyc/inst (%.3f ms) | together: MFMA wave %6.2f cyc/MFMA, other wave %6.2f cyc/inst (%.3f ms)"
         "  -> together/max(alone) = %.2f, together/sum(alone) = %.2f\n",
         name, cyc_m[1] / nm, ms[1], name, cyc_v[2] / nv, ms[2], cyc_m[3] / nm, cyc_v[3] / nv, ms[3],
         ms[3] / (ms[1] > ms[2] ? ms[1] : ms[2]), ms[3] / (ms[1] + ms[2]));
  (void)hipFree(out); (void)hipFree(st);
}
int main() {
  run<0>("v_fma_f32", 3000);
  run<1>("v_rcp_f32", 3000);
  run<2>("ds_read_b128", 3000);
  return 0;
}
