// Diagnostic microbenchmark (not part of the product): which clock does gfx950 hold in an fp32
// MFMA loop, and what does feeding operands from LDS cost at one wave per SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: operands in registers; 1: B operand ds_read_b32 per MFMA; 2: + A operand too
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* stamps, int iters) {
  __shared__ float lds[16 * 272 * 2];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  for (int i = tid; i < 16 * 272 * 2; i += 256) lds[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * 1e-5f - 0.3f;
  __syncthreads();
  f32x4 acc[36];
  for (int j = 0; j < 36; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a[12], b[12];
  for (int j = 0; j < 12; ++j) { a[j] = lds[tid + j * 7]; b[j] = lds[tid * 3 + j]; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < (MODE == 3 ? 0 : iters); ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = 4 * m + lq;
      float aa[3], bb[12];
#pragma unroll
      for (int g = 0; g < 3; ++g) aa[g] = (MODE >= 2) ? lds[row * 272 + g * 64 + li + (it & 1) * 16 * 272] : a[g + 3 * m];
#pragma unroll
      for (int kb = 0; kb < 12; ++kb) bb[kb] = (MODE >= 1) ? lds[row * 272 + kb * 16 + li + (it & 1) * 16 * 272 + 1] : b[kb];
#pragma unroll
      for (int kb = 0; kb < 12; ++kb)
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g * 12 + kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[g], bb[kb], acc[g * 12 + kb], 0, 0, 0);
    }
  }
  if (MODE == 3) {   // explicit double-buffering: group m+1's operands are requested before group m's MFMAs
    for (int it = 0; it < iters; ++it) {
      float aa[2][3], bb[2][12];
      const int base = (it & 1) * 16 * 272;
#pragma unroll
      for (int g = 0; g < 3; ++g) aa[0][g] = lds[lq * 272 + g * 64 + li + base];
#pragma unroll
      for (int kb = 0; kb < 12; ++kb) bb[0][kb] = lds[lq * 272 + kb * 16 + li + base + 1];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int cur = m & 1, nxt = cur ^ 1;
        if (m < 3) {
          const int row = 4 * (m + 1) + lq;
#pragma unroll
          for (int g = 0; g < 3; ++g) aa[nxt][g] = lds[row * 272 + g * 64 + li + base];
#pragma unroll
          for (int kb = 0; kb < 12; ++kb) bb[nxt][kb] = lds[row * 272 + kb * 16 + li + base + 1];
        }
#pragma unroll
        for (int kb = 0; kb < 12; ++kb)
#pragma unroll
          for (int g = 0; g < 3; ++g) acc[g * 12 + kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[cur][g], bb[cur][kb], acc[g * 12 + kb], 0, 0, 0);
        // interleave: 2 MFMAs, then one prefetch read in their shadow, ... (one wave per SIMD: a burst of
        // reads with no MFMA in flight is fully exposed, a read next to its use waits out the LDS latency)
#pragma unroll
        for (int i = 0; i < 15; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int j = 0; j < 36; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE> void run(const char* name, int iters) {
  float* out; unsigned long long* st;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&st, 256 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<MODE><<<256, 256>>>(out, st, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512]; hipMemcpy(h, st, 256 * 16, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0; for (int i = 0; i < 256; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    cyc /= 256; rt /= 256;
    const double mfma = (double)iters * 144;                       // per wave
    const double tf = 256.0 * 4 * mfma * 2048.0 / (ms * 1e-3) / 1e12;  // 16x16x4 = 1024 MAC = 2048 FLOP
    printf("%-28s %8.3f ms  clock %.3f GHz  cycles/MFMA %.2f  %.1f TFLOP/s\n", name, ms, cyc / rt * 0.1, cyc / mfma, tf);
  }
}
int main() {
  run<0>("regs only", 4000);
  run<1>("B from LDS (b32 per MFMA/3)", 4000);
  run<2>("A and B from LDS", 4000);
  run<3>("A and B from LDS, double-buffered", 4000);
  return 0;
}
