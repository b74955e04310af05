// What does a v_mfma_f32_16x16x32_bf16 cost when TWO waves of a SIMD issue MFMA streams (the layout of gru_fwd_ws / gru_bwd_b5 / b6)?
// One workgroup of 256 or 512 threads on one CU; every wave runs REP x 36 MFMAs in two dependent chains of six (the recurrence's
// pattern).  Variants: A operand from arch VGPRs or from AccVGPRs ("a" pinned, as the resident weights are); 16x16x32 or 32x32x16.
// Prints cycles per MFMA per wave (s_memtime) — at 16 (32) cycles per MFMA and two waves per SIMD the expected value is 32 (64).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>   // 0: A in VGPR, 16x16x32; 1: A in AGPR, 16x16x32; 2: A in AGPR, 32x32x16; 3: A in VGPR, 32x32x16
__global__ __launch_bounds__(512, 1) void k(unsigned long long* out, float* sink, int rep) {
  bf16x8 a[6], b[6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(0.001f * (threadIdx.x + i + j)); b[i][j] = (__bf16)(0.002f * (i - j)); }
  if (MODE == 1 || MODE == 2) for (int i = 0; i < 6; ++i) asm volatile("" : "+a"(a[i]));
  else for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(a[i]));
  for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(b[i]));
  f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  f32x16 d0 = {0}, d1 = {0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rep; ++r) {
#pragma unroll
    for (int kb = 0; kb < 6; ++kb)
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (MODE < 2) {
          if (kb & 1) { c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb], b[t], c1, 0, 0, 0); asm volatile("" : "+a"(c1)); }
          else { c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb], b[t], c0, 0, 0, 0); asm volatile("" : "+a"(c0)); }
        } else {
          if (kb & 1) { d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb], b[t], d1, 0, 0, 0); asm volatile("" : "+a"(d1)); }
          else { d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kb], b[t], d0, 0, 0, 0); asm volatile("" : "+a"(d0)); }
        }
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  sink[threadIdx.x] = c0[0] + c1[1] + d0[2] + d1[3];
}
int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 8 * 8); hipMalloc(&sink, 512 * 4);
  const int rep = 2000;
  const char* names[4] = {"16x16x32, A in arch VGPRs", "16x16x32, A in AccVGPRs", "32x32x16, A in AccVGPRs", "32x32x16, A in arch VGPRs"};
  for (int mode = 0; mode < 4; ++mode)
    for (int threads = 256; threads <= 512; threads += 256) {
      for (int it = 0; it < 2; ++it) {
        if (mode == 0) k<0><<<1, threads>>>(out, sink, rep); else if (mode == 1) k<1><<<1, threads>>>(out, sink, rep);
        else if (mode == 2) k<2><<<1, threads>>>(out, sink, rep); else k<3><<<1, threads>>>(out, sink, rep);
        hipDeviceSynchronize();
      }
      unsigned long long h[8]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      printf("%-30s %d waves per SIMD: %.1f cycles per MFMA and wave (wave 0), %.1f (last wave)\n", names[mode], threads / 256,
             (double)h[0] / (36.0 * rep), (double)h[threads / 64 - 1] / (36.0 * rep));
    }
  return 0;
}
