// Diagnostic microbenchmark (not part of the product): what does one global memory instruction cost a LONE wave
// (one wave per SIMD, a handful of workgroups so bandwidth is irrelevant) in issue time — vaddr vs saddr addressing,
// stores vs loads — and how much of it hides behind MFMAs?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: 5 stores, per-lane 64-bit pointers; 1: 5 stores, scalar base + lane offset; 2: 5 loads (per-lane ptr); 3: none (loop only)
__global__ __launch_bounds__(256, 1) void k(float4* buf, unsigned long long* stamps, int iters, int with_mfma) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float4* p = buf + ((size_t)blockIdx.x * 4 + w) * (size_t)iters * 5 * 64 + lane;          // per-lane running pointer
  const size_t wave_base = ((size_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(w)) * (size_t)iters * 5 * 64;
  float4 v = make_float4(tid, 1.f, 2.f, 3.f), acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (with_mfma) {
#pragma unroll
      for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, v.y, acc, 0, 0, 0);
    }
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 5; ++j) p[j * 64] = v;
      p += 5 * 64;
    } else if (MODE == 1) {
      float4* q = buf + wave_base + (size_t)it * 5 * 64;       // uniform
#pragma unroll
      for (int j = 0; j < 5; ++j) q[j * 64 + lane] = v;
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 5; ++j) { const float4 t = p[j * 64]; acc4.x += t.x; }
      p += 5 * 64;
    }
    v.x += 1.0f;
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
  if (acc[0] + acc4.x == 12345.678f) buf[0] = v;
}

int main() {
  const int iters = 2000, nwg = 8;
  float4* buf; unsigned long long* st;
  (void)hipMalloc(&buf, sizeof(float4) * (size_t)nwg * 4 * iters * 5 * 64);
  (void)hipMemset(buf, 0, sizeof(float4) * (size_t)nwg * 4 * iters * 5 * 64);
  (void)hipMalloc(&st, 8 * nwg);
  unsigned long long h[8];
  const char* names[4] = {"5 stores, per-lane 64-bit pointers", "5 stores, scalar base + lane offset", "5 loads, per-lane pointers", "no memory instructions"};
  for (int mf = 0; mf < 2; ++mf)
    for (int mode = 0; mode < 4; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) k<0><<<nwg, 256>>>(buf, st, iters, mf); else if (mode == 1) k<1><<<nwg, 256>>>(buf, st, iters, mf);
        else if (mode == 2) k<2><<<nwg, 256>>>(buf, st, iters, mf); else k<3><<<nwg, 256>>>(buf, st, iters, mf);
      }
      (void)hipDeviceSynchronize(); (void)hipMemcpy(h, st, 8 * nwg, hipMemcpyDeviceToHost);
      double c = 0; for (int i = 0; i < nwg; ++i) c += (double)h[i]; c /= nwg;
      printf("%-40s %s: %.1f cycles per iteration\n", names[mode], mf ? "+ 16 fp32 MFMAs" : "alone          ", c / iters);
    }
  return 0;
}
