// Diagnostic microbenchmark (not part of the product): would split-bf16 MFMA ("bf16x3": x = x1 + x2 + x3 with three
// bf16 pieces, six cross products accumulated in fp32) be an admissible replacement for the fp32 MFMA in the GRU
// contractions?  Measures (a) the error of a 16x16 tile with K = 64 against an fp64 reference for fp32-MFMA, bf16x3
// (6 products), bf16x2 (3 products) and plain bf16, on data shaped like the recurrence's (|h| <= 1, |w| <= 0.125), and
// (b) cycles per K = 64 tile for fp32 MFMA vs the six-product scheme (operands already split, as a producer would
// leave them in LDS).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifdef SPLIT_TRUNC      // pieces by truncation (bit masks, full-rate VALU) instead of round-to-nearest conversions
__device__ __forceinline__ __bf16 top16(float x, float& rem) {
  const unsigned u = __float_as_uint(x) & 0xFFFF0000u;
  rem = x - __uint_as_float(u);
  unsigned short h = (unsigned short)(u >> 16);
  __bf16 r; __builtin_memcpy(&r, &h, 2); return r;
}
__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  float r1, r2, r3;
  a = top16(x, r1); b = top16(r1, r2); c = top16(r2, r3);
}
#else
__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x; const float r1 = x - (float)a;
  b = (__bf16)r1; const float r2 = r1 - (float)b;
  c = (__bf16)r2;
}
#endif

// A: [16][64] row-major, B: [64][16] row-major, out[mode][16][16]
__global__ void accuracy_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out) {
  const int lane = threadIdx.x, li = lane & 15, lq = lane >> 4;
  f32x4 c32 = {0.f, 0.f, 0.f, 0.f}, c6 = c32, c3 = c32, c1 = c32;
  for (int k0 = 0; k0 < 64; k0 += 4) c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[li * 64 + k0 + lq], B[(k0 + lq) * 16 + li], c32, 0, 0, 0);
  for (int kb = 0; kb < 64; kb += 32) {
    bf16x8 a1, a2, a3, b1, b2, b3;
    for (int j = 0; j < 8; ++j) {
      __bf16 p, q, r;
      split3(A[li * 64 + kb + lq * 8 + j], p, q, r); a1[j] = p; a2[j] = q; a3[j] = r;
      split3(B[(kb + lq * 8 + j) * 16 + li], p, q, r); b1[j] = p; b2[j] = q; b3[j] = r;
    }
    // smallest terms first
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, c6, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c6, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c3, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c1, 0, 0, 0);
  }
  for (int e = 0; e < 4; ++e) {
    const int row = lq * 4 + e, col = li;
    out[0 * 256 + row * 16 + col] = c32[e]; out[1 * 256 + row * 16 + col] = c6[e];
    out[2 * 256 + row * 16 + col] = c3[e]; out[3 * 256 + row * 16 + col] = c1[e];
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void speed_kernel(float* out, unsigned long long* stamps, int iters) {
  const int tid = threadIdx.x;
  f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float a[16], b[16];
  bf16x8 ah[2][3], bh[2][3];
  for (int j = 0; j < 16; ++j) { a[j] = (tid + j) * 1e-4f; b[j] = (tid * 3 + j) * 1e-4f; }
  for (int kb = 0; kb < 2; ++kb) for (int p = 0; p < 3; ++p) for (int j = 0; j < 8; ++j) { ah[kb][p][j] = (__bf16)((tid + j + p) * 1e-3f); bh[kb][p][j] = (__bf16)((tid + 2 * j + p) * 1e-3f); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {          // three output tiles per wave-step, K = 64 each (the recurrence's shape)
      if (MODE == 0) {
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[(m + g) & 15], acc[g], 0, 0, 0);
      } else {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kb][2], bh[kb][0], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kb][0], bh[kb][2], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kb][1], bh[kb][1], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kb][1], bh[kb][0], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kb][0], bh[kb][1], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kb][0], bh[kb][0], acc[g], 0, 0, 0);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int g = 0; g < 3; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

int main() {
  float hA[16 * 64], hB[64 * 16], hout[4 * 256];
  double worst[4] = {0, 0, 0, 0}, rms[4] = {0, 0, 0, 0}, scale = 0;
  float *dA, *dB, *dout;
  (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dout, sizeof(hout));
  srand(1);
  const int trials = 200;
  for (int t = 0; t < trials; ++t) {
    for (int i = 0; i < 16 * 64; ++i) hA[i] = (float)((rand() / (double)RAND_MAX) * 2 - 1);              // h in [-1, 1]
    for (int i = 0; i < 64 * 16; ++i) hB[i] = (float)(((rand() / (double)RAND_MAX) * 2 - 1) * 0.125);      // w in [-1/8, 1/8]
    (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    accuracy_kernel<<<1, 64>>>(dA, dB, dout);
    (void)hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
      double ref = 0; for (int k = 0; k < 64; ++k) ref += (double)hA[r * 64 + k] * (double)hB[k * 16 + c];
      scale += ref * ref;
      for (int m = 0; m < 4; ++m) { const double e = fabs((double)hout[m * 256 + r * 16 + c] - ref); if (e > worst[m]) worst[m] = e; rms[m] += e * e; }
    }
  }
  const double n = trials * 256.0, ref_rms = sqrt(scale / n);
  const char* names[4] = {"fp32 MFMA 16x16x4", "bf16x3 (6 products)", "bf16x2 (3 products)", "bf16 (1 product)"};
  printf("K = 64 dot products, reference rms %.3f:\n", ref_rms);
  for (int m = 0; m < 4; ++m) printf("  %-22s max abs error %.3e   rms error %.3e   (rms error / rms value %.2e)\n", names[m], worst[m], sqrt(rms[m] / n), sqrt(rms[m] / n) / ref_rms);
  float* o; unsigned long long* st; (void)hipMalloc(&o, 256 * 256 * 4); (void)hipMalloc(&st, 256 * 8);
  unsigned long long h[256];
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) { if (mode == 0) speed_kernel<0><<<256, 256>>>(o, st, 2000); else speed_kernel<1><<<256, 256>>>(o, st, 2000); }
    (void)hipDeviceSynchronize(); (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += (double)h[i]; c /= 256;
    printf("%s: %.1f cycles per wave-step of three K = 64 tiles (%d MFMAs)\n", mode == 0 ? "fp32 MFMA        " : "bf16x3 six-product", c / 2000, mode == 0 ? 48 : 36);
  }
  return 0;
}
