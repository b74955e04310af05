// Diagnostic microbenchmark (not part of the product), round 5: the convolution contractions of the front end on
// split-bf16 (v_mfma_f32_16x16x32_bf16, six cross products) against the shipped v_mfma_f32_16x16x4_f32, INCLUDING what a real kernel
// pays around the matrix instructions: the B operand comes from an LDS-staged, pooled NLC tile (conv2: K = 5 taps x 16 channels = 80),
// and the split form has to cut every staged element into three bf16 pieces first.
//   fp32:   per 16 positions x 32 outputs: 20 k-steps (K = 4 each) x 2 output blocks = 40 MFMAs of 32 cycles; 20 ds_read_b32
//   split:  K padded to 96 = 3 blocks (taps 0-1, 2-3, 4 + zero tap): 3 x 6 x 2 = 36 MFMAs of 16 cycles; 9 ds_read_b128; + the split of
//           the staged rows (amortised over the 32 outputs and 5 taps that reuse an element)
// One workgroup = 4 waves x 2 position blocks = 128 output positions per item (pool1_conv2_fwd's chunk), rows 2 t0 - 2 .. 2 t0 + 256.
// Reports cycles per item (s_memtime, average over the workgroups of a full-chip launch) for: MFMA + operand reads alone, and with the
// staging (global -> LDS, for split: + split3) in the loop; and the error of both against fp64 on BatchNorm-like data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define CHUNK 128
#define ROWS (2 * CHUNK + 3)          // p1 rows an item needs
#define PS 20                         // fp32 row stride (floats) of a 16-channel row (the product's C2_PS)
#define PB 24                         // bf16 plane row stride (elements): 48 B = 12 dwords

__device__ __forceinline__ __bf16 top16(float x, float& rem) {
  const unsigned u = __float_as_uint(x) & 0xFFFF0000u;
  rem = x - __uint_as_float(u);
  unsigned short h = (unsigned short)(u >> 16);
  __bf16 r; __builtin_memcpy(&r, &h, 2); return r;
}
__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) { float r1, r2, r3; a = top16(x, r1); b = top16(r1, r2); c = top16(r2, r3); }

// p1: [items][ROWS][16] fp32; w2: [32][16][5]; y: [items][CHUNK][32]
template <bool SPLIT, bool STAGE>
__global__ __launch_bounds__(256, 2) void conv2_item_kernel(const float* __restrict__ p1, const float* __restrict__ w2, float* __restrict__ y,
                                                            unsigned long long* stamps, int iters) {
  __shared__ __attribute__((aligned(16))) float ps[ROWS * PS];
  __shared__ __attribute__((aligned(16))) __bf16 pb[3][ROWS + 1][PB];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  // A operands. fp32: A[ob][m] = w2[o = ob*16 + li][c = lq*4 + (m & 3)][kk = m >> 2]  (k = kk*16 + c, k-step m covers 4 channels of one tap)
  float A[2][20];
  bf16x8 As[2][3][3];                 // split: [ob][k block][piece], k = 32 blk + 8 lq + j -> tap 2 blk + (lq >> 1), channel 8 (lq & 1) + j
  for (int ob = 0; ob < 2; ++ob) {
    for (int m = 0; m < 20; ++m) A[ob][m] = w2[((ob * 16 + li) * 16 + lq * 4 + (m & 3)) * 5 + (m >> 2)];
    for (int blk = 0; blk < 3; ++blk)
      for (int j = 0; j < 8; ++j) {
        const int kk = 2 * blk + (lq >> 1), c = 8 * (lq & 1) + j;
        __bf16 p0, p1_, p2; split3(kk < 5 ? w2[((ob * 16 + li) * 16 + c) * 5 + kk] : 0.f, p0, p1_, p2);
        As[ob][blk][0][j] = p0; As[ob][blk][1][j] = p1_; As[ob][blk][2][j] = p2;
      }
  }
  for (int i = tid; i < 3 * (ROWS + 1) * PB; i += 256) (&pb[0][0][0])[i] = (__bf16)0.f;
  const float* src = p1 + (size_t)blockIdx.x * ROWS * 16;
  auto stage = [&]() {
    for (int i = tid; i < ROWS * 4; i += 256) {
      const int row = i >> 2, c4 = i & 3;
      const float4 q = *(const float4*)(src + (size_t)row * 16 + c4 * 4);
      if (!SPLIT) *(float4*)&ps[row * PS + c4 * 4] = q;
      else {
        const float v[4] = {q.x, q.y, q.z, q.w};
        bf16x4 o[3];
        for (int e = 0; e < 4; ++e) { __bf16 a, b, c; split3(v[e], a, b, c); o[0][e] = a; o[1][e] = b; o[2][e] = c; }
        for (int p = 0; p < 3; ++p) *(bf16x4*)&pb[p][row][c4 * 4] = o[p];
      }
    }
  };
  stage();
  __syncthreads();
  f32x4 acc[2][2];
  for (int pbi = 0; pbi < 2; ++pbi) for (int ob = 0; ob < 2; ++ob) acc[pbi][ob] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (STAGE) { __syncthreads(); stage(); __syncthreads(); }
#pragma unroll
    for (int pbi = 0; pbi < 2; ++pbi) {
      const int pl = (w * 2 + pbi) * 16 + li;           // output position within the chunk; taps read rows 2 pl + kk
      if (!SPLIT) {
#pragma unroll
        for (int m = 0; m < 20; ++m) {
          const float bv = ps[(2 * pl + (m >> 2)) * PS + lq * 4 + (m & 3)];      // NOTE: B[k = 4m' + lq] layout of the product differs in detail; same reads per MFMA
          acc[pbi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0][m], bv, acc[pbi][0], 0, 0, 0);
          acc[pbi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[1][m], bv, acc[pbi][1], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int blk = 0; blk < 3; ++blk) {
          const int row = 2 * pl + 2 * blk + (lq >> 1);                           // the zero tap (kk = 5) reads a real row against zero weights
          bf16x8 q[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) q[p] = *(const bf16x8*)&pb[p][row][8 * (lq & 1)];
#pragma unroll
          for (int ob = 0; ob < 2; ++ob) {
            f32x4 a = acc[pbi][ob];
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(As[ob][blk][2], q[0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(As[ob][blk][0], q[2], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(As[ob][blk][1], q[1], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(As[ob][blk][1], q[0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(As[ob][blk][0], q[1], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(As[ob][blk][0], q[0], a, 0, 0, 0);
            acc[pbi][ob] = a;
          }
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  // D layout: lane (li = output o within the block ... A rows are outputs, B columns are positions): rows 4 lq + e = outputs, col li = position
  for (int pbi = 0; pbi < 2; ++pbi)
    for (int ob = 0; ob < 2; ++ob)
      for (int e = 0; e < 4; ++e)
        y[((size_t)blockIdx.x * CHUNK + (w * 2 + pbi) * 16 + li) * 32 + ob * 16 + lq * 4 + e] = acc[pbi][ob][e] / (float)iters;
  if (tid == 0) stamps[blockIdx.x] = t1 - t0;
}

int main() {
  const int items = 512, iters = 200;
  const size_t np1 = (size_t)items * ROWS * 16;
  float* hp = (float*)malloc(np1 * 4); float hw[32 * 16 * 5];
  srand(3);
  for (size_t i = 0; i < np1; ++i) { double u = rand() / (double)RAND_MAX + rand() / (double)RAND_MAX + rand() / (double)RAND_MAX - 1.5; hp[i] = (float)fmax(0.0, u * 2.5 + 0.3); }   // pooled BN + ReLU
  for (int i = 0; i < 32 * 16 * 5; ++i) hw[i] = (float)((rand() / (double)RAND_MAX * 2 - 1) * 0.1118);                                                                             // U(+-1/sqrt(80))
  float *dp, *dw, *dy; unsigned long long* ds;
  (void)hipMalloc(&dp, np1 * 4); (void)hipMalloc(&dw, sizeof(hw)); (void)hipMalloc(&dy, (size_t)items * CHUNK * 32 * 4); (void)hipMalloc(&ds, items * 8);
  (void)hipMemcpy(dp, hp, np1 * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice);
  float* hy = (float*)malloc((size_t)items * CHUNK * 32 * 4);
  unsigned long long hs[512];
  double err[2] = {0, 0}, rms[2] = {0, 0}, scale = 0;
  for (int mode = 0; mode < 4; ++mode) {
    const bool split = mode & 1, stage = mode & 2;
    for (int rep = 0; rep < 2; ++rep) {
      if (!split && !stage) conv2_item_kernel<false, false><<<items, 256>>>(dp, dw, dy, ds, iters);
      if (split && !stage) conv2_item_kernel<true, false><<<items, 256>>>(dp, dw, dy, ds, iters);
      if (!split && stage) conv2_item_kernel<false, true><<<items, 256>>>(dp, dw, dy, ds, iters);
      if (split && stage) conv2_item_kernel<true, true><<<items, 256>>>(dp, dw, dy, ds, iters);
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(hs, ds, sizeof(hs), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < items; ++i) c += (double)hs[i]; c /= items;
    printf("%-10s %-22s: %8.1f cycles per item of 128 positions x 32 outputs x K = 80 (2 workgroups per CU)\n", split ? "split-bf16" : "fp32 MFMA", stage ? "with staging (+ split)" : "MFMA + operand reads", c / iters);
    if (!stage) {
      (void)hipMemcpy(hy, dy, (size_t)items * CHUNK * 32 * 4, hipMemcpyDeviceToHost);
      for (int it = 0; it < 8; ++it)
        for (int pos = 0; pos < CHUNK; ++pos)
          for (int o = 0; o < 32; ++o) {
            double ref = 0;
            for (int kk = 0; kk < 5; ++kk) for (int c2 = 0; c2 < 16; ++c2) ref += (double)hw[(o * 16 + c2) * 5 + kk] * (double)hp[((size_t)it * ROWS + 2 * pos + kk) * 16 + c2];
            const double e = fabs((double)hy[((size_t)it * CHUNK + pos) * 32 + o] - ref);
            if (e > err[split]) err[split] = e;
            rms[split] += e * e; if (!split) scale += ref * ref;
          }
    }
  }
  const double n = 8.0 * CHUNK * 32;
  printf("error against fp64 (K = 80, reference rms %.3f): fp32 MFMA chain max %.3e rms %.3e | split-bf16 max %.3e rms %.3e\n", sqrt(scale / n), err[0], sqrt(rms[0] / n), err[1], sqrt(rms[1] / n));
  return 0;
}
