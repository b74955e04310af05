// Diagnostic (not part of the product): checks the operand path of gru_bwd_b4's weight-gradient contraction in isolation —
// dW[u][c] = sum over the 16 batch rows of ONE step of dg[row][u] * x[row][c] on v_mfma_f32_32x32x16_bf16, both operands
// fetched column-major from row-major split-bf16 planes by ds_read_b64_tr_b16 with the kernel's row strides (16 * odd dwords)
// and its quad swizzle (8-element chunks XORed with g(row >> 2), g = [0,3,2,1]).
//   32x32x16 lane map:  A[i = l & 31][k = 8 (l >> 5) + j],  B[k = 8 (l >> 5) + j][n = l & 31],  j = 0..7
//                       D[i = 8 (r >> 2) + 4 (l >> 5) + (r & 3)][n = l & 31],  r = 0..15
//   transposed read h (0, 1) of a 32-column block: lane l = 32 g + 16 half + i supplies the address of
//   row 8 g + 4 h + (i >> 2), columns c0 + 16 half + 4 (i & 3) .. + 3 and receives rows 8 g + 4 h + 0..3 of column c0 + 16 half + i.
// Also checks the row reads (ds_read_b128, B operand of the 16x16x32 recurrence) of the same swizzled planes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define LDSP __attribute__((address_space(3)))

__device__ __forceinline__ bf16x4 tr_read(const __bf16* p) {
  typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 v4;
  v4 r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDSP v4*)p);
  return __builtin_bit_cast(bf16x4, r);
}
__device__ __forceinline__ __bf16 top16(float x, float& rem) {
  const uint32_t u = __float_as_uint(x) & 0xFFFF0000u;
  rem = x - __uint_as_float(u);
  const unsigned short h = (unsigned short)(u >> 16);
  __bf16 r; __builtin_memcpy(&r, &h, 2); return r;
}
__host__ __device__ constexpr int swz(int row) { return ((4 - (row >> 2)) & 3) * 8; }

constexpr int NU = 256, NC = 96;             // dg columns [dr|dz|dhn|dn], [x | h_prev] columns (layer 0)
constexpr int SD = 288, SX = 96;             // plane row strides in bf16 elements: 144 and 48 dwords = 16 * odd

template <int T> __device__ __forceinline__ f32x16 mf32(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc) {
  constexpr int ai[6] = {2, 0, 1, 1, 0, 0}, bi[6] = {0, 2, 1, 0, 1, 0};
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ai[T]], b[bi[T]], acc, 0, 0, 0);
}

// dW [NU][NC] for one step; one wave per (32-unit block, 32-column block) pair, 256 threads = 4 waves looping over pairs
__global__ void k_dw(const float* dg, const float* x, float* dW, float* rowcheck) {
  __shared__ __attribute__((aligned(16))) __bf16 dgp[3][16 * SD];
  __shared__ __attribute__((aligned(16))) __bf16 xp[3][16 * SX];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 16 * NU; i += 256) {
    const int r = i / NU, c = i % NU;
    float r1, r2, r3;
    const int o = r * SD + (c ^ swz(r));
    dgp[0][o] = top16(dg[i], r1); dgp[1][o] = top16(r1, r2); dgp[2][o] = top16(r2, r3);
  }
  for (int i = tid; i < 16 * NC; i += 256) {
    const int r = i / NC, c = i % NC;
    float r1, r2, r3;
    const int o = r * SX + (c ^ swz(r));
    xp[0][o] = top16(x[i], r1); xp[1][o] = top16(r1, r2); xp[2][o] = top16(r2, r3);
  }
  __syncthreads();
  const int g = lane >> 5, half = (lane >> 4) & 1, i16 = lane & 15;
  auto frag = [&](const __bf16* plane0, int stride, int piece_stride, int c0, bf16x8 (&f)[3]) {
    for (int p = 0; p < 3; ++p) {
      bf16x4 v[2];
      for (int h = 0; h < 2; ++h) {
        const int row = 8 * g + 4 * h + (i16 >> 2);
        const int sw = ((4 - (2 * g + h)) & 3) * 8;                 // swz(row): row >> 2 = 2 g + h
        v[h] = tr_read(plane0 + p * piece_stride + row * stride + c0 + ((16 * half + 4 * (i16 & 3)) ^ sw));
      }
      f[p] = (bf16x8){v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3]};
    }
  };
  for (int pair = w; pair < (NU / 32) * (NC / 32); pair += 4) {
    const int ub = pair / (NC / 32), cb = pair % (NC / 32);
    bf16x8 A[3], B[3];
    frag(&dgp[0][0], SD, 16 * SD, ub * 32, A);
    frag(&xp[0][0], SX, 16 * SX, cb * 32, B);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = mf32<0>(A, B, acc); acc = mf32<1>(A, B, acc); acc = mf32<2>(A, B, acc);
    acc = mf32<3>(A, B, acc); acc = mf32<4>(A, B, acc); acc = mf32<5>(A, B, acc);
    for (int r = 0; r < 16; ++r) {
      const int u = ub * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), c = cb * 32 + (lane & 31);
      dW[u * NC + c] = acc[r];
    }
  }
  // row reads: lane (li, lq) reads 8 consecutive columns kb*32 + lq*8 .. of row li of piece 0 (bf16 -> float), all 8 k blocks
  if (w == 0) {
    const int li = lane & 15, lq = lane >> 4;
    for (int kb = 0; kb < NU / 32; ++kb) {
      const bf16x8 q = *(const bf16x8*)&dgp[0][li * SD + kb * 32 + ((lq * 8) ^ swz(li))];
      for (int j = 0; j < 8; ++j) rowcheck[(li * NU) + kb * 32 + lq * 8 + j] = (float)q[j];
    }
  }
}

int main() {
  std::vector<float> dg(16 * NU), x(16 * NC), dW(NU * NC), rc(16 * NU);
  srand(7);
  for (auto& v : dg) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : x) v = ((float)rand() / RAND_MAX - 0.5f) * 3.0f;
  float *d_dg, *d_x, *d_dW, *d_rc;
  (void)hipMalloc(&d_dg, dg.size() * 4); (void)hipMalloc(&d_x, x.size() * 4); (void)hipMalloc(&d_dW, dW.size() * 4); (void)hipMalloc(&d_rc, rc.size() * 4);
  (void)hipMemcpy(d_dg, dg.data(), dg.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  k_dw<<<1, 256>>>(d_dg, d_x, d_dW, d_rc);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  (void)hipMemcpy(dW.data(), d_dW, dW.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(rc.data(), d_rc, rc.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0, maxerr32 = 0;
  for (int u = 0; u < NU; ++u)
    for (int c = 0; c < NC; ++c) {
      double ref = 0; float f32 = 0.f;
      for (int r = 0; r < 16; ++r) { ref += (double)dg[r * NU + u] * (double)x[r * NC + c]; f32 = fmaf(dg[r * NU + u], x[r * NC + c], f32); }
      maxerr = fmax(maxerr, fabs(dW[u * NC + c] - ref)); maxref = fmax(maxref, fabs(ref)); maxerr32 = fmax(maxerr32, fabs((double)f32 - ref));
    }
  int bad_rows = 0;
  for (int i = 0; i < 16 * NU; ++i) {
    const uint32_t u = *(const uint32_t*)&dg[i] & 0xFFFF0000u;
    if (rc[i] != *(const float*)&u) ++bad_rows;
  }
  printf("one-step dW on 32x32x16 split-bf16 via transposed reads of quad-swizzled planes: max |err| %.3e (fp32 fmaf chain: %.3e), max |ref| %.3e -> %s\n",
         maxerr, maxerr32, maxref, maxerr <= 2 * maxerr32 + 1e-7 ? "OK" : "MISMATCH");
  printf("row reads (ds_read_b128) of the swizzled plane: %d mismatches -> %s\n", bad_rows, bad_rows == 0 ? "OK" : "MISMATCH");
  return (maxerr <= 2 * maxerr32 + 1e-7 && bad_rows == 0) ? 0 : 1;
}
