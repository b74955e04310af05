// Diagnostic (not part of the product): checks the lane map of gfx950's ds_read_b64_tr_b16 and the weight-gradient
// contraction built on it — dW[u][c] = sum over (step in {0,1}, batch row 0..15) dg[step][row][u] * x[step][row][c] —
// exactly as gru_bwd_fused uses it: both operands are row-major [row][col] bf16 planes (three split-bf16 pieces each),
// the contraction index k of v_mfma_f32_16x16x32_bf16 is (step, batch row), and both fragments come from transposed reads.
//   lane l = 16 g + i  (g = l >> 4 = the MFMA's k group, i = l & 15):  step = g >> 1, half = g & 1
//   read h (0, 1): supplies the address of row 8 h + 4 half + (i >> 2), columns c0 + 4 (i & 3) .. +3
//                  receives rows 8 h + 4 half + 0..3 of column c0 + i   ->  fragment elements 4 h + 0..3
// Row strides are 8 * odd dwords so that the eight rows a 32-lane half touches per read fall into disjoint 8-bank windows.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
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
__device__ __forceinline__ f32x4 mfma_bf16x3(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
  return acc;
}

constexpr int NU = 64, NC = 32;             // dg columns (units), x columns
constexpr int SD = 80, SX = 48;             // plane row strides in bf16 elements: 40 and 24 dwords = 8 * odd

// out_map[lane][h][q] = the value lane received (plane filled with row * 64 + col: exact in bf16 up to 255 -> use 16 x 16)
__global__ void k_map(float* out_map) {
  __shared__ __attribute__((aligned(16))) __bf16 pl[16 * SD];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16 * SD; i += 64) { const int r = i / SD, c = i % SD; pl[i] = (__bf16)(c < 16 ? (float)(r * 16 + c) : -1.0f); }
  __syncthreads();
  const int g = tid >> 4, i = tid & 15;
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * h + 4 * (g & 1) + (i >> 2);
    const bf16x4 v = tr_read(&pl[row * SD + 4 * (i & 3)]);
    for (int q = 0; q < 4; ++q) out_map[(tid * 2 + h) * 4 + q] = (float)v[q];
  }
}

__global__ void k_dw(const float* dg, const float* x, float* dW) {      // dg [2][16][NU], x [2][16][NC], dW [NU][NC]
  __shared__ __attribute__((aligned(16))) __bf16 dgp[2][3][16][SD];
  __shared__ __attribute__((aligned(16))) __bf16 xp[2][3][16][SX];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  for (int i = tid; i < 2 * 16 * NU; i += 256) {
    const int s = i / (16 * NU), r = (i / NU) % 16, c = i % NU;
    float r1, r2, r3; dgp[s][0][r][c] = top16(dg[i], r1); dgp[s][1][r][c] = top16(r1, r2); dgp[s][2][r][c] = top16(r2, r3);
  }
  for (int i = tid; i < 2 * 16 * NC; i += 256) {
    const int s = i / (16 * NC), r = (i / NC) % 16, c = i % NC;
    float r1, r2, r3; xp[s][0][r][c] = top16(x[i], r1); xp[s][1][r][c] = top16(r1, r2); xp[s][2][r][c] = top16(r2, r3);
  }
  __syncthreads();
  const int step = lq >> 1, half = lq & 1;
  auto frag = [&](const __bf16* plane0, int stride, int piece_stride, int c0, bf16x8 (&f)[3]) {
    for (int p = 0; p < 3; ++p) {
      const __bf16* base = plane0 + p * piece_stride + (4 * half + (li >> 2)) * stride + c0 + 4 * (li & 3);
      const bf16x4 lo = tr_read(base), hi = tr_read(base + 8 * stride);
      f[p] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };
  bf16x8 A[3], Bx[NC / 16][3];
  frag(&dgp[step][0][0][0], SD, 16 * SD, w * 16, A);                  // wave w: units w*16 ..
  for (int cb = 0; cb < NC / 16; ++cb) frag(&xp[step][0][0][0], SX, 16 * SX, cb * 16, Bx[cb]);
  for (int cb = 0; cb < NC / 16; ++cb) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = mfma_bf16x3(A, Bx[cb], acc);
    for (int e = 0; e < 4; ++e) dW[(w * 16 + 4 * lq + e) * NC + cb * 16 + li] = acc[e];
  }
}

int main() {
  // ---- 1. lane map with exact integers ----
  float* dmap; (void)hipMalloc(&dmap, 64 * 8 * 4);
  k_map<<<1, 64>>>(dmap);
  std::vector<float> hm(64 * 8);
  (void)hipMemcpy(hm.data(), dmap, 64 * 8 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int h = 0; h < 2; ++h)
      for (int q = 0; q < 4; ++q) {
        const int g = l >> 4, i = l & 15, row = 8 * h + 4 * (g & 1) + q;
        const float want = (float)(row * 16 + i), got = hm[(l * 2 + h) * 4 + q];
        if (want != got) { if (bad < 8) printf("map mismatch lane %d h %d q %d: got %g want %g\n", l, h, q, got, want); ++bad; }
      }
  printf("ds_read_b64_tr_b16 lane map: %s (%d mismatches)\n", bad ? "WRONG" : "as documented", bad);
  // ---- 2. two-step dW contraction on split-bf16 pieces ----
  std::vector<float> dg(2 * 16 * NU), x(2 * 16 * NC), dW(NU * NC);
  srand(7);
  for (auto& v : dg) v = ((float)rand() / RAND_MAX - 0.5f) * 0.02f;
  for (auto& v : x) v = ((float)rand() / RAND_MAX - 0.5f) * 2.0f;
  float *ddg, *dx, *ddW;
  (void)hipMalloc(&ddg, dg.size() * 4); (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&ddW, dW.size() * 4);
  (void)hipMemcpy(ddg, dg.data(), dg.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  k_dw<<<1, 256>>>(ddg, dx, ddW);
  (void)hipMemcpy(dW.data(), ddW, dW.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0, max32 = 0;
  for (int u = 0; u < NU; ++u)
    for (int c = 0; c < NC; ++c) {
      double ref = 0; float f32 = 0.f;
      for (int s = 0; s < 2; ++s)
        for (int r = 0; r < 16; ++r) {
          ref += (double)dg[(s * 16 + r) * NU + u] * (double)x[(s * 16 + r) * NC + c];
          f32 = fmaf(dg[(s * 16 + r) * NU + u], x[(s * 16 + r) * NC + c], f32);
        }
      maxerr = fmax(maxerr, fabs(ref - dW[u * NC + c])); maxref = fmax(maxref, fabs(ref)); max32 = fmax(max32, fabs(ref - (double)f32));
    }
  printf("two-step dW on split-bf16 via transposed reads: max |err| %.3e (fp32 fmaf chain: %.3e), max |ref| %.3e -> %s\n", maxerr, max32, maxref,
         maxerr <= 4 * max32 + 1e-12 ? "OK" : "WRONG");
  return (bad || maxerr > 4 * max32 + 1e-12) ? 1 : 0;
}
