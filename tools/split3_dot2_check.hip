// Diagnostic: is a three-piece bf16 split whose residuals come from v_dot2c_f32_bf16 (r = x - piece, with the piece read as the
// packed pair the MFMA operand needs anyway) bit-identical to the mask-and-subtract split of msig_dev.h?
//   hipcc -O3 --offload-arch=gfx950 tools/split3_dot2_check.hip -o /tmp/split3_dot2_check && /tmp/split3_dot2_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t top_pair(float a, float b) {            // (bits(b) & 0xffff0000) | (bits(a) >> 16)
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ void split3_pair_dot2(float a, float b, uint32_t& P0, uint32_t& P1, uint32_t& P2) {
  // {-1, 0} as packed bf16 is 0x0000BF80; written as a literal the compiler folds it to the inline constant "-1.0", which this
  // instruction then reads as the fp32 pattern 0xBF800000 = {0, -1} (observed: tools/split3_dot2_check before this workaround)
  uint32_t lo_bits, hi_bits;
  asm volatile("s_mov_b32 %0, 0xbf80\n\ts_mov_b32 %1, 0xbf800000" : "=s"(lo_bits), "=s"(hi_bits));
  const bf16x2 lo_m1 = __builtin_bit_cast(bf16x2, lo_bits), hi_m1 = __builtin_bit_cast(bf16x2, hi_bits);
  P0 = top_pair(a, b);
  a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, P0), lo_m1, a, false);
  b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, P0), hi_m1, b, false);
  P1 = top_pair(a, b);
  a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, P1), lo_m1, a, false);
  b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, P1), hi_m1, b, false);
  P2 = top_pair(a, b);
}
__device__ __forceinline__ void split3_ref(float x, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  const uint32_t u0 = __float_as_uint(x) & 0xffff0000u; const float r1 = x - __uint_as_float(u0);
  const uint32_t u1 = __float_as_uint(r1) & 0xffff0000u; const float r2 = r1 - __uint_as_float(u1);
  p0 = u0 >> 16; p1 = u1 >> 16; p2 = __float_as_uint(r2) >> 16;
}
__global__ void check(const float* x, int n, unsigned long long* bad, float* first_bad) {
  unsigned long long nb = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; 2 * i + 1 < n; i += gridDim.x * blockDim.x) {
    const float a = x[2 * i], b = x[2 * i + 1];
    uint32_t P0, P1, P2, a0, a1, a2, b0, b1, b2;
    split3_pair_dot2(a, b, P0, P1, P2);
    split3_ref(a, a0, a1, a2); split3_ref(b, b0, b1, b2);
    const bool ok = P0 == ((b0 << 16) | a0) && P1 == ((b1 << 16) | a1) && P2 == ((b2 << 16) | a2);
    if (!ok) { if (!nb && !atomicAdd(bad + 1, 1ull)) { first_bad[0] = a; first_bad[1] = b; } ++nb; }
  }
  if (nb) atomicAdd(bad, nb);
}
int main() {
  const int n = 1 << 24;
  std::vector<float> h(n);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int i = 0; i < n; ++i) {
    const uint64_t r = rnd();
    uint32_t bits;
    switch (i & 7) {
      case 0: bits = (uint32_t)r; break;                                                   // any bit pattern (incl. denormals, huge)
      case 1: { float f = (float)((double)(r >> 11) / 9007199254740992.0 * 2.0 - 1.0); memcpy(&bits, &f, 4); } break;      // (-1, 1)
      case 2: { float f = (float)((double)(r >> 11) / 9007199254740992.0 * 1e-3); memcpy(&bits, &f, 4); } break;            // small gradients
      case 3: bits = ((uint32_t)r & 0x807fffffu) | (((uint32_t)(r >> 40) % 40 + 100) << 23); break;                          // exponents 2^-27 .. 2^12
      case 4: bits = (uint32_t)r & 0xffff0000u; break;                                     // exactly one piece
      case 5: bits = ((uint32_t)r & 0x80000000u) | ((uint32_t)(r >> 32) & 0x007fffffu); break;      // denormals
      case 6: bits = ((uint32_t)r & 0xff800000u) | 0x007fffffu; break;                     // all-ones mantissa
      default: bits = (uint32_t)r & 0xffffff00u; break;
    }
    float f; memcpy(&f, &bits, 4);
    if (std::isnan(f) || std::isinf(f)) f = 1.5f;
    h[i] = f;
  }
  float *d, *fb; unsigned long long* bad;
  hipMalloc(&d, n * 4); hipMalloc(&bad, 16); hipMalloc(&fb, 8);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 16);
  check<<<1024, 256>>>(d, n, bad, fb);
  unsigned long long hb[2]; float hfb[2];
  hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(hfb, fb, 8, hipMemcpyDeviceToHost);
  printf("pairs checked %d, mismatching pairs %llu", n / 2, hb[0]);
  if (hb[0]) printf("  first: a=%a b=%a", hfb[0], hfb[1]);
  printf("\n");
  // same again restricted to normal numbers of moderate size (what the kernels see)
  for (int i = 0; i < n; ++i) if (!(std::fabs(h[i]) > 1e-30f && std::fabs(h[i]) < 1e30f)) h[i] = 0.25f + 1e-3f * (i & 1023);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 16);
  check<<<1024, 256>>>(d, n, bad, fb);
  hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(hfb, fb, 8, hipMemcpyDeviceToHost);
  printf("normal range only: mismatching pairs %llu", hb[0]);
  if (hb[0]) printf("  first: a=%a b=%a", hfb[0], hfb[1]);
  printf("\n");
  return 0;
}
