// Building blocks of the software-pipelined GRU backward kernels (gru_bwd4.hip: gru_bwd_b4 / b5 / seq4; gru_bwd6.hip: gru_bwd_b6):
// slot fences, register-class pins, the staged three-piece split, split-bf16 MFMA terms, transposed LDS reads.
#pragma once
#include "gru_args.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDS_AS __attribute__((address_space(3)))
// A slot boundary.  sched_barrier(0) fences the machine scheduler only: instruction selection linearises a basic block's DAG by
// register pressure and moves every node without a chain (MFMAs, VALU) to its use, across any number of fences (first version of
// this kernel: 18 MFMAs back to back, then 26 VALU in a row).  What does hold an operation in its slot is a dependence on an
// ordered node: every MFMA and every filler operation passes its RESULT through an empty `asm volatile` (PINV / PINA — no code,
// but volatile asms keep their program order), and the boundary is a volatile asm with a memory clobber, which orders the LDS
// and global memory operations of the slots as well.
#define FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PIN_ACC(v) asm volatile("" : "+a"(v))
#define PINA(v) asm volatile("" : "+a"(v))
#define PINV(v) asm volatile("" : "+v"(v))
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")

template <int N> using ic = std::integral_constant<int, N>;
template <typename F, int... Is> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(ic<Is>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ bf16x4 lds_tr_read4(const __bf16* p) {      // ds_read_b64_tr_b16; EXEC must be all ones
  typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 v4;
  const v4 r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS v4*)p);
  return __builtin_bit_cast(bf16x4, r);
}
// term T of the six-term split-bf16 product (smallest cross terms first, as mfma_bf16x3)
template <int T, int KIND> __device__ __forceinline__ f32x4 mf16(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 acc) {
  constexpr int ai[6] = {2, 0, 1, 1, 0, 0}, bi[6] = {0, 2, 1, 0, 1, 0};
  if constexpr (T == 2 && msig_drop_ct<KIND>()) return acc;       // negative controls of the parity tolerances only (msig_dev.h)
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ai[T]], b[bi[T]], acc, 0, 0, 0);
}
template <int T, int KIND> __device__ __forceinline__ f32x16 mf32(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc) {
  constexpr int ai[6] = {2, 0, 1, 1, 0, 0}, bi[6] = {0, 2, 1, 0, 1, 0};
  if constexpr (T == 2 && msig_drop_ct<KIND>()) return acc;
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ai[T]], b[bi[T]], acc, 0, 0, 0);
}
__device__ __forceinline__ uint32_t top_pair_u(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // (b & 0xffff0000) | (a >> 16)

// One instruction of the three-piece split of a PAIR of fp32 values into packed bf16 pairs, plain VALU only (and / sub / perm:
// these hide in MFMA gaps, v_dot2c_f32_bf16 does not).  Same pieces, bit for bit, as split3_pair (msig_dev.h).  11 stages.
struct SplitPair { float a, b; uint32_t ta, tb, P[3]; };
template <int S> __device__ __forceinline__ void split_stage(SplitPair& s) {
  if constexpr (S == 0) { s.P[0] = top_pair_u(__float_as_uint(s.a), __float_as_uint(s.b)); PINV(s.P[0]); }
  if constexpr (S == 1 || S == 6) { s.ta = __float_as_uint(s.a) & 0xFFFF0000u; PINV(s.ta); }
  if constexpr (S == 2 || S == 7) { s.tb = __float_as_uint(s.b) & 0xFFFF0000u; PINV(s.tb); }
  if constexpr (S == 3 || S == 8) { s.a = s.a - __uint_as_float(s.ta); PINV(s.a); }
  if constexpr (S == 4 || S == 9) { s.b = s.b - __uint_as_float(s.tb); PINV(s.b); }
  if constexpr (S == 5) { s.P[1] = top_pair_u(__float_as_uint(s.a), __float_as_uint(s.b)); PINV(s.P[1]); }
  if constexpr (S == 10) { s.P[2] = top_pair_u(__float_as_uint(s.a), __float_as_uint(s.b)); PINV(s.P[2]); }
}
constexpr int SPLIT_STAGES = 11;


template <int E> __device__ __forceinline__ float f4e(const float4& v) {
  if constexpr (E == 0) return v.x; else if constexpr (E == 1) return v.y; else if constexpr (E == 2) return v.z; else return v.w;
}
template <int E> __device__ __forceinline__ void f4mul(float4& v, float m) {
  if constexpr (E == 0) v.x *= m; else if constexpr (E == 1) v.y *= m; else if constexpr (E == 2) v.z *= m; else v.w *= m;
}
template <int H> __device__ __forceinline__ void put_half(bf16x8& f, const bf16x4 v) {
  f[4 * H + 0] = v[0]; f[4 * H + 1] = v[1]; f[4 * H + 2] = v[2]; f[4 * H + 3] = v[3];
}

