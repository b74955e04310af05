// Device-side helpers shared by every kernel file (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/msig.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// D(16x16) += A(16x4) * B(4x16), exact fp32 (v_mfma_f32_16x16x4_f32).
// lane l: A[row l&15][k l>>4], B[k l>>4][col l&15]; D: col l&15, rows (l>>4)*4 + reg.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// LDS-only barrier: does not drain outstanding global loads/stores (vmcnt).
// ---- split-bf16 ("bf16x3") contraction --------------------------------------------------------------------
// An fp32 value x is carried as three bf16 pieces x1 + x2 + x3 (x1 = top 16 bits of x, x2 = top 16 bits of x - x1, x3 likewise:
// 24 significant bits), and a product a*b as the six cross terms of order <= 2 accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16, smallest first.  Measured on K = 64 dot products of the recurrence's value ranges
// (tools/mfma_bf16x3.hip, profiles/r01_bf16x3_microbench.log): rms error 3.0e-8 against fp64 vs 4.8e-8 for the
// v_mfma_f32_16x16x4_f32 chain — at least fp32 accuracy — at 16.5 instead of 3 x 32 cycles per 16x16x32 block.
// Lane map of the 16x16x32 instruction: A[i = lane&15][k = 8*(lane>>4) + j], B[k = 8*(lane>>4) + j][n = lane&15],
// j = 0..7; D as for 16x16x4 (col = lane&15, rows 4*(lane>>4) + e).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// The pieces are taken by TRUNCATION (the upper 16 bits of the fp32 pattern; the remainder x - piece is exact): bit
// masks and subtractions at full VALU rate instead of round-to-nearest conversions, and no less accurate in the sum
// (rms error 3.5e-8 on the same test).
__device__ __forceinline__ __bf16 top16(float x, float& rem) {
  const uint32_t u = __float_as_uint(x) & 0xFFFF0000u;
  rem = x - __uint_as_float(u);
  const unsigned short h = (unsigned short)(u >> 16);
  __bf16 r;
  __builtin_memcpy(&r, &h, 2);
  return r;
}
__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  float r1, r2, r3;
  a = top16(x, r1); b = top16(r1, r2); c = top16(r2, r3);
}
// The same split for FOUR values at once, straight into the packed form the LDS planes take (bf16x4 per piece), in 7 instead of 11
// VALU instructions per pair of values: the packed pair of leading pieces P = (top16(a), top16(b)) is one v_perm_b32, and
// v_dot2c_f32_bf16 with the packed constant {-1, 0} resp. {0, -1} computes x - piece from it exactly (fp32 accumulate of an exact
// product: bit-identical to mask-and-subtract on 2^24 values incl. denormals, tools/split3_dot2_check.hip), so the mask and the
// separate subtraction go.  The constants must come from SGPRs: as a literal, {-1, 0} = 0x0000BF80 is folded to the inline
// constant -1.0, which this instruction reads as the fp32 pattern 0xBF800000 = {0, -1}.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t top_pair(float a, float b) {            // (bits(b) & 0xffff0000) | (bits(a) >> 16)
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& P0, uint32_t& P1, uint32_t& P2) {
  uint32_t klo, khi;
  asm("s_mov_b32 %0, 0xbf80\n\ts_mov_b32 %1, 0xbf800000" : "=s"(klo), "=s"(khi));
  const bf16x2 lo_m1 = __builtin_bit_cast(bf16x2, klo), hi_m1 = __builtin_bit_cast(bf16x2, khi);
  P0 = top_pair(a, b);
  a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, P0), lo_m1, a, false);
  b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, P0), hi_m1, b, false);
  P1 = top_pair(a, b);
  a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, P1), lo_m1, a, false);
  b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, P1), hi_m1, b, false);
  P2 = top_pair(a, b);
}
struct U2 { uint32_t x, y; };
__device__ __forceinline__ void split3_quad(const float (&v)[4], bf16x4 (&out)[3]) {
  U2 q[3];
  split3_pair(v[0], v[1], q[0].x, q[1].x, q[2].x);
  split3_pair(v[2], v[3], q[0].y, q[1].y, q[2].y);
#pragma unroll
  for (int pp = 0; pp < 3; ++pp) out[pp] = __builtin_bit_cast(bf16x4, q[pp]);
}
__device__ __forceinline__ void split3_quad(const f32x4& v, bf16x4 (&out)[3]) {
  const float t[4] = {v[0], v[1], v[2], v[3]};
  split3_quad(t, out);
}
// Negative controls of the parity tolerances (make negctl NEGCTL=k; never defined in the product library): the split-bf16 product
// WITHOUT its a1 * b1 cross term — a 2^-16 relative error per product — in ONE class of contractions, so that the parity gate is
// shown to catch a regression confined to it.  Every call site names its class: CT_BWD_REC the BPTT recurrence W_hh^T dgh, CT_DX the
// input gradient W_ih^T dgi, CT_DW the weight gradients, CT_FWD_REC the forward recurrence W_hh h (and gru_bwd_b6's recomputation of
// it), CT_FWD_PROJ the input projection W_ih x.  NEGCTL = 9 drops the term everywhere (round 3's control).
enum { CT_ANY = 0, CT_BWD_REC = 1, CT_DX = 2, CT_DW = 3, CT_FWD_REC = 4, CT_FWD_PROJ = 5 };
#ifdef MSIG_DROP_CROSS_TERM
template <int KIND> constexpr bool msig_drop_ct() { return MSIG_DROP_CROSS_TERM == 9 || MSIG_DROP_CROSS_TERM == KIND; }
#else
template <int KIND> constexpr bool msig_drop_ct() { return false; }
#endif
// acc += A . B over one 32-wide k block, A and B given as their three pieces ([0] = leading piece)
template <int KIND = CT_ANY>
__device__ __forceinline__ f32x4 mfma_bf16x3(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
  if constexpr (!msig_drop_ct<KIND>()) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
  return acc;
}

// ---- two-piece fp16 contraction ("f16x2"), round 5: the FORWARD recurrences and the layer-1 input projection ---------------------
// x S = x0 + x1 with x0 = fp16(x S), x1 = fp16(x S - x0) (round to nearest; S a power of two that puts x S in the middle of fp16's
// range, so that the low piece is a NORMAL fp16 number: 2 x 11 significant bits + the sign of the remainder = fp32's 24), and
//     sum a b  =  [ sum a0 b0 + a0 b1 + a1 b0 ] / (S_a S_b)         (a1 b1: 2^-22 relative, dropped)
// — THREE v_mfma_f32_16x16x32_f16 per 16x16x32 block instead of split-bf16's six, one accumulator (both operands pre-scaled, the
// three terms share the scale), exact power-of-two post-scale.  Measured on the forward pass's three contraction shapes against
// fp64 (tools/mfma_f16x2.hip, profiles/r05_f16x2_microbench.log): rms error 2.97e-8 / 6.8e-8 / 1.07e-7 (K = 64 recurrence / K = 128
// layer-1 projection / K = 32) against 4.83e-8 / 9.7e-8 / 1.19e-7 for the v_mfma_f32_16x16x4_f32 chain and 3.46e-8 / 1.03e-7 / 8.4e-8
// for split-bf16; 304.6 instead of 592.6 cycles per wave-step of three K = 64 tiles.
// Range is what decides where it may be used: an operand must satisfy |x| S < 65504.  h is in [-1, 1] by construction (a convex
// combination of tanh values), the layer-1 input is h times the dropout scale, and the weights get their scale from their own
// maximum when a wave loads its fragment — so the forward recurrences and the layer-1 projection qualify.  The layer-0 input
// (BatchNorm output: unbounded outliers) and every backward contraction (gate gradients: no bound, and rows of very different
// magnitude in one tile) stay on split-bf16, whose pieces have fp32's exponent range.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define F16X2_H_SCALE 4096.0f                 // |h| <= 1 -> |h S| <= 4096; the low piece is normal for |h| >= 2^-15
__device__ __forceinline__ void split2_quad(const f32x4& v, const float scale, f16x4& hi, f16x4& lo) {
  const f32x4 t = v * scale;
  hi = __builtin_convertvector(t, f16x4);
  const f32x4 back = __builtin_convertvector(hi, f32x4);
  lo = __builtin_convertvector(t - back, f16x4);
}
__device__ __forceinline__ void split2(const float x, const float scale, _Float16& hi, _Float16& lo) {
  const float t = x * scale;
  hi = (_Float16)t;
  lo = (_Float16)(t - (float)hi);
}
// power-of-two scale for a weight fragment whose largest magnitude over the WAVE is m: m S in [2^13, 2^14) (fp16's largest finite
// value is 65504); an all-zero fragment takes 2^12.  Exact: built from the exponent field.
__device__ __forceinline__ float f16x2_weight_scale(float m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  const int e = (int)((__float_as_uint(m) >> 23) & 0xFF);            // biased exponent of the maximum: m in [2^(e-127), 2^(e-126))
  if (e == 0) return 4096.0f;
  const int se = 127 + 13 - (e - 127);                               // S = 2^(13 - (e - 127))
  return __uint_as_float((uint32_t)(se < 1 ? 1 : (se > 254 ? 254 : se)) << 23);
}
// acc += A . B over one 32-wide k block, A and B as their two pieces ([0] = leading piece); the result carries the factor S_a S_b.
// Negative control (MSIG_DROP_CROSS_TERM): without a1 * b0 — a 2^-11 relative error per product.
template <int KIND = CT_ANY>
__device__ __forceinline__ f32x4 mfma_f16x2(const f16x8 (&a)[2], const f16x8 (&b)[2], f32x4 acc) {
  if constexpr (!msig_drop_ct<KIND>()) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);
  return acc;
}

// LDS plane swizzle shared by the split-bf16 kernels: rows 16 * odd dwords apart, 8-element (16-byte) chunks of row r XORed with
// g(r >> 2), g = [0,3,2,1].  Row reads by ds_read_b128 (lane = (row li, chunk lq)), the producers' 8-byte stores and transposed
// reads of 32-column blocks are all conflict-free (derivation: gru_bwd4.hip; exact-integer check: tools/dw32_check.hip).
__device__ __forceinline__ int quad_swz(int row) { return ((4 - (row >> 2)) & 3) * 8; }

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// v_exp_f32 / v_rcp_f32 based (1 ulp each): abs error ~1e-7, far inside the parity tolerance
__device__ __forceinline__ float sigmoidf_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }
// The GRU gate math of one lane (4 elements), written on 4-vectors so that the non-transcendental half compiles to
// packed fp32 instructions (v_pk_mul/add/fma_f32: two elements per issue slot) — VALU time adds to MFMA time on this
// part, so instruction count is what matters.  Same formulas as sigmoidf_fast / tanhf_fast:
//   r = sigmoid(a_r), z = sigmoid(a_z), n = tanh(a_in + r * a_hn), h' = n + z * (h - n)   [= (1-z) n + z h]
__device__ __forceinline__ void gru_gates(const f32x4& a_r, const f32x4& a_z, const f32x4& a_in, const f32x4& a_hn, const f32x4& h,
                                          f32x4& r, f32x4& z, f32x4& n, f32x4& hnew) {
  constexpr float L2E = 1.4426950408889634f;
  f32x4 er = a_r * (-L2E), ez = a_z * (-L2E);
#pragma unroll
  for (int e = 0; e < 4; ++e) { er[e] = __builtin_amdgcn_exp2f(er[e]); ez[e] = __builtin_amdgcn_exp2f(ez[e]); }
  er = er + 1.0f; ez = ez + 1.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = __builtin_amdgcn_rcpf(er[e]); z[e] = __builtin_amdgcn_rcpf(ez[e]); }
  f32x4 t = (a_in + r * a_hn) * (2.0f * L2E);
#pragma unroll
  for (int e = 0; e < 4; ++e) t[e] = __builtin_amdgcn_exp2f(t[e]);
  t = t + 1.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e) t[e] = __builtin_amdgcn_rcpf(t[e]);
  n = 1.0f - 2.0f * t;
  hnew = n + z * (h - n);
}

// The backward pass does not read n_t from the stash (round 2: the forward kernels are bound by their stash writes — layer 0
// moves bytes at 4.3 TB/s whatever their number: 0.71 ms without, 0.82 ms with two, 1.28 ms with four stash vectors per step —
// so the stash holds r, z and W_hn h + b_hn only); it recovers n_t from the step's own output, h_t = n + z (h_{t-1} - n):
//     n = (h_t - z h_{t-1}) / (1 - z),   clamped to tanh's range.
// The quotient loses accuracy as z -> 1 (error ~ eps |h| / (1 - z)), but every use of n in the backward pass carries the factor
// (1 - z):  dn = dh (1-z)(1 - n^2),  dz = dh (h_{t-1} - n) z (1-z)  — so the error that reaches a gradient stays ~ eps |dh|,
// the clamp bounds it by (1 - z) |dh| where the quotient is noise, and z = 1 exactly (1 - z = 0) gives 0 * finite = 0, as it should.
__device__ __forceinline__ float gru_n_from_h(float h_new, float h_prev, float z, float one_minus_z) {
  const float num = __builtin_fmaf(-z, h_prev, h_new);
  return __builtin_amdgcn_fmed3f(num * __builtin_amdgcn_rcpf(__builtin_fmaxf(one_minus_z, 1e-30f)), -1.0f, 1.0f);
}

__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
// Dropout: element e is kept iff byte (e&3) of fmix32((e>>2) ^ key) >= thr.
__device__ __forceinline__ uint32_t drop_word(uint32_t elem_idx, uint32_t key) { return fmix32((elem_idx >> 2) ^ key); }
__device__ __forceinline__ float drop_mul(uint32_t word, int byte, int thr, float scale) {
  return (((word >> (8 * byte)) & 0xFFu) >= (uint32_t)thr) ? scale : 0.0f;
}
__host__ __device__ __forceinline__ float drop_scale(int thr) { return thr >= 256 ? 0.0f : 256.0f / (256.0f - (float)thr); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- fold batching (msig_*_multi): one launch covers several independent models ("folds" of the LOSO loop) -------------
// Every buffer of fold f (parameters, gradients, Adam moments, BN state, workspace, input batch, labels) lives at the SAME
// offset inside a per-fold arena, arenas are `stride` bytes apart, and blockIdx.z selects the fold: every pointer a kernel
// receives is fold 0's and is shifted by slot[blockIdx.z] * stride at kernel entry (FOLD_BEGIN / FS).  The single-model entry
// points launch with n = 1, slot[0] = 0 (stride irrelevant).  Per-fold scalars (dropout keys, learning rate) are arrays
// indexed by blockIdx.z.
#define MSIG_FORM_AUTO (-1)      // internal: "no form named" (msig_batch.fwd_form / bwd_form == 0 and no environment default)
struct FoldCtx {
  int32_t n;
  int32_t slot[MSIG_MAX_FOLDS];
  int64_t stride;                         // bytes between consecutive arenas
  uint32_t key_gru[MSIG_MAX_FOLDS], key_head[MSIG_MAX_FOLDS];
  float lr_over_bc1[MSIG_MAX_FOLDS];      // Adam: lr / (1 - beta1^step) of each fold
  float inv_sqrt_bc2[MSIG_MAX_FOLDS];     //       1 / sqrt(1 - beta2^step) of each fold (folds may be at different step counts)
  int32_t form_folds;                     // fold count the GRU kernel forms are chosen for (msig_multi.form_folds; >= 1)
  int32_t fused_step;                     // host side: forward and backward of this step come from ONE descriptor (msig_train_step*)
};
__device__ __forceinline__ const void* msig_fold_addr(const void* p, int64_t off) { return p ? (const void*)((const char*)p + off) : p; }
#define FOLD_BEGIN const int64_t foff_ = (int64_t)fc.slot[blockIdx.z] * fc.stride
#define FS(p) p = (decltype(p))msig_fold_addr((const void*)(p), foff_)
inline FoldCtx single_fold(const msig_batch* b) {
  FoldCtx fc{};
  fc.n = 1; fc.stride = 0; fc.form_folds = 1;
  fc.key_gru[0] = b ? b->key_gru : 0; fc.key_head[0] = b ? b->key_head : 0;
  return fc;
}

struct StageDims {
  int B, C, T, K, L1, P1, L2, TP, Cr, NT;  // NT = batch tiles of 16 rows
};
__host__ __device__ inline StageDims make_dims(const msig_shape& s) {
  StageDims d;
  d.B = s.B; d.C = s.C; d.T = s.T; d.K = s.K;
  d.L1 = (s.T + 6 - 7) / 2 + 1;
  d.P1 = (d.L1 + 2 - 3) / 2 + 1;
  d.L2 = (d.P1 + 4 - 5) / 2 + 1;
  d.TP = (d.L2 + 2 - 3) / 2 + 1;
  d.Cr = s.C / 4;
  d.NT = (s.B + 15) / 16;
  return d;
}

#define MSIG_LAUNCH_CHECK()                          \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

// ---- optional per-kernel HIP-event timing (api.hip) -----------------------------
struct MsigProfScope {
  const char* name; hipStream_t st; void* rec;
  MsigProfScope(const char* name, hipStream_t st);
  ~MsigProfScope();
};
#define MSIG_K(name, st) MsigProfScope prof_scope__(name, st)

// ---- internal launchers (one per kernel file) ---------------------------------
struct WsPtrs {
  char* base;
  int64_t off[MSIG_NWS + 1];
  template <typename T> __host__ T* p(int region) const { return (T*)(base + off[region]); }
};

int launch_frontend_fwd(const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po, const FoldCtx& fc, hipStream_t st);
struct ColsumPlan;
int launch_frontend_bwd(const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po, ColsumPlan& plan, const FoldCtx& fc, hipStream_t st);
int launch_gru_fwd(const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po, const FoldCtx& fc, hipStream_t st);
int launch_gru_bwd(const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po, ColsumPlan& plan, const FoldCtx& fc, hipStream_t st);
int launch_head_fwd(const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po, const FoldCtx& fc, hipStream_t st);
int launch_head_bwd(const msig_batch* b, const float* dlogits, const StageDims& d, const WsPtrs& w, const int64_t* po, ColsumPlan& plan,
                    const FoldCtx& fc, hipStream_t st);
// head forward + CrossEntropy + head backward of a fused train step in one launch (few windows: see head.hip); false = not applicable
bool head_step_applies(const msig_batch* b, const StageDims& d);
int launch_head_step(const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po, ColsumPlan& plan, const FoldCtx& fc, hipStream_t st);
// Weight-gradient reductions.  Every backward kernel leaves per-workgroup partials in its OWN sub-region of
// MSIG_WS_GRAD_PART (nothing aliases), and only records what has to be summed: out[c] = sum_r part[r*stride +
// col0 + c], fp64 accumulation in a fixed order.  The whole backward pass is then reduced by ONE launch
// (blockIdx.y = job) instead of one launch per stage — at the reference's batch size a step is bound by the
// number of launches, not by their work.
struct ColsumJob { const float* part; int nrows, row_stride, col0, ncols; float* out; };
#define MSIG_MAX_JOBS 40
// The loss of a fused train step whose head ran as ONE kernel (head.hip head_step_kernel): summed, in ce_kernel's order, by one
// extra workgroup of the step's last launch.
struct LossFin { const float* logits; const int64_t* labels; float* lossbuf; double* lacc; int B, K; };
struct ColsumPlan {
  ColsumJob job[MSIG_MAX_JOBS];
  int n = 0;
  LossFin loss{nullptr, nullptr, nullptr, nullptr, 0, 0};
  bool add(const float* part, int nrows, int row_stride, int col0, int ncols, float* out) {
    if (n >= MSIG_MAX_JOBS) return false;
    if (ncols > 0 && nrows > 0) job[n++] = ColsumJob{part, nrows, row_stride, col0, ncols, out};
    return true;
  }
  bool add_in_place(float* grad, int ncols) {          // gradient written directly by its kernel: nothing to sum
    if (n >= MSIG_MAX_JOBS) return false;
    if (ncols > 0) job[n++] = ColsumJob{nullptr, 0, 0, 0, ncols, grad};
    return true;
  }
};
int launch_colsum_plan(const ColsumPlan& plan, const FoldCtx& fc, hipStream_t st);
// The same reduction with the Adam update of every reduced element applied in the same launch (the train step's
// last two launches in one).  Jobs with nrows == 0 are "gradient already in place" ranges (BN affine, gate weights).
struct AdamArgs { float *p, *g, *m, *v; float lr_over_bc1, inv_sqrt_bc2, b1, b2, eps, wd; };
int launch_colsum_adam_plan(const ColsumPlan& plan, const AdamArgs& ad, const FoldCtx& fc, hipStream_t st);
// sub-regions of MSIG_WS_GRAD_PART, in floats
struct PartOffsets { int64_t head, l1, l0, conv2, conv1, total; int gru_rows; };
PartOffsets part_offsets(const StageDims& d);

// Number of persistent workgroups used by reduction-style kernels; partial buffers are sized for it.
#define MSIG_PERSIST_WG 1024
#define MSIG_DW_WG 512
#define MSIG_CONV_DW_WG 1024      // conv weight-gradient kernels: 4 workgroups per CU hide their staging latency
