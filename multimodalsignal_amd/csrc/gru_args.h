// Declarations shared by the GRU kernel files (gru.hip, gru_bwd4.hip): argument blocks, fold-batching macros, diagnostic stamps.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>
#include "msig_dev.h"

#define HS 68    // LDS row stride (floats) of the 16x64 state tile
#define DGS 196  // LDS row stride of the 16x192 dgh tile (backward recurrence)
#define RS 272   // LDS row stride of the 16x256 dg tile (bulk kernels)

// In-kernel phase stamps: compiled only into the diagnostic library (make stamps); never in the product .so.
#ifdef MSIG_STAMPS
#define STAMP_DECL unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = stamp_now()
#define STAMP(i) do { const unsigned long long tn_ = stamp_now(); ph_[i] += tn_ - tprev_; tprev_ = tn_; } while (0)
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#else
#define STAMP_DECL
#define STAMP(i)
#endif

// compile-time unrolled loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <typename F, int... Is> __device__ __forceinline__ void sfor_n_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor_n(F&& f) { sfor_n_impl(f, std::make_integer_sequence<int, N>{}); }

struct GruDir {
  const float *Wih, *Whh, *bih, *bhh;
  int t_start, t_sign, n_steps;     // time index of step s: t = t_start + t_sign*s
  float* h;                         // h[b*h_bs + t*h_ts + h_col + u]
  int64_t h_bs, h_ts;
  int h_col;
  float* h_last;                    // optional copy of the final state: h_last[b*hl_bs + hl_col + u]
  int64_t hl_bs;
  int hl_col;
  float4* stash;                    // [(tile*n_steps + s)*4 + w][gate][lane] float4; NULL in eval
  // backward only
  const float* dh;                  // upstream gradient, see dh_mode
  int64_t dh_bs, dh_ts;
  int dh_col;
  int dh_mode;                      // 0: every step, dropout-masked (layer 0); 1: only the last step
  float* dx;                        // dx[b*dx_bs + t*dx_ts + k]
  int64_t dx_bs, dx_ts;
  int dx_accumulate;
  float* part;                      // dW partials [wg][192*I + 192*64 + 256]
};

struct GruArgs {
  GruDir dir[2];
  const float* x;                   // x[b*x_bs + t*x_ts + k]
  int64_t x_bs, x_ts;
  int B;
  int drop_thr;                     // dropout on x (layer-1 input) / on dh (layer-0 upstream grad)
  uint32_t drop_key;
  float drop_scale;
  float4* gi;                       // latency form only: input projections [(tile*n_steps + s)*4 + w][gate r,z,n][lane]
  size_t gi_dir_stride;             // float4 elements between the two directions' gi blocks
  unsigned long long* dbg;          // diagnostic stamps (MSIG_STAMPS builds only)
  int x_drop_thr;                   // fused backward only: dropout of the x tile (layer 1), independent of the dh mask
  uint32_t x_drop_key;
  float x_drop_scale;
  int stash_skip_hn;                // forward, layer 0: the backward form recomputes W_hn h + b_hn (gru_bwd_b6) — store r, z only
};

// Fold batching (msig_dev.h FoldCtx): the latency-form kernels run several independent models in one launch, blockIdx.z = fold.
// Every pointer of the argument block is fold 0's; the kernel shifts the ones it uses into this fold's arena (a by-value copy
// of ITS direction's GruDir — never of the whole argument block, whose dynamic indexing would land in scratch) and takes this
// fold's dropout key.
__device__ __forceinline__ void fold_dir(GruDir& g, const FoldCtx& fc) {
  FOLD_BEGIN;
  FS(g.Wih); FS(g.Whh); FS(g.bih); FS(g.bhh); FS(g.h); FS(g.h_last); FS(g.stash); FS(g.dh); FS(g.dx); FS(g.part);
}
#define FOLD_GRU_ARGS                                                         \
  GruDir D = a.dir[blockIdx.y];                                               \
  fold_dir(D, fc);                                                            \
  const float* ax_ = a.x; float4* agi_ = a.gi;                                \
  { FOLD_BEGIN; FS(ax_); FS(agi_); }                                          \
  [[maybe_unused]] const uint32_t akey_ = fc.key_gru[blockIdx.z]

// The throughput-form kernels come in two instantiations: FOLDS = false is the single-model kernel (arguments read straight from
// the kernarg segment); FOLDS = true shifts every pointer to the arena of fold blockIdx.z and takes that fold's dropout key —
// same arithmetic, so a fold's numbers are bit-identical in a fold batch and alone.
#define FOLD_GRU_ARGS_IF(FOLDS)                                                            \
  GruDir Dv_; const float* ax_ = a.x;                                                      \
  [[maybe_unused]] uint32_t akey_ = a.drop_key; [[maybe_unused]] uint32_t axkey_ = a.x_drop_key;  \
  if constexpr (FOLDS) {                                                                   \
    Dv_ = a.dir[blockIdx.y]; fold_dir(Dv_, fc);                                            \
    FOLD_BEGIN; FS(ax_);                                                                   \
    akey_ = axkey_ = fc.key_gru[blockIdx.z];                                               \
  }                                                                                        \
  const GruDir& D = FOLDS ? Dv_ : a.dir[blockIdx.y]

