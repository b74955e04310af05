// GRU temporal encoder for gfx950: nn.GRU(32, 64, num_layers=2, bidirectional) of
// models.py:56-63,78 and its autograd backward (trainer.py:148), as fp32-MFMA kernels.
//
// Decomposition (all kernels: 256 threads = 4 waves; one workgroup owns a TILE of 16
// batch rows for one direction of one layer):
//   D-layout: v_mfma_f32_16x16x4_f32 leaves, in lane l = (lq = l>>4, li = l&15), register e,
//   the value for batch row li and hidden unit  u = w*16 + lq*4 + e  (w = wave id).  Gate
//   pre-activations r,z,n for that (row, unit) therefore sit in the same lane, so all gate
//   math is lane-local; the recurrent state only crosses lanes once per step, through a
//   16x64 fp32 tile in LDS that feeds the next step's B operand.
//   Weights are MFMA A operands held in registers for the whole sequence:
//     lane (li,lq), k-step m  <->  W[row = gate*64 + w*16 + li][col = lq*(Kdim/4) + m].
//
//   gru_fwd_seq<I>  input projection fused with the recurrence (no gi tensor in HBM);
//                   writes h_t, and in training the gate stash (r,z,n,W_hn h+b_hn).
//   gru_bwd_seq4    BPTT recurrence: dh_{t-1} = dh_t*z + W_hh^T dgh_t (gru_bwd4.hip); overwrites the
//                   stash in place with (dr_pre, dz_pre, dn_pre, dhn_pre).
//   gru_bwd_dx<I>   dx_t = W_ih^T dgi_t  (bulk over all (row, t)).
//   gru_bwd_dw<I>   dW_ih, dW_hh, db_ih, db_hh as split-K partials + colsum.
//
// The top layer's reverse direction is evaluated for ONE step only (t = T'-1, h0 = 0):
// that is all outputs[:, -1, :] (models.py:79) consumes; it runs through the same kernels
// with n_steps = 1.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <type_traits>
#include "msig_dev.h"

#include "gru_args.h"
#include "gru_bwd4.h"
#include "gru_dw2.h"

template <int KI, bool DROP>
__device__ __forceinline__ void load_x_operand(float (&xB)[KI], uint32_t (&xw)[DROP ? KI / 4 : 1], const float* __restrict__ xp,
                                               uint32_t e0, uint32_t key) {
  // B operand of the input projection: KI contiguous floats at xp (x[b][t][lq*KI ..]); e0 is the flat
  // element index of xp[0].  Only ISSUES the loads (plus the data-independent dropout hash words):
  // the mask is applied by apply_x_mask at the point of use one step later, so no consumer sits next to
  // the loads.  The row is always valid (callers clamp): predicated loads cost a branch + vmcnt(0).
#pragma unroll
  for (int v = 0; v < KI / 4; ++v) {
    const float4 q = *(const float4*)(xp + 4 * v);
    if constexpr (DROP) xw[v] = drop_word(e0 + 4 * v, key);
    xB[4 * v + 0] = q.x; xB[4 * v + 1] = q.y; xB[4 * v + 2] = q.z; xB[4 * v + 3] = q.w;
  }
}
template <int KI, bool DROP>
__device__ __forceinline__ void apply_x_mask(float (&xB)[KI], const uint32_t (&xw)[DROP ? KI / 4 : 1], int thr, float scale) {
  if constexpr (DROP) {   // branch-free: thr == 0 keeps everything with scale 1
#pragma unroll
    for (int v = 0; v < KI / 4; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e) xB[4 * v + e] *= drop_mul(xw[v], e, thr, scale);
  }
}

// ------------------------------------------------------------------------------------
// Forward recurrence, input projection fused.
// ------------------------------------------------------------------------------------
// Layer 0 (I = 32) must stay within 128 VGPRs: its grid is 4 workgroups per CU and a 129th register
// would drop residency to 3, i.e. a ragged second round (measured: 1.63 ms vs 1.2 ms).
// fp32-MFMA throughput form: the input projection W_ih x_t is fused into the step (no gi tensor in HBM).  Kept as the
// MSIG_GRU_FWD=fp32 alternative of gru_fwd_b3 (split-bf16 MFMA), which is the default above 192 batch tiles.
template <int I, bool STASH>
__global__ __launch_bounds__(256, (I == 32) ? 4 : 2) void gru_fwd_seq(const GruArgs a) {
  constexpr int KI = I / 4;
  constexpr bool DROP = (I == 128);                 // only the layer-1 input carries the inter-layer dropout
  // Layer 1 (I = 128) would need 144 weight VGPRs per lane; at 2 waves/SIMD that spills.  Its W_hh
  // operands (48 per lane) therefore live in LDS in a lane-linear image [gate][wave][m/4][lane][4]
  // (one conflict-free ds_read_b128 per 4 k-steps); W_ih stays in VGPRs.  48 KiB + state tile per
  // workgroup still leaves 2 workgroups per CU.
  constexpr bool HH_LDS = (I == 128);
  // Layer 0 (I = 32) keeps W_hh in VGPRs and moves W_ih (24 per lane) to LDS instead, which brings it
  // under the 128-VGPR line for 4 workgroups per CU (34 KiB of LDS each).
  constexpr bool IH_LDS = (I == 32);
  // Layer 1's input is the inter-layer-dropped layer-0 output.  Every wave needs the whole 16 x 128 x tile as its B
  // operand, so with per-lane loads each of the four waves hashed and masked all 32 of its lane's values: 190 of the
  // kernel's 245 VALU instructions per wave-step, and VALU time adds to fp32-MFMA time on this part.  Instead the
  // workgroup stages the tile once per step through LDS: two float4 per thread are loaded, masked (one hash each)
  // and written a step ahead; the waves read their operands back with eight ds_read_b128.
  constexpr bool XLDS = (I == 128);
  constexpr int XSS = 132;                              // row stride (floats) of the staged x tile
  __shared__ __attribute__((aligned(16))) float xs_[XLDS ? 2 * 16 * XSS : 4];
  __shared__ __attribute__((aligned(16))) float hbuf[2][16][HS];
  __shared__ __attribute__((aligned(16))) float whh_s[HH_LDS ? 3 * 4 * 4 * 64 * 4 : 4];
  __shared__ __attribute__((aligned(16))) float wih_s[IH_LDS ? 3 * 4 * (KI / 4) * 64 * 4 : 4];
  __shared__ __attribute__((aligned(16))) float bias_s[4][64];          // [kind r,z,in,hn][unit]
  const GruDir& D = a.dir[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int tile = blockIdx.x, b = tile * 16 + li;
  const bool valid = b < a.B;
  const int bl = valid ? b : a.B - 1;      // row used for loads (rows >= B replay the last row; never stored)
  const int u0 = w * 16 + lq * 4;

  // A operands: weights, resident for the whole sequence
  float Ahh[HH_LDS ? 1 : 3][HH_LDS ? 1 : 16], Aih[IH_LDS ? 1 : 3][IH_LDS ? 1 : KI];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const float* wr = D.Whh + (size_t)(g * 64 + w * 16 + li) * 64 + lq * 16;
    if constexpr (HH_LDS) {
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4)
        *(float4*)&whh_s[((((g * 4 + w) * 4 + m4) * 64) + lane) * 4] = *(const float4*)(wr + 4 * m4);
    } else {
#pragma unroll
      for (int m = 0; m < 16; ++m) Ahh[g][m] = wr[m];
    }
    const float* wi = D.Wih + (size_t)(g * 64 + w * 16 + li) * I + lq * KI;
    if constexpr (IH_LDS) {
#pragma unroll
      for (int m4 = 0; m4 < KI / 4; ++m4)
        *(float4*)&wih_s[((((g * 4 + w) * (KI / 4) + m4) * 64) + lane) * 4] = *(const float4*)(wi + 4 * m4);
    } else {
#pragma unroll
      for (int m = 0; m < KI; ++m) Aih[g][m] = wi[m];
    }
  }
  if (tid < 64) {
    bias_s[0][tid] = D.bih[tid] + D.bhh[tid];
    bias_s[1][tid] = D.bih[64 + tid] + D.bhh[64 + tid];
    bias_s[2][tid] = D.bih[128 + tid];
    bias_s[3][tid] = D.bhh[128 + tid];
  }
  for (int i = tid; i < 2 * 16 * HS; i += 256) (&hbuf[0][0][0])[i] = 0.f;
  // staged-x state (XLDS): thread -> two float4 pieces (row, c4) of the tile; running pointers like everything else
  const float* xq[2] = {nullptr, nullptr};
  uint32_t xqe[2] = {0, 0};
  float4 xv[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  int xoff[2] = {0, 0};
  auto stage_x = [&](int buf) {      // mask the two loaded pieces and put them into xs_[buf]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t wd = drop_word(xqe[j], a.drop_key);
      float4 q = xv[j];
      q.x *= drop_mul(wd, 0, a.drop_thr, a.drop_scale); q.y *= drop_mul(wd, 1, a.drop_thr, a.drop_scale);
      q.z *= drop_mul(wd, 2, a.drop_thr, a.drop_scale); q.w *= drop_mul(wd, 3, a.drop_thr, a.drop_scale);
      *(float4*)&xs_[buf * 16 * XSS + xoff[j]] = q;
    }
  };
  if constexpr (XLDS) {
    const int64_t xs0 = (int64_t)D.t_sign * a.x_ts;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + 256 * j, row = idx >> 5, c4 = idx & 31;
      const int br = min(tile * 16 + row, a.B - 1);
      const int64_t e0 = (int64_t)br * a.x_bs + (int64_t)D.t_start * a.x_ts + 4 * c4;
      xq[j] = a.x + e0; xqe[j] = (uint32_t)e0; xoff[j] = row * XSS + 4 * c4;
      xv[j] = *(const float4*)xq[j];
    }
    stage_x(0);                                           // x of step 0
    if (D.n_steps > 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { xq[j] += xs0; xqe[j] += (uint32_t)xs0; }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) xv[j] = *(const float4*)xq[j];      // x of step 1 (or a harmless reload)
  }
  __syncthreads();   // bias_s / weight images / the first x tile are read by OTHER waves in step 0, before the loop's first barrier

  // Everything the loop needs from the argument block is copied out once, and all addresses are
  // per-lane running pointers advanced by a constant stride per step: re-deriving them from the
  // kernel arguments (scalar-cache reloads + 64-bit multiplies) cost ~2000 cycles per step.
  const int n_steps = D.n_steps;
  const int thr = a.drop_thr; const uint32_t key = a.drop_key; const float dscale = a.drop_scale;
  const int64_t xstep = (int64_t)D.t_sign * a.x_ts, hstep = (int64_t)D.t_sign * D.h_ts;
  const float* xp = a.x + (int64_t)bl * a.x_bs + (int64_t)D.t_start * a.x_ts + lq * KI;
  uint32_t xe = (uint32_t)((int64_t)bl * a.x_bs + (int64_t)D.t_start * a.x_ts + lq * KI);
  float* hptr = D.h + (int64_t)bl * D.h_bs + (int64_t)D.t_start * D.h_ts + D.h_col + u0;
  float4* sp = STASH ? D.stash + ((size_t)((size_t)tile * n_steps) * 4 + w) * 4 * 64 + lane : nullptr;
  f32x4 hprev = {0.f, 0.f, 0.f, 0.f};
  float xB[KI];
  uint32_t xw[DROP ? KI / 4 : 1];
  if constexpr (!XLDS) load_x_operand<KI, DROP>(xB, xw, xp, xe, key);
  int cur = 0;
  STAMP_DECL;
  for (int s = 0; s < n_steps; ++s) {
    STAMP(0);
    f32x4 acc_r, acc_z, acc_in, acc_hn = *(const f32x4*)&bias_s[3][u0];
    {
      if constexpr (XLDS) {
        // x of step s+1 (loaded one iteration ago) goes into the other buffer — every wave is past the barrier that
        // followed its last read of it — and the loads for step s+2 are issued; both sit a whole step from their use
        stage_x((s + 1) & 1);
        if (s + 2 < n_steps) {
#pragma unroll
          for (int j = 0; j < 2; ++j) { xq[j] += xstep; xqe[j] += (uint32_t)xstep; }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) xv[j] = *(const float4*)xq[j];
#pragma unroll
        for (int v = 0; v < KI / 4; ++v) {
          const float4 q = *(const float4*)&xs_[(s & 1) * 16 * XSS + li * XSS + lq * KI + 4 * v];
          xB[4 * v] = q.x; xB[4 * v + 1] = q.y; xB[4 * v + 2] = q.z; xB[4 * v + 3] = q.w;
        }
      } else {
        apply_x_mask<KI, DROP>(xB, xw, thr, dscale);
      }
      acc_r = *(const f32x4*)&bias_s[0][u0]; acc_z = *(const f32x4*)&bias_s[1][u0]; acc_in = *(const f32x4*)&bias_s[2][u0];
    }
    {
#pragma unroll
    for (int v = 0; v < KI / 4; ++v) {
      float4 ar, az, an;
      if constexpr (IH_LDS) {
        ar = *(const float4*)&wih_s[((((0 * 4 + w) * (KI / 4) + v) * 64) + lane) * 4];
        az = *(const float4*)&wih_s[((((1 * 4 + w) * (KI / 4) + v) * 64) + lane) * 4];
        an = *(const float4*)&wih_s[((((2 * 4 + w) * (KI / 4) + v) * 64) + lane) * 4];
      } else {
        ar = make_float4(Aih[0][4 * v], Aih[0][4 * v + 1], Aih[0][4 * v + 2], Aih[0][4 * v + 3]);
        az = make_float4(Aih[1][4 * v], Aih[1][4 * v + 1], Aih[1][4 * v + 2], Aih[1][4 * v + 3]);
        an = make_float4(Aih[2][4 * v], Aih[2][4 * v + 1], Aih[2][4 * v + 2], Aih[2][4 * v + 3]);
      }
      const float wr_[4] = {ar.x, ar.y, ar.z, ar.w}, wz_[4] = {az.x, az.y, az.z, az.w}, wn_[4] = {an.x, an.y, an.z, an.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc_r = mfma16(wr_[e], xB[4 * v + e], acc_r);
        acc_z = mfma16(wz_[e], xB[4 * v + e], acc_z);
        acc_in = mfma16(wn_[e], xB[4 * v + e], acc_in);
      }
    }
    STAMP(1);
    if constexpr (!XLDS) {
      if (s + 1 < n_steps) { xp += xstep; xe += (uint32_t)xstep; }                  // last step: harmless reload
      load_x_operand<KI, DROP>(xB, xw, xp, xe, key);                                    // prefetch for step s+1
    }
    }
    STAMP(2);
    lds_barrier();   // h_{s-1} from every wave is in hbuf[cur]
    STAMP(3);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float4 q = *(const float4*)&hbuf[cur][li][lq * 16 + 4 * v];
      const float hq[4] = {q.x, q.y, q.z, q.w};
      float4 ar, az, an;
      if constexpr (HH_LDS) {
        ar = *(const float4*)&whh_s[((((0 * 4 + w) * 4 + v) * 64) + lane) * 4];
        az = *(const float4*)&whh_s[((((1 * 4 + w) * 4 + v) * 64) + lane) * 4];
        an = *(const float4*)&whh_s[((((2 * 4 + w) * 4 + v) * 64) + lane) * 4];
      } else {
        ar = make_float4(Ahh[0][4 * v], Ahh[0][4 * v + 1], Ahh[0][4 * v + 2], Ahh[0][4 * v + 3]);
        az = make_float4(Ahh[1][4 * v], Ahh[1][4 * v + 1], Ahh[1][4 * v + 2], Ahh[1][4 * v + 3]);
        an = make_float4(Ahh[2][4 * v], Ahh[2][4 * v + 1], Ahh[2][4 * v + 2], Ahh[2][4 * v + 3]);
      }
      const float wr_[4] = {ar.x, ar.y, ar.z, ar.w}, wz_[4] = {az.x, az.y, az.z, az.w}, wn_[4] = {an.x, an.y, an.z, an.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc_r = mfma16(wr_[e], hq[e], acc_r);
        acc_z = mfma16(wz_[e], hq[e], acc_z);
        acc_hn = mfma16(wn_[e], hq[e], acc_hn);
      }
    }
    STAMP(4);
    f32x4 r, z, n, hn;
    gru_gates(acc_r, acc_z, acc_in, acc_hn, hprev, r, z, n, hn);
    hprev = hn;
    STAMP(5);
    *(float4*)&hbuf[cur ^ 1][li][u0] = make_float4(hn[0], hn[1], hn[2], hn[3]);
    // Unconditional stores (a fixed number per step lets the compiler wait for the x prefetch with a
    // counted vmcnt instead of draining the stores): rows >= B replay row B-1 bit for bit, so they
    // store identical values to row B-1's address.
    *(float4*)hptr = make_float4(hn[0], hn[1], hn[2], hn[3]);
    hptr += hstep;
    if constexpr (STASH) {
      sp[0 * 64] = make_float4(r[0], r[1], r[2], r[3]);
      sp[1 * 64] = make_float4(z[0], z[1], z[2], z[3]);
      sp[3 * 64] = make_float4(acc_hn[0], acc_hn[1], acc_hn[2], acc_hn[3]);
      sp += 4 * 4 * 64;
    }
    STAMP(6);
    cur ^= 1;
  }
#ifdef MSIG_STAMPS
  if (a.dbg && tid == 0 && blockIdx.y == 0 && blockIdx.x < 256)
    for (int i = 0; i < 8; ++i) a.dbg[(size_t)blockIdx.x * 8 + i] = ph_[i];
#endif
  if (D.h_last != nullptr && valid)
    *(float4*)(D.h_last + (int64_t)b * D.hl_bs + D.hl_col + u0) = make_float4(hprev[0], hprev[1], hprev[2], hprev[3]);
}

// ------------------------------------------------------------------------------------
// Forward recurrence with fused input projection on split-bf16 MFMA (msig_dev.h) — the throughput form.
// Per wave-step layer 1 needs 108 bf16 MFMAs (~1780 cycles) where the fp32 kernel needs 144 fp32 MFMAs (4608), and
// layer 0 54 (~890) instead of 72 (2304), at an error no larger than the fp32 chain's.  The three-piece weights are
// 1.5x the fp32 ones (216 VGPRs per lane for layer 1): they live in registers for the whole sequence and the kernel
// runs ONE workgroup per CU (512 registers per lane at one wave per SIMD) — with the contraction 2.6x shorter there
// is little left that a second workgroup could hide, the rest of a step being VALU, DS and memory instructions
// that only add up on this part (tools/mfma_coissue.hip).
// Both operand tiles of a step cross lanes as three bf16 planes in LDS: the input tile x_t (16 rows x I; for layer 1
// the inter-layer dropout mask is applied on the way in), staged cooperatively one step ahead — two (I = 128) or
// half a (I = 32) float4 per thread — and the state h_{t-1}, split by its producers.  One barrier per step.
// ------------------------------------------------------------------------------------
template <int I, bool STASH>
__global__ __launch_bounds__(256, 1) void gru_fwd_b3(const GruArgs a) {
  constexpr int NKX = I / 32;                           // 32-wide k blocks of the input
  constexpr bool DROP = (I == 128);
  constexpr int HSB = 72, XSB = I + 8;                  // plane row strides in bf16 elements (16-byte aligned rows)
  constexpr int NXP = (16 * I / 4 + 255) / 256;         // float4 pieces of the x tile per thread (1 or 2)
  __shared__ __attribute__((aligned(16))) __bf16 hb[2][3][16][HSB];
  __shared__ __attribute__((aligned(16))) __bf16 xb[2][3][16][XSB];
  const GruDir& D = a.dir[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int tile = blockIdx.x, b = tile * 16 + li;
  const bool valid = b < a.B;
  const int bl = valid ? b : a.B - 1;      // rows >= B replay the last row bit for bit (stores hit the same address)
  const int u0 = w * 16 + lq * 4;
  const int n_steps = D.n_steps;

  // ---- A operands, split once: rows of this wave's 16 units per gate ----
  bf16x8 Ah[3][2][3], Ai[3][NKX][3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const float* wr = D.Whh + (size_t)(g * 64 + w * 16 + li) * 64 + kb * 32 + lq * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) { __bf16 p0, p1, p2; split3(wr[j], p0, p1, p2); Ah[g][kb][0][j] = p0; Ah[g][kb][1][j] = p1; Ah[g][kb][2][j] = p2; }
    }
#pragma unroll
    for (int kb = 0; kb < NKX; ++kb) {
      const float* wi = D.Wih + (size_t)(g * 64 + w * 16 + li) * I + kb * 32 + lq * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) { __bf16 p0, p1, p2; split3(wi[j], p0, p1, p2); Ai[g][kb][0][j] = p0; Ai[g][kb][1][j] = p1; Ai[g][kb][2][j] = p2; }
    }
  }
  f32x4 b_r, b_z, b_in, b_hn;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    b_r[e] = D.bih[u0 + e] + D.bhh[u0 + e];
    b_z[e] = D.bih[64 + u0 + e] + D.bhh[64 + u0 + e];
    b_in[e] = D.bih[128 + u0 + e];
    b_hn[e] = D.bhh[128 + u0 + e];
  }
  for (int i = tid; i < 2 * 3 * 16 * HSB; i += 256) (&hb[0][0][0][0])[i] = (__bf16)0.0f;

  // ---- staging of the x tile: thread -> NXP float4 pieces (row, c4); running pointers; a step ahead ----
  constexpr int C4 = I / 4;
  const int64_t xstep = (int64_t)D.t_sign * a.x_ts, hstep = (int64_t)D.t_sign * D.h_ts;
  // The x pieces are prefetched PD = 3 steps ahead into a ring of register slots (slot = step mod 3, the step loop is
  // unrolled by three so every slot index is static): under this kernel's store traffic a one-step prefetch left more
  // than a step of load latency exposed in the staging phase.
  constexpr int PD = 3;
  const float* xq[NXP]; uint32_t xqe_ring[PD][NXP]; int xrow[NXP], xcol[NXP]; bool xlive[NXP]; float4 xv[PD][NXP];
  uint32_t xqe[NXP];
  int loaded = 0;                                          // step whose x the pointers address next
  auto issue_x = [&](auto slot_tag) {                      // load the x pieces of step `loaded` into ring slot, move on (clamped)
    constexpr int SL = decltype(slot_tag)::value;
#pragma unroll
    for (int j = 0; j < NXP; ++j) { xv[SL][j] = *(const float4*)xq[j]; xqe_ring[SL][j] = xqe[j]; }
    if (loaded + 1 < n_steps) {
#pragma unroll
      for (int j = 0; j < NXP; ++j) { xq[j] += xstep; xqe[j] += (uint32_t)xstep; }
    }
    ++loaded;
  };
#pragma unroll
  for (int j = 0; j < NXP; ++j) {
    const int idx = tid + 256 * j;
    xlive[j] = idx < 16 * C4;
    const int ic = xlive[j] ? idx : 0;
    xrow[j] = ic / C4; xcol[j] = 4 * (ic - xrow[j] * C4);
    const int br = min(tile * 16 + xrow[j], a.B - 1);
    const int64_t e0 = (int64_t)br * a.x_bs + (int64_t)D.t_start * a.x_ts + xcol[j];
    xq[j] = a.x + e0; xqe[j] = (uint32_t)e0;
  }
  auto stage_x = [&](auto slot_tag, int buf) {      // (mask,) split and store the pieces of ring slot into xb[buf]
    constexpr int SL = decltype(slot_tag)::value;
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
      float q[4] = {xv[SL][j].x, xv[SL][j].y, xv[SL][j].z, xv[SL][j].w};
      if constexpr (DROP) {
        const uint32_t wd = drop_word(xqe_ring[SL][j], a.drop_key);
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] *= drop_mul(wd, e, a.drop_thr, a.drop_scale);
      }
      bf16x4 p[3];
      split3_quad(q, p);
      if (xlive[j]) {
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) *(bf16x4*)&xb[buf][pp][xrow[j]][xcol[j]] = p[pp];
      }
    }
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;
  issue_x(S0{});                                           // x of step 0 -> slot 0 ...
  stage_x(S0{}, 0);                                        // ... and straight into its LDS buffer
  issue_x(S1{}); issue_x(S2{}); issue_x(S0{});             // steps 1, 2, 3 (clamped reloads beyond the sequence) -> slots 1, 2, 0
  __syncthreads();

  float* hptr = D.h + (int64_t)bl * D.h_bs + (int64_t)D.t_start * D.h_ts + D.h_col + u0;
  float4* sp = STASH ? D.stash + ((size_t)((size_t)tile * n_steps) * 4 + w) * 4 * 64 + lane : nullptr;
  f32x4 hprev = {0.f, 0.f, 0.f, 0.f};
  STAMP_DECL;
  auto body = [&](int s, auto next_slot) {       // next_slot = (s + 1) mod 3: holds x of step s+1, refilled with step s+4
    const int cur = s & 1;
    STAMP(0);
    // x of step s+1 (loaded three iterations ago) into the other buffer — every wave is past the barrier that followed
    // its last read of it — and the loads for step s+4 into the slot just freed
    stage_x(next_slot, cur ^ 1);
    issue_x(next_slot);
    STAMP(1);
    // input projection: x planes of this step
    f32x4 acc_r = b_r, acc_z = b_z, acc_in = b_in, acc_hn = b_hn;
#pragma unroll
    for (int kb = 0; kb < NKX; ++kb) {
      bf16x8 xo[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) xo[p] = *(const bf16x8*)&xb[cur][p][li][kb * 32 + lq * 8];
      acc_r = mfma_bf16x3<CT_FWD_PROJ>(Ai[0][kb], xo, acc_r);
      acc_z = mfma_bf16x3<CT_FWD_PROJ>(Ai[1][kb], xo, acc_z);
      acc_in = mfma_bf16x3<CT_FWD_PROJ>(Ai[2][kb], xo, acc_in);
    }
    STAMP(2);
    lds_barrier();   // h_{s-1} of every wave is in hb[cur]; x of step s+1 is complete in xb[cur^1]; all reads of xb[cur] are done
    STAMP(3);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 ho[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) ho[p] = *(const bf16x8*)&hb[cur][p][li][kb * 32 + lq * 8];
      acc_r = mfma_bf16x3<CT_FWD_REC>(Ah[0][kb], ho, acc_r);
      acc_z = mfma_bf16x3<CT_FWD_REC>(Ah[1][kb], ho, acc_z);
      acc_hn = mfma_bf16x3<CT_FWD_REC>(Ah[2][kb], ho, acc_hn);
    }
    STAMP(4);
    f32x4 r, z, n, hn;
    gru_gates(acc_r, acc_z, acc_in, acc_hn, hprev, r, z, n, hn);
    hprev = hn;
    {
      bf16x4 hp[3];
      split3_quad(hn, hp);
#pragma unroll
      for (int p = 0; p < 3; ++p) *(bf16x4*)&hb[cur ^ 1][p][li][u0] = hp[p];
    }
    STAMP(5);
    *(float4*)hptr = make_float4(hn[0], hn[1], hn[2], hn[3]);
    hptr += hstep;
    if constexpr (STASH) {
      sp[0 * 64] = make_float4(r[0], r[1], r[2], r[3]);
      sp[1 * 64] = make_float4(z[0], z[1], z[2], z[3]);
      sp[3 * 64] = make_float4(acc_hn[0], acc_hn[1], acc_hn[2], acc_hn[3]);
      sp += 4 * 4 * 64;
    }
    STAMP(6);
  };
  for (int s = 0; s < n_steps; s += 3) {
    body(s, S1{});
    if (s + 1 < n_steps) body(s + 1, S2{});
    if (s + 2 < n_steps) body(s + 2, S0{});
  }
#ifdef MSIG_STAMPS
  if (a.dbg && tid == 0 && blockIdx.y == 0 && blockIdx.x < 256)
    for (int i = 0; i < 8; ++i) a.dbg[(size_t)blockIdx.x * 8 + i] = ph_[i];
#endif
  if (D.h_last != nullptr && valid)
    *(float4*)(D.h_last + (int64_t)b * D.hl_bs + D.hl_col + u0) = make_float4(hprev[0], hprev[1], hprev[2], hprev[3]);
}

// ------------------------------------------------------------------------------------
// Wave-specialised throughput forward (gru_fwd_ws): 512 threads = TWO waves per SIMD with different jobs.
// One wave per SIMD cannot hide anything: its VALU, LDS and memory instructions add to its MFMA time (tools/mfma_coissue*.hip:
// a filler beside v_mfma_f32_16x16x32_bf16 costs the issuing wave ~3.3 of its 4 cycles), so gru_fwd_b3 keeps the matrix pipe
// 27-41 % busy.  But ANOTHER wave of the same SIMD does run beside it (tools/wave_roles.hip, r02_wave_roles_microbench.log:
// an MFMA-only wave keeps its 16.1 cycles per MFMA while its SIMD partner retires 2.6 v_fma_f32 or 1.3 transcendentals per
// MFMA).  So the step is split by dependency, not by units:
//   waves 0-3  "chain": the recurrent part that depends on h_{t-1} — 36 MFMAs (W_hh), the gate math, the state split /
//                       exchange through LDS, the h / stash stores;
//   waves 4-7  "bulk" : everything that does not — staging (mask, split) of the input tile and the input projection
//                       W_ih x_t + b (72 MFMAs for I = 128), ONE STEP AHEAD, handed over as the chain's accumulator
//                       initial values through a lane-linear LDS ring (3 x 1 KiB per wave and step, conflict-free b128).
// Wave w and wave w+4 own the same 16 units of every gate and sit on the same SIMD (waves are dealt to SIMDs cyclically).
// One workgroup barrier per step: interval k = chain step k | projection of step k+1, staging of step k+2.
// ------------------------------------------------------------------------------------
// The stash is written once and read once, a whole backward pass later: MSIG_STASH_NT=1 stores it non-temporally.
#ifndef MSIG_STASH_NT
#define MSIG_STASH_NT 1
#endif
#if MSIG_STASH_NT
#define STASH_STORE(p, v) __builtin_nontemporal_store((v), (f32x4*)(p))
#else
#define STASH_STORE(p, v) (*(float4*)(p) = make_float4((v)[0], (v)[1], (v)[2], (v)[3]))
#endif
template <int I, bool STASH, bool FOLDS>
__global__ __launch_bounds__(512, 2) void gru_fwd_ws(const GruArgs a, const FoldCtx fc) {
  constexpr int NKX = I / 32;                           // 32-wide k blocks of the input
  constexpr bool DROP = (I == 128);
  // plane row strides in bf16 elements.  Layer 1: 16 * odd dwords (48 / 80) + the quad swizzle of msig_dev.h — round 2's 36- and
  // 68-dword rows cost 36 % of this kernel's LDS cycles in bank conflicts (profiles/r02_pmc_lds_B8192.csv); conflict-free rows
  // take layer 1 from 0.80 to 0.75 ms.  Layer 0 keeps the padded rows: it is bound by the bytes it writes (4.5 GB per launch),
  // and with the swizzled planes it measured SLOWER (1.04 -> 1.23 ms, profiles/r03_fwd_ws_swizzle.log) — its chain waves then
  // issue their store bursts closer together.
  constexpr bool SWZ = I == 128;
  constexpr int HSB = SWZ ? 96 : 72, XSB = SWZ ? 160 : I + 8;
  constexpr int NXP = (16 * I / 4 + 255) / 256;         // float4 pieces of the x tile per bulk thread (1 or 2)
  constexpr int C4 = I / 4;
  // Round 5: the recurrence W_hh h (both layers) and the layer-1 projection W_ih x (x = dropped layer-0 output: |x| <= the dropout
  // scale) run on two-piece fp16 (msig_dev.h f16x2: three MFMAs per block instead of six, two planes instead of three); the
  // layer-0 projection stays on split-bf16 (its input is a BatchNorm output: no bound that fp16's range could rely on).
  constexpr bool XF16 = I == 128;
  constexpr int NPX = XF16 ? 2 : 3;                     // piece planes of the x tile
  __shared__ __attribute__((aligned(16))) _Float16 hb[2][2][16][HSB];
  __shared__ __attribute__((aligned(16))) unsigned short xb[2][NPX][16][XSB];       // fp16 (layer 1) or bf16 (layer 0) pieces
  __shared__ __attribute__((aligned(16))) float4 gi[2][4][3][64];      // [slot][unit block][gate r,z,n][lane]
  // h_t leaves through a 16 x 64 fp32 tile so that every store instruction writes whole 256-byte rows (16 lanes x float4).
  // In the MFMA layout a wave holds 64 bytes of each of 16 rows: stored directly, those half-line pieces cost the layer-0
  // kernel 0.48 ms of its 1.38 ms although h is a fifth of its bytes (measured by leaving the store out).
  constexpr int HFS = 68;
  __shared__ __attribute__((aligned(16))) float hf[2][16][HFS];
  FOLD_GRU_ARGS_IF(FOLDS);
  const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6, w = w8 & 3, li = lane & 15, lq = lane >> 4;
  const bool bulk = w8 >= 4;                            // wave-uniform
  const int tile = blockIdx.x, b = tile * 16 + li;
  const bool valid = b < a.B;
  // rows >= B replay row B-1 bit for bit (clamped loads), so their stores may land on row B-1's addresses
  const int u0 = w * 16 + lq * 4;
  const int sw_li = SWZ ? quad_swz(li) : 0;
  const int n_steps = D.n_steps;
  const int64_t xstep = (int64_t)D.t_sign * a.x_ts, hstep = (int64_t)D.t_sign * D.h_ts;

  if (bulk) {
    // ================= bulk waves: x staging + input projection, one step ahead =================
    const int tb = tid - 256;
    bf16x8 Ai[XF16 ? 1 : 3][XF16 ? 1 : NKX][3];           // layer 0: split-bf16 pieces
    f16x8 Af[XF16 ? 3 : 1][XF16 ? NKX : 1][2];            // layer 1: f16x2 pieces, scaled per gate fragment (f16x2_weight_scale)
    [[maybe_unused]] float posti[3] = {1.f, 1.f, 1.f};    // 1 / (S_w S_x)
    [[maybe_unused]] float sx = F16X2_H_SCALE;            // |x| <= dropout scale: S_x = 2^12 / 2^ceil(log2(scale))
    if constexpr (XF16) {
      for (float ds = a.drop_scale; ds > 1.0f; ds *= 0.5f) sx *= 0.5f;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        float wv[NKX][8], m = 0.f;                          // loaded ONCE: the scale needs the fragment's maximum before the split
#pragma unroll
        for (int kb = 0; kb < NKX; ++kb) {
          const float4* wi = (const float4*)(D.Wih + (size_t)(g * 64 + w * 16 + li) * I + kb * 32 + lq * 8);
          const float4 a0 = wi[0], a1 = wi[1];
          wv[kb][0] = a0.x; wv[kb][1] = a0.y; wv[kb][2] = a0.z; wv[kb][3] = a0.w; wv[kb][4] = a1.x; wv[kb][5] = a1.y; wv[kb][6] = a1.z; wv[kb][7] = a1.w;
#pragma unroll
          for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(wv[kb][j]));
        }
        const float sw = f16x2_weight_scale(m);
        posti[g] = 1.0f / (sw * sx);
#pragma unroll
        for (int kb = 0; kb < NKX; ++kb)
#pragma unroll
          for (int j = 0; j < 8; ++j) { _Float16 p0, p1; split2(wv[kb][j], sw, p0, p1); Af[g][kb][0][j] = p0; Af[g][kb][1][j] = p1; }
      }
    } else {
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kb = 0; kb < NKX; ++kb) {
          const float* wi = D.Wih + (size_t)(g * 64 + w * 16 + li) * I + kb * 32 + lq * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) { __bf16 p0, p1, p2; split3(wi[j], p0, p1, p2); Ai[g][kb][0][j] = p0; Ai[g][kb][1][j] = p1; Ai[g][kb][2][j] = p2; }
        }
    }
    f32x4 b_r, b_z, b_in;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      b_r[e] = D.bih[u0 + e] + D.bhh[u0 + e];
      b_z[e] = D.bih[64 + u0 + e] + D.bhh[64 + u0 + e];
      b_in[e] = D.bih[128 + u0 + e];
    }
    // x pieces prefetched PD = 3 steps ahead into a ring of register slots (as gru_fwd_b3)
    constexpr int PD = 3;
    const float* xq[NXP]; uint32_t xqe_ring[PD][NXP]; int xrow[NXP], xcol[NXP]; bool xlive[NXP]; float4 xv[PD][NXP];
    uint32_t xqe[NXP];
    int loaded = 0;
    auto issue_x = [&](auto slot_tag) {
      constexpr int SL = decltype(slot_tag)::value;
#pragma unroll
      for (int j = 0; j < NXP; ++j) { xv[SL][j] = *(const float4*)xq[j]; xqe_ring[SL][j] = xqe[j]; }
      if (loaded + 1 < n_steps) {
#pragma unroll
        for (int j = 0; j < NXP; ++j) { xq[j] += xstep; xqe[j] += (uint32_t)xstep; }
      }
      ++loaded;
    };
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
      const int idx = tb + 256 * j;
      xlive[j] = idx < 16 * C4;
      const int ic = xlive[j] ? idx : 0;
      xrow[j] = ic / C4; xcol[j] = 4 * (ic - xrow[j] * C4);
      const int br = min(tile * 16 + xrow[j], a.B - 1);
      const int64_t e0 = (int64_t)br * a.x_bs + (int64_t)D.t_start * a.x_ts + xcol[j];
      xq[j] = ax_ + e0; xqe[j] = (uint32_t)e0;
    }
    auto stage_x = [&](auto slot_tag, int buf) {      // (mask,) split and store the pieces of ring slot into xb[buf]
      constexpr int SL = decltype(slot_tag)::value;
#pragma unroll
      for (int j = 0; j < NXP; ++j) {
        float q[4] = {xv[SL][j].x, xv[SL][j].y, xv[SL][j].z, xv[SL][j].w};
        if constexpr (DROP) {
          const uint32_t wd = drop_word(xqe_ring[SL][j], akey_);
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] *= drop_mul(wd, e, a.drop_thr, a.drop_scale);
        }
        if constexpr (XF16) {
          f16x4 p[2];
          split2_quad((f32x4){q[0], q[1], q[2], q[3]}, sx, p[0], p[1]);
          if (xlive[j]) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) *(f16x4*)&xb[buf][pp][xrow[j]][xcol[j] ^ (SWZ ? quad_swz(xrow[j]) : 0)] = p[pp];
          }
        } else {
          bf16x4 p[3];
          split3_quad(q, p);
          if (xlive[j]) {
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) *(bf16x4*)&xb[buf][pp][xrow[j]][xcol[j] ^ (SWZ ? quad_swz(xrow[j]) : 0)] = p[pp];
          }
        }
      }
    };
    auto project = [&](int step) {                     // gi[step & 1] = W_ih x_step + b from xb[step & 1]
      const int sl = step & 1;
      f32x4 acc_r, acc_z, acc_in;
      if constexpr (XF16) {
        acc_r = acc_z = acc_in = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKX; ++kb) {
          f16x8 xo[2];
#pragma unroll
          for (int p = 0; p < 2; ++p) xo[p] = *(const f16x8*)&xb[sl][p][li][kb * 32 + ((lq * 8) ^ sw_li)];
          acc_r = mfma_f16x2<CT_FWD_PROJ>(Af[0][kb], xo, acc_r);
          acc_z = mfma_f16x2<CT_FWD_PROJ>(Af[1][kb], xo, acc_z);
          acc_in = mfma_f16x2<CT_FWD_PROJ>(Af[2][kb], xo, acc_in);
        }
        acc_r = b_r + acc_r * posti[0]; acc_z = b_z + acc_z * posti[1]; acc_in = b_in + acc_in * posti[2];
      } else {
        acc_r = b_r; acc_z = b_z; acc_in = b_in;
#pragma unroll
        for (int kb = 0; kb < NKX; ++kb) {
          bf16x8 xo[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) xo[p] = *(const bf16x8*)&xb[sl][p][li][kb * 32 + ((lq * 8) ^ sw_li)];
          acc_r = mfma_bf16x3<CT_FWD_PROJ>(Ai[0][kb], xo, acc_r);
          acc_z = mfma_bf16x3<CT_FWD_PROJ>(Ai[1][kb], xo, acc_z);
          acc_in = mfma_bf16x3<CT_FWD_PROJ>(Ai[2][kb], xo, acc_in);
        }
      }
      gi[sl][w][0][lane] = make_float4(acc_r[0], acc_r[1], acc_r[2], acc_r[3]);
      gi[sl][w][1][lane] = make_float4(acc_z[0], acc_z[1], acc_z[2], acc_z[3]);
      gi[sl][w][2][lane] = make_float4(acc_in[0], acc_in[1], acc_in[2], acc_in[3]);
    };
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>; using S2 = std::integral_constant<int, 2>;
    issue_x(S0{});                                           // x of step 0 -> slot 0 ...
    stage_x(S0{}, 0);                                        // ... and straight into xb[0]
    issue_x(S1{}); issue_x(S2{}); issue_x(S0{});             // steps 1, 2, 3 (clamped reloads beyond the sequence) -> slots 1, 2, 0
    lds_barrier();                                           // barrier P1: xb[0] complete
    project(0);                                              // gi[0]
    stage_x(S1{}, 1);                                        // x of step 1 -> xb[1]
    issue_x(S1{});                                           // step 4 -> slot 1
    lds_barrier();                                           // barrier P2: gi[0], xb[1] (and the chain's zeroed state) visible
    // interval k: projection of step k+1 (from xb[(k+1)&1]) -> gi[(k+1)&1]; staging of step k+2 -> xb[k&1]; prefetch of step k+5
    auto body = [&](int k, auto slot_tag) {                  // slot_tag = (k + 2) mod 3: holds x of step k+2
      if (k + 1 < n_steps) project(k + 1);
      stage_x(slot_tag, k & 1);
      issue_x(slot_tag);
      lds_barrier();
    };
    for (int k = 0; k < n_steps; k += 3) {
      body(k, S2{});
      if (k + 1 < n_steps) body(k + 1, S0{});
      if (k + 2 < n_steps) body(k + 2, S1{});
    }
    return;
  }
  // ================= chain waves: recurrent part, gates, state exchange, stores =================
  // The chain is the step's critical path and shares its SIMD's matrix pipe with a bulk wave that works one step ahead and has
  // twice the MFMAs: raised priority lets the chain's instructions win the arbitration (layer 0 1.05 -> 1.02 ms, layer 1 -1 %).
  // Measured and left: deferring the chain's stores by a step and threading them through the next step's MFMA groups, as the
  // latency-form recurrence does, made layer 1 2 % slower (0.80 vs 0.78 ms) and costs layer 0 its second workgroup per CU.
  __builtin_amdgcn_s_setprio(3);
  f16x8 Ah[3][2][2];
  float post[3];                                             // 1 / (S_w S_h) of each gate's fragment
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    float wv[2][8], m = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const float* wr = D.Whh + (size_t)(g * 64 + w * 16 + li) * 64 + kb * 32 + lq * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) { wv[kb][j] = wr[j]; m = fmaxf(m, fabsf(wv[kb][j])); }
    }
    const float sw = f16x2_weight_scale(m);
    post[g] = 1.0f / (sw * F16X2_H_SCALE);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < 8; ++j) { _Float16 p0, p1; split2(wv[kb][j], sw, p0, p1); Ah[g][kb][0][j] = p0; Ah[g][kb][1][j] = p1; }
  }
  const f32x4 b_hn = {D.bhh[128 + u0], D.bhh[128 + u0 + 1], D.bhh[128 + u0 + 2], D.bhh[128 + u0 + 3]};
  for (int i = tid; i < 2 * 2 * 16 * HSB; i += 256) (&hb[0][0][0][0])[i] = (_Float16)0.0f;       // tid < 256 here
  const int hrow = tid >> 4, hc4 = (tid & 15) * 4;                         // store role: row hrow, columns hc4 .. hc4+3 of the tile
  float* hptr = D.h + (int64_t)min(tile * 16 + hrow, a.B - 1) * D.h_bs + (int64_t)D.t_start * D.h_ts + D.h_col + hc4;
  float4* sp = STASH ? D.stash + ((size_t)((size_t)tile * n_steps) * 4 + w) * 4 * 64 + lane : nullptr;
  f32x4 hprev = {0.f, 0.f, 0.f, 0.f};
  const bool skip_hn = a.stash_skip_hn != 0;                 // gru_bwd_b6 recomputes W_hn h + b_hn from h (wave-uniform)
  lds_barrier();                                             // barrier P1
  lds_barrier();                                             // barrier P2
  for (int k = 0; k < n_steps; ++k) {
    const int cur = k & 1;
    if (k > 0) {                                             // h of step k-1, complete in hf[cur ^ 1] since the last barrier
      *(float4*)hptr = *(const float4*)&hf[cur ^ 1][hrow][hc4];
      hptr += hstep;
    }
    const float4 g_r = gi[cur][w][0][lane], g_z = gi[cur][w][1][lane], g_n = gi[cur][w][2][lane];
    f16x8 ho[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int p = 0; p < 2; ++p) ho[kb][p] = *(const f16x8*)&hb[cur][p][li][kb * 32 + ((lq * 8) ^ sw_li)];
    const f32x4 acc_in = {g_n.x, g_n.y, g_n.z, g_n.w};
    f32x4 acc_r = {0.f, 0.f, 0.f, 0.f}, acc_z = acc_r, acc_hn = acc_r;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      acc_r = mfma_f16x2<CT_FWD_REC>(Ah[0][kb], ho[kb], acc_r);
      acc_z = mfma_f16x2<CT_FWD_REC>(Ah[1][kb], ho[kb], acc_z);
      acc_hn = mfma_f16x2<CT_FWD_REC>(Ah[2][kb], ho[kb], acc_hn);
    }
    acc_r = (f32x4){g_r.x, g_r.y, g_r.z, g_r.w} + acc_r * post[0];
    acc_z = (f32x4){g_z.x, g_z.y, g_z.z, g_z.w} + acc_z * post[1];
    acc_hn = b_hn + acc_hn * post[2];
    f32x4 r, z, n, hn;
    gru_gates(acc_r, acc_z, acc_in, acc_hn, hprev, r, z, n, hn);
    hprev = hn;
    {
      f16x4 hp[2];
      split2_quad(hn, F16X2_H_SCALE, hp[0], hp[1]);
#pragma unroll
      for (int p = 0; p < 2; ++p) *(f16x4*)&hb[cur ^ 1][p][li][u0 ^ sw_li] = hp[p];
    }
    *(float4*)&hf[cur][li][u0] = make_float4(hn[0], hn[1], hn[2], hn[3]);
    if constexpr (STASH) {
      STASH_STORE(&sp[0 * 64], r);
      STASH_STORE(&sp[1 * 64], z);
      if (!skip_hn) STASH_STORE(&sp[3 * 64], acc_hn);
      sp += 4 * 4 * 64;
    }
    lds_barrier();
  }
  *(float4*)hptr = *(const float4*)&hf[(n_steps - 1) & 1][hrow][hc4];      // the last step's h
  if (D.h_last != nullptr && valid)
    *(float4*)(D.h_last + (int64_t)b * D.hl_bs + D.hl_col + u0) = make_float4(hprev[0], hprev[1], hprev[2], hprev[3]);
}

// ------------------------------------------------------------------------------------
// Bulk input projection for the latency form: gi[unit] = W_ih x_t + b  (b_r = b_ir + b_hr,
// b_z = b_iz + b_hz, b_n = b_in) for every (tile, step) unit of direction 0, in the D layout the
// recurrence consumes.  No LDS, no barriers: units are independent and spread over all CUs.
// ------------------------------------------------------------------------------------
template <int I>
__global__ __launch_bounds__(256, 2) void gru_fwd_proj(const GruArgs a, int n_tiles, const FoldCtx fc) {
  // Round 3: on split-bf16 MFMA like every other GRU contraction (it ran 3 x I/4 v_mfma_f32_16x16x4_f32 per unit: 3072 matrix
  // cycles for layer 1 against 1152 now).  The B operand needs no LDS: a lane's eight consecutive k of a 32-wide k block are two
  // float4 of its own row of x, masked and split in registers.  Same operands, split and term order as gru_fwd_ws's projection.
  constexpr int NKB = I / 32;
  constexpr bool DROP = (I == 128);
  FOLD_GRU_ARGS;
  float4* gi = agi_ + (size_t)blockIdx.y * a.gi_dir_stride;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int u0 = w * 16 + lq * 4;
  constexpr bool XF16 = I == 128;                        // layer 1 on f16x2, layer 0 on split-bf16: as gru_fwd_ws (msig_dev.h)
  bf16x8 Ai[XF16 ? 1 : 3][XF16 ? 1 : NKB][3];
  f16x8 Af[XF16 ? 3 : 1][XF16 ? NKB : 1][2];
  [[maybe_unused]] float posti[3] = {1.f, 1.f, 1.f};
  [[maybe_unused]] float sx = F16X2_H_SCALE;
  if constexpr (XF16) {
    for (float ds = a.drop_scale; ds > 1.0f; ds *= 0.5f) sx *= 0.5f;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      float wv[NKB][8], m = 0.f;                            // loaded ONCE (two 16-byte loads per k block): the scale needs the maximum first
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const float4* wi = (const float4*)(D.Wih + (size_t)(g * 64 + w * 16 + li) * I + kb * 32 + lq * 8);
        const float4 a0 = wi[0], a1 = wi[1];
        wv[kb][0] = a0.x; wv[kb][1] = a0.y; wv[kb][2] = a0.z; wv[kb][3] = a0.w; wv[kb][4] = a1.x; wv[kb][5] = a1.y; wv[kb][6] = a1.z; wv[kb][7] = a1.w;
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(wv[kb][j]));
      }
      const float sw = f16x2_weight_scale(m);
      posti[g] = 1.0f / (sw * sx);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int j = 0; j < 8; ++j) { _Float16 p0, p1; split2(wv[kb][j], sw, p0, p1); Af[g][kb][0][j] = p0; Af[g][kb][1][j] = p1; }
    }
  } else {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const float* wi = D.Wih + (size_t)(g * 64 + w * 16 + li) * I + kb * 32 + lq * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { __bf16 p0, p1, p2; split3(wi[j], p0, p1, p2); Ai[g][kb][0][j] = p0; Ai[g][kb][1][j] = p1; Ai[g][kb][2][j] = p2; }
      }
  }
  f32x4 b_r, b_z, b_n;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    b_r[e] = D.bih[u0 + e] + D.bhh[u0 + e];
    b_z[e] = D.bih[64 + u0 + e] + D.bhh[64 + u0 + e];
    b_n[e] = D.bih[128 + u0 + e];
  }
  const int n_steps = D.n_steps, n_units = n_tiles * n_steps;
  // the x rows of the NEXT unit are loaded while this unit is split and contracted (round 5: a unit's loads used to sit in front of
  // their first use, a full memory latency per unit: layer 1 24 -> 23 us per launch for one model, 0.17 -> 0.14 ms in a 15-fold batch)
  auto x_index = [&](int unit) -> int64_t {
    const int tile = unit / n_steps, s = unit - tile * n_steps, t = D.t_start + D.t_sign * s;
    const int b = tile * 16 + li, bl = b < a.B ? b : a.B - 1;
    return (int64_t)bl * a.x_bs + (int64_t)t * a.x_ts + lq * 8;
  };
  float4 qn[NKB][2];
  if ((int)blockIdx.x < n_units) {
    const int64_t e1 = x_index(blockIdx.x);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) { qn[kb][0] = *(const float4*)(ax_ + e1 + kb * 32); qn[kb][1] = *(const float4*)(ax_ + e1 + kb * 32 + 4); }
  }
  for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
    const int64_t e0 = x_index(unit);
    float4 q[NKB][2];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) { q[kb][0] = qn[kb][0]; q[kb][1] = qn[kb][1]; }
    if (unit + (int)gridDim.x < n_units) {
      const int64_t e1 = x_index(unit + gridDim.x);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) { qn[kb][0] = *(const float4*)(ax_ + e1 + kb * 32); qn[kb][1] = *(const float4*)(ax_ + e1 + kb * 32 + 4); }
    }
    f32x4 acc_r, acc_z, acc_n;
    if constexpr (XF16) acc_r = acc_z = acc_n = (f32x4){0.f, 0.f, 0.f, 0.f};
    else { acc_r = b_r; acc_z = b_z; acc_n = b_n; }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      float v[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v[h][0] = q[kb][h].x; v[h][1] = q[kb][h].y; v[h][2] = q[kb][h].z; v[h][3] = q[kb][h].w;
        if constexpr (DROP) {
          const uint32_t wd = drop_word((uint32_t)(e0 + kb * 32 + 4 * h), akey_);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[h][e] *= drop_mul(wd, e, a.drop_thr, a.drop_scale);
        }
      }
      if constexpr (XF16) {
        f16x4 lo[2], hi[2];
        split2_quad((f32x4){v[0][0], v[0][1], v[0][2], v[0][3]}, sx, lo[0], lo[1]);
        split2_quad((f32x4){v[1][0], v[1][1], v[1][2], v[1][3]}, sx, hi[0], hi[1]);
        f16x8 xo[2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
          for (int e = 0; e < 4; ++e) { xo[pp][e] = lo[pp][e]; xo[pp][4 + e] = hi[pp][e]; }
        acc_r = mfma_f16x2<CT_FWD_PROJ>(Af[0][kb], xo, acc_r);
        acc_z = mfma_f16x2<CT_FWD_PROJ>(Af[1][kb], xo, acc_z);
        acc_n = mfma_f16x2<CT_FWD_PROJ>(Af[2][kb], xo, acc_n);
      } else {
        bf16x4 lo[3], hi[3];
        split3_quad(v[0], lo);
        split3_quad(v[1], hi);
        bf16x8 xo[3];
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
          for (int e = 0; e < 4; ++e) { xo[pp][e] = lo[pp][e]; xo[pp][4 + e] = hi[pp][e]; }
        acc_r = mfma_bf16x3<CT_FWD_PROJ>(Ai[0][kb], xo, acc_r);
        acc_z = mfma_bf16x3<CT_FWD_PROJ>(Ai[1][kb], xo, acc_z);
        acc_n = mfma_bf16x3<CT_FWD_PROJ>(Ai[2][kb], xo, acc_n);
      }
    }
    if constexpr (XF16) { acc_r = b_r + acc_r * posti[0]; acc_z = b_z + acc_z * posti[1]; acc_n = b_n + acc_n * posti[2]; }
    float4* gp = gi + ((size_t)unit * 4 + w) * 3 * 64 + lane;
    gp[0] = make_float4(acc_r[0], acc_r[1], acc_r[2], acc_r[3]);
    gp[64] = make_float4(acc_z[0], acc_z[1], acc_z[2], acc_z[3]);
    gp[128] = make_float4(acc_n[0], acc_n[1], acc_n[2], acc_n[3]);
  }
}

// ------------------------------------------------------------------------------------
// Forward recurrence of the latency form (few batch tiles: one workgroup per CU and nothing to overlap
// with but itself).  Input projections come from gru_fwd_proj, so a step is the 48 recurrent MFMAs per
// wave (1536 cycles) + the gate math + the LDS exchange of h.  At one wave per SIMD the VALU does NOT run
// in an MFMA's shadow (tools/mfma_coissue.hip: 32 cycles per MFMA alone, +4 per interleaved VALU op, +9 per
// transcendental; dependent MFMAs cost nothing extra), so the step cannot go below MFMA + gates; what can
// be taken off the critical path is the memory traffic: the stores of step s-1 (h and the four stash
// vectors) and the projection prefetch (round 4: for step s+2) are issued right after the barrier of step s, in
// front of its MFMAs, instead of between the gate math and the barrier where every wave waits for them.
// ------------------------------------------------------------------------------------
template <bool STASH>
__global__ __launch_bounds__(256, 2) void gru_fwd_rec(const GruArgs a, const FoldCtx fc) {
  // h_{s-1} crosses lanes as TWO fp16 planes (the pieces of the f16x2 contraction, msig_dev.h; three bf16 planes until round 4):
  // the producer splits its four fresh values once, every consumer reads ready-made B operands (16 bytes per piece and k block)
  constexpr int HSB = 72;                               // row stride in fp16 elements (144 B: 16-byte aligned rows)
  __shared__ __attribute__((aligned(16))) _Float16 hb[2][2][16][HSB];
  FOLD_GRU_ARGS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int tile = blockIdx.x, b = tile * 16 + li;
  const bool valid = b < a.B;
  const int bl = valid ? b : a.B - 1;      // rows >= B replay the last row bit for bit (stores hit the same address)
  const int u0 = w * 16 + lq * 4;
  // A operands: W_hh rows of this wave's 16 units per gate, k block kb, split once for the whole sequence; the scale of a gate's
  // fragment comes from its largest magnitude in this wave (f16x2_weight_scale), post[g] = 1 / (S_w S_h) undoes both scales
  f16x8 Aw[3][2][2];
  float post[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    float wv[2][8], m = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const float* wr = D.Whh + (size_t)(g * 64 + w * 16 + li) * 64 + kb * 32 + lq * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) { wv[kb][j] = wr[j]; m = fmaxf(m, fabsf(wv[kb][j])); }
    }
    const float sw = f16x2_weight_scale(m);
    post[g] = 1.0f / (sw * F16X2_H_SCALE);                // a power of two: exact
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < 8; ++j) { _Float16 p0, p1; split2(wv[kb][j], sw, p0, p1); Aw[g][kb][0][j] = p0; Aw[g][kb][1][j] = p1; }
  }
  const f32x4 bhn = {D.bhh[128 + u0], D.bhh[128 + u0 + 1], D.bhh[128 + u0 + 2], D.bhh[128 + u0 + 3]};
  for (int i = tid; i < 2 * 2 * 16 * HSB; i += 256) (&hb[0][0][0][0])[i] = (_Float16)0.0f;
  __syncthreads();

  const int n_steps = D.n_steps;
  const int64_t hstep = (int64_t)D.t_sign * D.h_ts;
  float* hptr = D.h + (int64_t)bl * D.h_bs + (int64_t)D.t_start * D.h_ts + D.h_col + u0;
  float4* sp = STASH ? D.stash + ((size_t)((size_t)tile * n_steps) * 4 + w) * 4 * 64 + lane : nullptr;
  // The projections (three float4 per lane and step) are prefetched THREE steps ahead into three register sets that the unrolled
  // loop rotates (round 4: two; the two-piece fp16 recurrence of round 5 shortened a step from ~1700 to ~1300 cycles, and with it
  // the distance in time).  Per launch, one / five / fifteen folds: depth 2: 147 / 201 / 287 us, depth 3: 146 / 185 / 275,
  // depth 4: 145 / 192 / 273 (layer 0; profiles/r05_fwd_rec_prefetch_depth.log).  What is left of the stretch in a fold batch is
  // not the prefetch distance.
  struct G3 { float4 r, z, n; };
  const float4* gq = agi_ + (size_t)blockIdx.y * a.gi_dir_stride + ((size_t)((size_t)tile * n_steps) * 4 + w) * 3 * 64 + lane;
#ifndef MSIG_REC_PREFETCH
#define MSIG_REC_PREFETCH 3
#endif
  constexpr int PD = MSIG_REC_PREFETCH;                   // prefetch distance in steps = register sets
  G3 g[PD];
  g[0] = G3{gq[0], gq[64], gq[128]};                      // step 0
#pragma unroll
  for (int i = 1; i < PD; ++i) {                          // steps 1 .. PD-1; gq points at the last step loaded: min(i, n_steps - 1)
    if (i < n_steps) gq += 4 * 3 * 64;                    // (a shorter sequence re-reads its last step, harmlessly)
    g[i] = G3{gq[0], gq[64], gq[128]};
  }
  f32x4 hprev = {0.f, 0.f, 0.f, 0.f};
  f32x4 sv_r = hprev, sv_z = hprev, sv_a = hprev;      // stash of the previous step (r, z, W_hn h + b_hn), stored one step late
  int cur = 0;
  STAMP_DECL;
  auto flush = [&]() {        // global stores of the step whose results are in (hprev, sv_*)
    *(float4*)hptr = make_float4(hprev[0], hprev[1], hprev[2], hprev[3]);
    hptr += hstep;
    if constexpr (STASH) {
      sp[0 * 64] = make_float4(sv_r[0], sv_r[1], sv_r[2], sv_r[3]);
      sp[1 * 64] = make_float4(sv_z[0], sv_z[1], sv_z[2], sv_z[3]);
      sp[3 * 64] = make_float4(sv_a[0], sv_a[1], sv_a[2], sv_a[3]);
      sp += 4 * 4 * 64;
    }
  };
  auto step = [&](auto first_tag, int s, G3& g) {            // g: this step's projections on entry; reloaded with those of step s + PD
    constexpr bool FIRST = decltype(first_tag)::value;
    // the recurrent sums start from zero and carry the factor S_w S_h; the projections / b_hn join after the exact post-scale
    const f32x4 g_r = {g.r.x, g.r.y, g.r.z, g.r.w}, g_z = {g.z.x, g.z.y, g.z.z, g.z.w}, acc_in = {g.n.x, g.n.y, g.n.z, g.n.w};
    f32x4 acc_r = {0.f, 0.f, 0.f, 0.f}, acc_z = acc_r, acc_hn = acc_r;
    STAMP(0);
    if constexpr (!FIRST) lds_barrier();      // h_{s-1} of every wave is in hbuf[cur]
    STAMP(1);
    f16x8 hq[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int p = 0; p < 2; ++p) hq[kb][p] = *(const f16x8*)&hb[cur][p][li][kb * 32 + lq * 8];
    __builtin_amdgcn_sched_barrier(0);        // all four ds_reads go out first (left alone, half of them sink below 24 MFMAs)
    // Memory instructions are NOT cheap for a lone wave (measured: ~125 cycles of wave time per global store
    // issued outside the MFMA stream), but they do overlap with a busy matrix pipe.  The projection prefetch
    // for step s+1 (3 loads) and the stores of step s-1 (h + 4 stash vectors) are therefore threaded through
    // the MFMA stream by hand, one memory instruction after every six MFMAs, fenced so they stay there.
    if (s + PD < n_steps) gq += 4 * 3 * 64;                                           // last PD steps: harmless reload of the last one
    // 18 fp16 MFMAs (3 gates x 2 k blocks x 3 cross terms, ~16.5 cycles each; 36 bf16 ones until round 4, 48 fp32 MFMAs at 32
    // cycles in round 1); the eight memory instructions ride between them as before
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      acc_r = mfma_f16x2<CT_FWD_REC>(Aw[0][kb], hq[kb], acc_r);
      __builtin_amdgcn_sched_barrier(0);
      if (kb == 0) { g.r = gq[0]; g.z = gq[64]; }
      else if constexpr (!FIRST) { *(float4*)hptr = make_float4(hprev[0], hprev[1], hprev[2], hprev[3]); hptr += hstep; }
      __builtin_amdgcn_sched_barrier(0);
      acc_z = mfma_f16x2<CT_FWD_REC>(Aw[1][kb], hq[kb], acc_z);
      __builtin_amdgcn_sched_barrier(0);
      if (kb == 0) g.n = gq[128];
      else if constexpr (!FIRST && STASH) { sp[0 * 64] = make_float4(sv_r[0], sv_r[1], sv_r[2], sv_r[3]); sp[1 * 64] = make_float4(sv_z[0], sv_z[1], sv_z[2], sv_z[3]); }
      __builtin_amdgcn_sched_barrier(0);
      acc_hn = mfma_f16x2<CT_FWD_REC>(Aw[2][kb], hq[kb], acc_hn);
      __builtin_amdgcn_sched_barrier(0);
      if (kb == 1) {
        if constexpr (!FIRST && STASH) {
          sp[3 * 64] = make_float4(sv_a[0], sv_a[1], sv_a[2], sv_a[3]);
          sp += 4 * 4 * 64;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(2);
    acc_r = g_r + acc_r * post[0]; acc_z = g_z + acc_z * post[1]; acc_hn = bhn + acc_hn * post[2];
    f32x4 r, z, n, hn;
    gru_gates(acc_r, acc_z, acc_in, acc_hn, hprev, r, z, n, hn);
    {   // split the four new state values once; 8 bytes per piece
      f16x4 hp[2];
      split2_quad(hn, F16X2_H_SCALE, hp[0], hp[1]);
#pragma unroll
      for (int p = 0; p < 2; ++p) *(f16x4*)&hb[cur ^ 1][p][li][u0] = hp[p];
    }
    hprev = hn; sv_r = r; sv_z = z; sv_a = acc_hn;
    cur ^= 1;
    STAMP(3);
  };
  step(std::true_type{}, 0, g[0]);
  int s = 1;                                               // s = 1 (mod PD) at the top of every round: step s + i uses set (1 + i) % PD
  for (; s + PD <= n_steps; s += PD)
    sfor_n<PD>([&](auto i) { step(std::false_type{}, s + decltype(i)::value, g[(1 + decltype(i)::value) % PD]); });
  sfor_n<PD - 1>([&](auto i) { if (s + decltype(i)::value < n_steps) step(std::false_type{}, s + decltype(i)::value, g[(1 + decltype(i)::value) % PD]); });
  flush();
#ifdef MSIG_STAMPS
  if (a.dbg && tid == 0 && blockIdx.y == 0 && blockIdx.x < 256)
    for (int i = 0; i < 8; ++i) a.dbg[(size_t)blockIdx.x * 8 + i] = ph_[i];
#endif
  if (D.h_last != nullptr && valid)
    *(float4*)(D.h_last + (int64_t)b * D.hl_bs + D.hl_col + u0) = make_float4(hprev[0], hprev[1], hprev[2], hprev[3]);
}

// ------------------------------------------------------------------------------------
// Backward recurrence (BPTT) of the latency form: gru_bwd_seq4 (gru_bwd4.hip, ROLE 3) — consumes the stash written by the
// forward kernels and replaces it with the pre-activation gradients (dr, dz, dn, dhn) that the bulk kernels below contract.
// Round 2's fp32-MFMA gru_bwd_seq (48 v_mfma_f32_16x16x4_f32 per wave-step, 1.23 us per step) is gone: the split-bf16 recurrence
// with the dh-independent gate math in its MFMA gaps and LDS-DMA operand prefetch takes 0.99 us (profiles/r03_bench_B64_kernels.log).
// ------------------------------------------------------------------------------------
// Bulk: dx[b][t][:] = W_ih^T dgi[b][t][:]   (dgi = dr,dz,dn of the stash)
// ------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------
// Fused backward — BPTT recurrence dh_{t-1} = dh_t z + W_hh^T dgh_t, dX = W_ih^T dgi and dW_ih / dW_hh / db in ONE kernel per
// layer — with EVERY contraction on split-bf16 MFMA (gru_bwd_b3): the throughput form above 192 batch tiles.  One workgroup owns
// a 16-row tile for all its steps and then moves on to its next tile with the dW accumulators still in registers (one partial
// per workgroup); the gate gradients of a step never leave the chip.  Round 1's gru_bwd_fused contracted dW on fp32 MFMA — 73 %
// of its matrix time for layer 0 — because dW sums over the 16 batch rows of a step and row-major bf16 planes cannot hand a
// lane 8 consecutive k of that index (2.82 / 2.41 ms per launch; removed in round 2, see the history).  Two things remove the
// obstacle:
//   * the contraction index of one v_mfma_f32_16x16x32_bf16 is (step parity, batch row): dW = sum_t sum_b dg_t[b]^T x_t[b]
//     sums over time as well, so K = 32 is TWO consecutive steps x 16 rows.  The gate-gradient / [x | h_prev] planes of a
//     step therefore live in a ring of THREE LDS buffers (step being written, step being propagated, the step before),
//     and dW is contracted every second step;
//   * gfx950's ds_read_b64_tr_b16 delivers a 4-row x 16-column block of a row-major plane column-major, i.e. exactly the
//     8 consecutive k (2 reads) of one unit / one input column that the A and the B fragment need (lane map checked by
//     tools/tr_dw_check.hip).  Lane group g = lane >> 4 (the MFMA's k group): step = g >> 1, rows 4 (g & 1) + 0..3 and
//     8 + 4 (g & 1) + 0..3; plane row strides are 8 * odd dwords so the eight rows a 32-lane half touches per read fall
//     into disjoint 8-bank windows.
// No fp32 tile is left in LDS (planes only), the recurrence of layer 1 is on split-bf16 as well (its W_hh^T
// pieces fit once the fp32 operand staging is gone), and every wave does the same work per pair of steps:
//   layer 0: recurrence 2 x 36, dX 36 (waves 0,1: the older step's two column blocks, waves 2,3: the newer step's),
//            dW_hh 72, dW_ih 36 bf16 MFMAs;   layer 1: recurrence 2 x 36, dX 2 x 72, dW_hh 72, dW_ih 144.
// Plane columns: [dr | dz | dhn | dn] (cols 0..191 = the rows of W_hh) and [x (I) | h_prev (64)].
// ------------------------------------------------------------------------------------
#define PIN_ACC(v) asm volatile("" : "+a"(v))

template <int I, bool FOLDS>
__global__ __launch_bounds__(256, 1) void gru_bwd_b3(const GruArgs a, int n_tiles, const FoldCtx fc) {
  using G = BwdB3<I>;
  constexpr bool L1K = G::L1K;
  constexpr int NKB = I / 16;                 // 16-wide column blocks of the input
  constexpr int NDX = L1K ? 2 : 1;            // dX column blocks per wave and step handled
  constexpr int SD = G::SD, SX = G::SX, DGP = G::DGP, XHP = G::XHP, BUFE = G::BUFE;
  extern __shared__ __attribute__((aligned(16))) __bf16 ring[];       // [3][ dg: 3 pieces x 16 x SD | xh: 3 pieces x 16 x SX ]
  FOLD_GRU_ARGS_IF(FOLDS);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int u0 = w * 16 + lq * 4;

  // ---- resident A operands, split once: six 32-wide k blocks over the 192 gate rows [r|z|n] ----
  //   recurrence  A[i = li][k] = W_hh[k][w*16 + li]        dX  A[i = li][k] = W_ih[k][cb*16 + li]
  bf16x8 AhB[6][3], AiB[NDX][6][3];
#pragma unroll
  for (int kb = 0; kb < 6; ++kb)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 p0, p1, p2;
      split3(D.Whh[(size_t)(kb * 32 + lq * 8 + j) * 64 + w * 16 + li], p0, p1, p2);
      AhB[kb][0][j] = p0; AhB[kb][1][j] = p1; AhB[kb][2][j] = p2;
#pragma unroll
      for (int kk = 0; kk < NDX; ++kk) {
        const int cb = L1K ? (2 * w + kk) : (w & 1);
        split3(D.Wih[(size_t)(kb * 32 + lq * 8 + j) * I + cb * 16 + li], p0, p1, p2);
        AiB[kk][kb][0][j] = p0; AiB[kk][kb][1][j] = p1; AiB[kk][kb][2][j] = p2;
      }
    }
  // The resident operands are only ever read by MFMAs, which take A / B from either half of the unified register file; left to
  // itself the allocator keeps part of them in arch VGPRs and parks the prefetched loads in AccVGPRs, from where every value the
  // gate math needs has to be copied back first (58 / 90 v_accvgpr_read per wave-step in layer 0 / 1, a quarter of the VALU work).
  // Passing them through an "a"-constrained asm pins them to AccVGPRs for the rest of the kernel.
#ifndef MSIG_NO_PIN
#pragma unroll
  for (int kb = 0; kb < 6; ++kb)
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) {
      PIN_ACC(AhB[kb][pp]);
#pragma unroll
      for (int kk = 0; kk < NDX; ++kk) PIN_ACC(AiB[kk][kb][pp]);
    }
#endif
  // ---- persistent accumulators: this wave's 16 units (w*16 ..) of every gate ----
  //   accH[g][cb]: dW_hh rows g*64 + w*16 + 4 lq + e (g = r, z, n<-dhn), columns cb*16 + li
  //   accI[g][cb]: dW_ih rows likewise (g = r, z, n<-dn),               columns cb*16 + li
  f32x4 accH[3][4], accI[3][NKB];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) accH[g][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < NKB; ++cb) accI[g][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#ifndef MSIG_NO_PIN
  if constexpr (!L1K) {                       // layer 0: 144 weight + 72 accumulator registers fit the 256 AccVGPRs
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) PIN_ACC(accH[g][cb]);
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) PIN_ACC(accI[g][cb]);
    }
  }
#endif
  float bacc[4][4];                           // bias gradients of this lane's (row, 4 units), summed over steps and tiles
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) bacc[g][e] = 0.f;
  constexpr int NXV = (16 * I / 4 + 255) / 256;     // float4 pieces of the x tile per thread
  // layer 0 <=> D.dh_mode 0 (upstream gradient at every step, inter-layer dropout mask); layer 1 <=> dh_mode 1 (launch_gru_bwd)
  constexpr int dh_mode = L1K ? 1 : 0;
  const int n_steps = D.n_steps, t_start = D.t_start, t_sign = D.t_sign;
  const int dthr = dh_mode == 0 ? a.drop_thr : 0, xthr = a.x_drop_thr;
  const uint32_t dkey = akey_, xkey = axkey_;
  const float dscale = a.drop_scale, xscale = a.x_drop_scale;
  const int64_t h_bs = D.h_bs, h_ts = D.h_ts, dh_bs = D.dh_bs, dh_ts = D.dh_ts, x_bs = a.x_bs, x_ts = a.x_ts;
  const int64_t dx_bs = D.dx_bs, dx_ts = D.dx_ts;
  const int dh_col = D.dh_col;
  const float* hbase = D.h + D.h_col + u0;
  const float* dhbase = D.dh + D.dh_col + u0;
  const float* xbase = ax_;
  float* dxbase = D.dx + lq * 4;
  const int64_t hstep = (int64_t)t_sign * h_ts, ustep = (dh_mode == 0) ? (int64_t)t_sign * dh_ts : 0;
  const int64_t xstep = (int64_t)t_sign * x_ts, dxstep = (int64_t)t_sign * dx_ts;

  // ---- per-lane LDS offsets (elements, relative to a ring buffer) ----
  // Bank swizzles.  All four waves write their planes at the same time and then wait for the barrier, so write conflicts
  // are exposed time.  With 8 * odd-dword row strides (what the transposed reads need) the 16 rows of a plain layout land
  // on 4 distinct banks (4-way).  Gate-gradient planes (read by rows with ds_read_b128 AND transposed): 16-byte pairs
  // XORed with bit 2 of the row -> 2-way stores, the floor for 8-byte stores into 16-byte-aligned rows.  [x | h_prev]
  // planes (only ever read transposed, 8-byte granules): 8-byte chunks XORed with bits 2..3 of the row -> conflict-free.
  // Transposed reads stay conflict-free: a row's four chunks of a 16-column block are permuted among themselves.
  const int sw_li = ((li >> 2) & 1) * 8;                                 // dg planes: element XOR for row li
  const int rd_row = li * SD + ((lq * 8) ^ sw_li);                       // row reads (recurrence, dX): B[k = 8 lq + j][n = li]
  const int wr_dg = li * SD + (u0 ^ sw_li);                              // this lane's 4-unit chunk of each gate
  const int wr_h = 3 * DGP + li * SX + ((I + u0) ^ (((li >> 2) & 3) * 4));
  // transposed reads: lane 16 g + i supplies the address of row 8 h + 4 (g & 1) + (i >> 2), columns c0 + 4 (i & 3) .. + 3
  const int trow = 4 * (lq & 1) + (li >> 2);
  const int tr_dg = trow * SD + ((4 * (li & 3)) ^ ((lq & 1) * 8));       // (row >> 2) & 1 = g & 1 for h = 0 and h = 1
  const int tr_xh0 = 3 * DGP + trow * SX + ((4 * (li & 3)) ^ ((lq & 1) * 4));              // h = 0: (row >> 2) & 3 = g & 1
  const int tr_xh1 = 3 * DGP + (trow + 8) * SX + ((4 * (li & 3)) ^ ((2 + (lq & 1)) * 4));  // h = 1: 2 + (g & 1)
  const bool newer = (lq >> 1) != 0;                                     // k groups 2,3 contract the newer step of a pair
  int xrow_off[NXV]; bool xlive[NXV];
#pragma unroll
  for (int v = 0; v < NXV; ++v) {
    const int idx = tid + 256 * v, row = idx / (I / 4), c4 = idx - row * (I / 4);
    xlive[v] = idx < 16 * I / 4;
    xrow_off[v] = xlive[v] ? 3 * DGP + row * SX + ((4 * c4) ^ (((row >> 2) & 3) * 4)) : 3 * DGP;
  }

  // Operands of a step that come from HBM (stash r,z,n,hn; h_{t-1}; upstream dh; the x pieces), prefetched into register
  // sets.  Layer 0 keeps TWO sets (steps m and m+1: a step's loads are issued two iterations before its gate math — one
  // iteration of distance left ~235 cycles of wait per step under this kernel's 3.4 TB/s) and issues both refills inside the
  // odd iteration's dX / dW MFMA stream; layer 1 has registers for one set only and threads its loads through dX.
  constexpr int NSETS = L1K ? 1 : 2;
  constexpr int NPIECE = 6 + NXV;                     // separately placeable load instructions of a step (slot 2 is empty: no n in the stash)
  struct LoadSet {
    float4 r4, z4, hn4, hp4, up4, xv[NXV];
    uint32_t wd_u; float sc_u, hkeep;
  };
  struct TileState {
    bool valid; float vmask;
    const float4* sp; const float* hq; const float* uq; uint32_t ue; const float* xq[NXV]; uint32_t xe[NXV]; float* dxq;
    float dhz[4];
    float4 hcur;      // h_t of the step whose gate math comes next: n_t is recovered from it (gru_n_from_h, msig_dev.h)
  };
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    TileState t;
    LoadSet ls[NSETS];
    const int tl = t_start + t_sign * (n_steps - 1);                 // time index of the last step (processed first)
    {
      const int b = tile * 16 + li;
      t.valid = b < a.B;
      const int bl = t.valid ? b : a.B - 1;
      t.vmask = t.valid ? 1.0f : 0.0f;
      t.sp = D.stash + ((size_t)((size_t)tile * n_steps + (n_steps - 1)) * 4 + w) * 4 * 64 + lane;
      t.hq = hbase + (int64_t)bl * h_bs + (int64_t)(n_steps > 1 ? tl - t_sign : tl) * h_ts;   // h_{t-1} of the last step
      t.hcur = *(const float4*)(hbase + (int64_t)bl * h_bs + (int64_t)tl * h_ts);             // h_t of the last step
      t.uq = dhbase + (int64_t)bl * dh_bs + (int64_t)(dh_mode == 0 ? tl : 0) * dh_ts;
      t.ue = (uint32_t)((int64_t)bl * dh_bs + (int64_t)(dh_mode == 0 ? tl : 0) * dh_ts + dh_col + u0);
#pragma unroll
      for (int v = 0; v < NXV; ++v) {
        const int idx = (tid + 256 * v) % (16 * I / 4), row = idx / (I / 4), c4 = idx - row * (I / 4);
        const int bb = min(tile * 16 + row, a.B - 1);
        const int64_t x0 = (int64_t)bb * x_bs + (int64_t)tl * x_ts + 4 * c4;
        t.xq[v] = xbase + x0;
        t.xe[v] = (uint32_t)x0;
      }
      // dX store pointer: layer 1 stores every step; layer 0 stores per pair — waves 0,1 the older step, waves 2,3 the newer
      const int first = L1K ? 0 : (w >> 1);
      t.dxq = dxbase + (int64_t)b * dx_bs + (int64_t)(tl - t_sign * first) * dx_ts;      // only dereferenced when valid and in range
#pragma unroll
      for (int q = 0; q < NSETS; ++q) { ls[q].wd_u = 0; ls[q].sc_u = 0.f; ls[q].hkeep = 0.f; }
    }
    // Piece i of the loads of time step s into set L: only ISSUES — a consumer placed next to a load drags an s_waitcnt vmcnt(0) with it and exposes the full HBM latency every step; every consumer of a loaded value
    // sits in `gates`, at least one iteration later.  The pointers address step s and move on to s-1 with their last user;
    // beyond step 0 they stay put (harmless reloads of valid addresses).
    auto load_piece = [&](LoadSet& L, int i, int s) {
      if (i == 0) L.r4 = t.sp[0];
      if (i == 1) L.z4 = t.sp[64];
      if (i == 3) { L.hn4 = t.sp[192]; if (s > 0) t.sp -= 4 * 4 * 64; }
      if (i == 4) { L.hp4 = *(const float4*)t.hq; if (s > 1) t.hq -= hstep; L.hkeep = (s == 0) ? 0.0f : 1.0f; }
      if (i == 5) {
        if constexpr (L1K) {      // dh_mode 1: the upstream gradient enters at the last time step only — the prologue's step
          if (s == n_steps - 1) L.up4 = *(const float4*)t.uq;
        } else {
          L.up4 = *(const float4*)t.uq;
          L.wd_u = drop_word(t.ue, dkey);
          L.sc_u = dscale * t.vmask;
          if (s > 0) { t.uq -= ustep; t.ue -= (uint32_t)ustep; }
        }
      }
#pragma unroll
      for (int v = 0; v < NXV; ++v)
        if (i == 6 + v) {
          L.xv[v] = *(const float4*)t.xq[v];
          if (s > 0) { t.xq[v] -= xstep; t.xe[v] -= (uint32_t)xstep; }
        }
    };
    auto issue_loads = [&](LoadSet& L, int s) {
#pragma unroll
      for (int i = 0; i < NPIECE; ++i) load_piece(L, i, s);
    };
    // gate gradients of one step from (stash, h_{t-1}, upstream dh, carried dh) -> bf16 planes of ring buffer `boff`
    auto gates = [&](LoadSet& L, int s, const f32x4& dh_in, int boff, auto first_tag) {     // s = time step being processed
      constexpr bool FIRST = decltype(first_tag)::value;        // the tile's first processed step (time step n_steps-1)
      const float rr[4] = {L.r4.x, L.r4.y, L.r4.z, L.r4.w}, zz[4] = {L.z4.x, L.z4.y, L.z4.z, L.z4.w};
      const float hh[4] = {L.hn4.x, L.hn4.y, L.hn4.z, L.hn4.w}, hc[4] = {t.hcur.x, t.hcur.y, t.hcur.z, t.hcur.w};
      const float hp[4] = {L.hp4.x * L.hkeep, L.hp4.y * L.hkeep, L.hp4.z * L.hkeep, L.hp4.w * L.hkeep};   // h_{-1} = 0
      t.hcur = L.hp4;                                     // this step's h_{t-1} is the next processed step's h_t
      float up[4];
      if constexpr (L1K) {
        if constexpr (FIRST) { up[0] = L.up4.x * t.vmask; up[1] = L.up4.y * t.vmask; up[2] = L.up4.z * t.vmask; up[3] = L.up4.w * t.vmask; }
        else { up[0] = up[1] = up[2] = up[3] = 0.0f; }
      } else {
        up[0] = L.up4.x * drop_mul(L.wd_u, 0, dthr, L.sc_u); up[1] = L.up4.y * drop_mul(L.wd_u, 1, dthr, L.sc_u);
        up[2] = L.up4.z * drop_mul(L.wd_u, 2, dthr, L.sc_u); up[3] = L.up4.w * drop_mul(L.wd_u, 3, dthr, L.sc_u);
      }
      float dr[4], dz[4], dn[4], dhn[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dh = dh_in[e] + up[e];
        const float omz = 1.0f - zz[e];
        const float nn = gru_n_from_h(hc[e], hp[e], zz[e], omz);
        const float dnn = dh * omz;
        dn[e] = dnn * (1.0f - nn * nn);
        dz[e] = dh * (hp[e] - nn) * zz[e] * omz;
        dr[e] = dn[e] * hh[e] * rr[e] * (1.0f - rr[e]);
        dhn[e] = dn[e] * rr[e];
        t.dhz[e] = dh * zz[e];
        bacc[0][e] += dr[e]; bacc[1][e] += dz[e]; bacc[2][e] += dhn[e]; bacc[3][e] += dn[e];     // plane column order [dr|dz|dhn|dn]
      }
      __bf16* pw = ring + boff + wr_dg;
      bf16x4 pc[4][3];
      split3_quad(dr, pc[0]); split3_quad(dz, pc[1]); split3_quad(dhn, pc[2]); split3_quad(dn, pc[3]);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) *(bf16x4*)&pw[pp * DGP + g * 64] = pc[g][pp];
      {
        bf16x4 hpc[3];
        split3_quad(hp, hpc);
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) *(bf16x4*)&ring[boff + wr_h + pp * XHP] = hpc[pp];
      }
#pragma unroll
      for (int v = 0; v < NXV; ++v) {
        float q[4] = {L.xv[v].x, L.xv[v].y, L.xv[v].z, L.xv[v].w};
        if constexpr (L1K) {      // the layer-1 input is the dropped layer-0 output; t.xe has moved on one step since this step's load
          const uint32_t wdx = drop_word(t.xe[v] + (uint32_t)(s > 0 ? xstep : 0), xkey);
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] *= drop_mul(wdx, e, xthr, xscale);
        }
        bf16x4 xpc[3];
        split3_quad(q, xpc);
        if (xlive[v]) {
#pragma unroll
          for (int pp = 0; pp < 3; ++pp) *(bf16x4*)&ring[boff + xrow_off[v] + pp * XHP] = xpc[pp];
        }
      }
    };
    auto recurrence = [&](int boff) -> f32x4 {                     // dh_{s-1} = dh_s * z_s + W_hh^T dgh_s
      const __bf16* pb = ring + boff + rd_row;
      f32x4 ah0 = {0.f, 0.f, 0.f, 0.f}, ah1 = {0.f, 0.f, 0.f, 0.f};
      // The operand reads go out ahead of the MFMAs that consume them (the chain is the step's critical path; left alone the
      // compiler issues four reads at a time and waits): all 18 at once for layer 0; for layer 1, which has no registers for
      // 72 operand VGPRs next to its 360 resident ones, in two halves (reading the second half under the first half's MFMAs
      // was measured slower: 48 live operand registers push resident weights into scratch).
      constexpr int KH = L1K ? 3 : 6;
#pragma unroll
      for (int k0 = 0; k0 < 6; k0 += KH) {
        bf16x8 q[KH][3];
#pragma unroll
        for (int kb = 0; kb < KH; ++kb)                            // columns [dr|dz|dhn] = 0..191
#pragma unroll
          for (int pp = 0; pp < 3; ++pp) q[kb][pp] = *(const bf16x8*)&pb[pp * DGP + (k0 + kb) * 32];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < KH; ++kb) {
          if ((k0 + kb) & 1) ah1 = mfma_bf16x3<CT_BWD_REC>(AhB[k0 + kb], q[kb], ah1); else ah0 = mfma_bf16x3<CT_BWD_REC>(AhB[k0 + kb], q[kb], ah0);
        }
        if constexpr (L1K) __builtin_amdgcn_sched_barrier(0);
      }
      f32x4 dh_next;
#pragma unroll
      for (int e = 0; e < 4; ++e) dh_next[e] = t.dhz[e] + ah0[e] + ah1[e];
      return dh_next;
    };
    // `hook(slot)` is called after every MFMA group (6 * NDX slots): the callers thread the global loads of later steps
    // through the MFMA stream there — a memory instruction costs a lone wave ~100 cycles outside an MFMA stream and almost
    // nothing inside one (gru_fwd_rec) — fenced so they stay where they are put.
    f32x4 dx_pend0 = {0.f, 0.f, 0.f, 0.f}, dx_pend1 = {0.f, 0.f, 0.f, 0.f}; bool dx_pend_store = false; int64_t dx_pend_adv = 0;
    auto dx_phase = [&](int boff, bool store, int64_t advance, auto&& hook) {     // dx_t = W_ih^T dgi_t of the step in buffer `boff`
      f32x4 ax[NDX][2];
#pragma unroll
      for (int kk = 0; kk < NDX; ++kk) ax[kk][0] = ax[kk][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const __bf16* pb = ring + boff + rd_row;
      bf16x8 q[2][3];                                               // operands of k block kb+1 are read under the MFMAs of kb
      auto rd = [&](int kb) {                                       // gate rows [r|z|n] <-> columns [dr|dz| . |dn]
        const int col0 = kb < 4 ? kb * 32 : 192 + (kb - 4) * 32;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) q[kb & 1][pp] = *(const bf16x8*)&pb[pp * DGP + col0];
      };
      rd(0);
#pragma unroll
      for (int kb = 0; kb < 6; ++kb) {
        if (kb + 1 < 6) rd(kb + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < NDX; ++kk) {
          ax[kk][kb & 1] = mfma_bf16x3<CT_DX>(AiB[kk][kb], q[kb & 1], ax[kk][kb & 1]);
          __builtin_amdgcn_sched_barrier(0);
          hook(kb * NDX + kk);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (L1K) {
        if (store) {
#pragma unroll
          for (int kk = 0; kk < NDX; ++kk)
            *(float4*)(t.dxq + (2 * w + kk) * 16) = make_float4(ax[kk][0][0] + ax[kk][1][0], ax[kk][0][1] + ax[kk][1][1],
                                                                ax[kk][0][2] + ax[kk][1][2], ax[kk][0][3] + ax[kk][1][3]);
        }
        t.dxq -= advance;
      } else {        // layer 0 has the registers to let the store wait for the dW MFMA stream (dx_flush): no MFMA drain, no lone store
        dx_pend0 = ax[0][0]; dx_pend1 = ax[0][1]; dx_pend_store = store; dx_pend_adv = advance;
      }
    };
    auto dx_flush = [&]() {
      if (dx_pend_store)
        *(float4*)(t.dxq + (w & 1) * 16) = make_float4(dx_pend0[0] + dx_pend1[0], dx_pend0[1] + dx_pend1[1],
                                                       dx_pend0[2] + dx_pend1[2], dx_pend0[3] + dx_pend1[3]);
      t.dxq -= dx_pend_adv;
    };
    // one fragment (three pieces) of eight consecutive k = (step, row) for column c of a plane, by two transposed reads each
    auto tr_frag = [&](int off0, int off1, int pstride, bf16x8 (&f)[3]) {       // off0 / off1: rows 0..7 / 8..15 of the block
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) {
        const bf16x4 lo = lds_tr_read(ring + off0 + pp * pstride), hi = lds_tr_read(ring + off1 + pp * pstride);
        f[pp] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    };
    auto dw_phase = [&](int older, int newer_off, auto&& hook) {      // dW += dg^T [x | h_prev] over the two steps in buffers `older`, `newer_off`
      const int sel = newer ? newer_off : older;
      const int ta = sel + tr_dg + w * 16;
      if constexpr (!L1K) {
        bf16x8 Ar[3], Az[3], Ahn[3], An[3];
        tr_frag(ta + 0 * 64, ta + 0 * 64 + 8 * SD, DGP, Ar);
        tr_frag(ta + 1 * 64, ta + 1 * 64 + 8 * SD, DGP, Az);
        tr_frag(ta + 2 * 64, ta + 2 * 64 + 8 * SD, DGP, Ahn);
        tr_frag(ta + 3 * 64, ta + 3 * 64 + 8 * SD, DGP, An);
        // B fragments: the four h_prev column blocks, then the x column blocks; block bi+1 is read under the MFMAs of bi
        bf16x8 Bf[2][3];
        auto rdB = [&](int bi) {
          const int c0 = bi < 4 ? I + bi * 16 : (bi - 4) * 16;
          tr_frag(sel + tr_xh0 + c0, sel + tr_xh1 + c0, XHP, Bf[bi & 1]);
        };
        rdB(0);
#pragma unroll
        for (int bi = 0; bi < 4 + NKB; ++bi) {
          if (bi + 1 < 4 + NKB) rdB(bi + 1);
          __builtin_amdgcn_sched_barrier(0);
          if (bi < 4) {
            accH[0][bi] = mfma_bf16x3<CT_DW>(Ar, Bf[bi & 1], accH[0][bi]);
            __builtin_amdgcn_sched_barrier(0); hook(3 * bi + 0); __builtin_amdgcn_sched_barrier(0);
            accH[1][bi] = mfma_bf16x3<CT_DW>(Az, Bf[bi & 1], accH[1][bi]);
            __builtin_amdgcn_sched_barrier(0); hook(3 * bi + 1); __builtin_amdgcn_sched_barrier(0);
            accH[2][bi] = mfma_bf16x3<CT_DW>(Ahn, Bf[bi & 1], accH[2][bi]);
            __builtin_amdgcn_sched_barrier(0); hook(3 * bi + 2); __builtin_amdgcn_sched_barrier(0);
          } else {
            accI[0][bi - 4] = mfma_bf16x3<CT_DW>(Ar, Bf[bi & 1], accI[0][bi - 4]);
            accI[1][bi - 4] = mfma_bf16x3<CT_DW>(Az, Bf[bi & 1], accI[1][bi - 4]);
            accI[2][bi - 4] = mfma_bf16x3<CT_DW>(An, Bf[bi & 1], accI[2][bi - 4]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
        // Layer 1 holds 360 resident VGPRs (weights 216, accumulators 144): the A fragments of r and z (24 VGPRs) are live
        // together, then dhn's, then dn's, and the B fragments are read once per pass (24 instead of 12 fragment reads).
        {
          bf16x8 Ar[3], Az[3];
          tr_frag(ta + 0 * 64, ta + 0 * 64 + 8 * SD, DGP, Ar);
          tr_frag(ta + 1 * 64, ta + 1 * 64 + 8 * SD, DGP, Az);
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            bf16x8 Bh[3];
            tr_frag(sel + tr_xh0 + I + cb * 16, sel + tr_xh1 + I + cb * 16, XHP, Bh);
            accH[0][cb] = mfma_bf16x3<CT_DW>(Ar, Bh, accH[0][cb]);
            accH[1][cb] = mfma_bf16x3<CT_DW>(Az, Bh, accH[1][cb]);
          }
#pragma unroll
          for (int cb = 0; cb < NKB; ++cb) {
            bf16x8 Bx[3];
            tr_frag(sel + tr_xh0 + cb * 16, sel + tr_xh1 + cb * 16, XHP, Bx);
            accI[0][cb] = mfma_bf16x3<CT_DW>(Ar, Bx, accI[0][cb]);
            accI[1][cb] = mfma_bf16x3<CT_DW>(Az, Bx, accI[1][cb]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          bf16x8 Ahn[3];
          tr_frag(ta + 2 * 64, ta + 2 * 64 + 8 * SD, DGP, Ahn);
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            bf16x8 Bh[3];
            tr_frag(sel + tr_xh0 + I + cb * 16, sel + tr_xh1 + I + cb * 16, XHP, Bh);
            accH[2][cb] = mfma_bf16x3<CT_DW>(Ahn, Bh, accH[2][cb]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
          bf16x8 An[3];
          tr_frag(ta + 3 * 64, ta + 3 * 64 + 8 * SD, DGP, An);
#pragma unroll
          for (int cb = 0; cb < NKB; ++cb) {
            bf16x8 Bx[3];
            tr_frag(sel + tr_xh0 + cb * 16, sel + tr_xh1 + cb * 16, XHP, Bx);
            accI[2][cb] = mfma_bf16x3<CT_DW>(An, Bx, accI[2][cb]);
          }
        }
      }
    };
    auto zero_fill = [&](int boff) {                     // phantom partner of an unpaired last step: dg = 0, operands finite
      for (int i = tid; i < BUFE / 8; i += 256) *(float4*)&ring[boff + 8 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
    };

    int cur = 0, nxt = BUFE, prv = 2 * BUFE;
    STAMP_DECL;
    auto no_hook = [](int) {};
    auto clamp0 = [](int s) { return s > 0 ? s : 0; };
    // Processing index j = 0 .. n_steps-1 (time step s = n_steps-1-j).  Iteration j: the recurrence of step j (short, on the
    // critical path) yields dh for step j+1, whose gate math and plane writes follow; then the dX / dW MFMAs of the steps
    // already in LDS with the loads of later steps threaded through them.  One barrier per step.
    // Layer 0: set m & 1 holds step m; prologue loads steps 0, 1, 2; the odd iteration j1 refills set 1 with step j1+2
    // (consumed by the next even iteration) and set 0 with step j1+3.  Layer 1: one set, refilled right after its use.
    issue_loads(ls[0], n_steps - 1);
    gates(ls[0], n_steps - 1, (f32x4){0.f, 0.f, 0.f, 0.f}, cur, std::true_type{});
    if constexpr (!L1K) {
      issue_loads(ls[1], clamp0(n_steps - 2));
      issue_loads(ls[0], clamp0(n_steps - 3));
    } else {
      issue_loads(ls[0], clamp0(n_steps - 2));
    }
    lds_barrier();
    const int n_pairs = (n_steps + 1) >> 1;
    for (int p = 0; p < n_pairs; ++p) {
      const int j0 = 2 * p, j1 = j0 + 1;
      // ---- even iteration: step j0 is in `cur` ----
      STAMP(0);
      if (j1 < n_steps) {
        const f32x4 dh_next = recurrence(cur);
        STAMP(1);
#ifdef MSIG_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // diagnostic only: how long the step waits for its prefetched operands
        STAMP(7);
#endif
        gates(ls[NSETS - 1], n_steps - 1 - j1, dh_next, nxt, std::false_type{});                  // step j1: set 1 (layer 0) / the set (layer 1)
        STAMP(2);
      } else {
        zero_fill(nxt);
      }
      if constexpr (L1K) {
        const int sl = clamp0(n_steps - 1 - (j0 + 2));
        dx_phase(cur, t.valid, dxstep, [&](int slot) { if (slot < NPIECE) load_piece(ls[0], slot, sl); });
      }
      STAMP(4);
      lds_barrier();
      STAMP(6);
      { const int o = prv; prv = cur; cur = nxt; nxt = o; }
      // ---- odd iteration: step j1 (or the all-zero phantom) is in `cur`, step j0 in `prv` ----
      if (j1 + 1 < n_steps) {
        const f32x4 dh_next = recurrence(cur);
        STAMP(1);
#ifdef MSIG_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(7);
#endif
        gates(ls[0], n_steps - 2 - j1, dh_next, nxt, std::false_type{});                          // step j1 + 1
        STAMP(2);
      }
      if constexpr (L1K) {
        const int sl = clamp0(n_steps - 1 - (j1 + 2));
        if (j1 < n_steps) dx_phase(cur, t.valid, dxstep, [&](int slot) { if (slot < NPIECE) load_piece(ls[0], slot, sl); });
        else issue_loads(ls[0], sl);
        STAMP(4);
        dw_phase(prv, cur, no_hook);
      } else {
        // 2 * NPIECE load pieces over the 6 dX + 12 dW_hh MFMA groups: step j1+2 -> set 1, then step j1+3 -> set 0
        const int sa = clamp0(n_steps - 1 - (j1 + 2)), sb = clamp0(n_steps - 1 - (j1 + 3));
        auto piece = [&](int i) {
          if (i < NPIECE) load_piece(ls[1], i, sa);
          else if (i < 2 * NPIECE) load_piece(ls[0], i - NPIECE, sb);
        };
        dx_phase((w >> 1) ? cur : prv, t.valid && ((w >> 1) == 0 || j1 < n_steps), 2 * dxstep, [&](int slot) { piece(slot); });
        STAMP(4);
        dw_phase(prv, cur, [&](int slot) { if (slot == 2) dx_flush(); piece(6 + slot); });
      }
      STAMP(5);
      lds_barrier();
      STAMP(6);
      { const int o = prv; prv = cur; cur = nxt; nxt = o; }
    }
#ifdef MSIG_STAMPS
    if (a.dbg && tid == 0 && tile == (int)blockIdx.x)
      for (int i = 0; i < 8; ++i) a.dbg[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + i] = ph_[i];
#endif
  }
  // ---- partial: [dW_ih 192*I][dW_hh 192*64][db 256 = dr,dz,dn,dhn] ----
  float* P = D.part + (size_t)blockIdx.x * (192 * I + 192 * 64 + 256);
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = g * 64 + w * 16 + lq * 4 + e;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) P[192 * I + (size_t)row * 64 + cb * 16 + li] = accH[g][cb][e];
#pragma unroll
      for (int cb = 0; cb < NKB; ++cb) P[(size_t)row * I + cb * 16 + li] = accI[g][cb][e];
    }
  // bias gradients: fold the 16 batch rows through LDS (every wave is past its last read of the ring: the loop ends on a
  // barrier).  Scratch columns are [dr|dz|dhn|dn]; the partial wants [dr|dz|dn|dhn].
  float* scratch = (float*)ring;
#pragma unroll
  for (int g = 0; g < 4; ++g) *(float4*)&scratch[li * RS + g * 64 + u0] = make_float4(bacc[g][0], bacc[g][1], bacc[g][2], bacc[g][3]);
  __syncthreads();
  float bsum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) bsum += scratch[r * RS + tid];
  P[192 * I + 192 * 64 + (tid < 128 ? tid : (tid < 192 ? tid + 64 : tid - 64))] = bsum;
}

// ------------------------------------------------------------------------------------
// Bulk kernels of the latency form, round 5: dW on split-bf16 and dX + dW of a layer as ONE launch.
//   * gru_bwd_dw contracted dW on v_mfma_f32_16x16x4_f32 (144 / 72 instructions of 32 cycles per wave and unit): the only GRU
//     contraction left on the fp32 pipe, and matrix-bound in a fold batch (0.22 + 0.19 ms of a 15-fold super-step).  dw2 uses
//     gru_bwd_b3's recipe: the contraction index of one v_mfma_f32_16x16x32_bf16 is (unit of a PAIR, batch row) — dW sums over
//     every (tile, step) unit, so ANY two units of a workgroup's share pair up — the gate gradients and [x | h_prev] of the two
//     units go to LDS as three bf16 piece planes each, and ds_read_b64_tr_b16 hands every lane its 8 consecutive k.  108 / 216
//     bf16 MFMAs of 16 cycles per wave and PAIR: 2.7 x fewer matrix cycles.  Plane geometry, swizzles and the transposed-read
//     lane map are BwdB3<I>'s (derivation there; exact-integer check tools/dw32_check.hip).
//   * a workgroup takes MSIG_DW_UNITS_PER_WG = 8 units (4 pairs) and leaves one 150 KB / 75 KB partial row for colsum_adam (16 units
//     halve the rows and the bytes of a fold batch but stretch one model's launch by ~7 us; the grouping fixes the summation order, so
//     it is the same for every fold count).
//   * dX and dW of a layer depend on the same recurrence output and on nothing of each other.  Layer 0, one model: one launch
//     (gru_bwd_dxdw<32>: blockIdx.x < gdx = dX workgroups, the rest dW); fold batches: two (the dX workgroups' two per CU).
//     Layer 1: the dX launch (gru_bwd_dx<128, true>) has the reverse direction's ONE step, whose dX accumulates into DH0[:, T'-1],
//     folded in — the dX workgroup of tile i computes the forward direction's last step AND the reverse step of that tile and stores
//     the sum (acc_rev + acc_fwd, the order of the two former launches) — and its dW, which layer 0's recurrence does not wait
//     for, rides in THAT launch beside the chains (gru_bwd4.hip gru_bwd_seq4_dw1; the role itself: gru_dw2.h).
//     Round 4's seven bulk launches of the two layers (dx_l1, dx_l1rev, dw_l1, dw_l1rev, dx_l0, dw_l0 ...) are two (one model).
// ------------------------------------------------------------------------------------
// dX role: gru_bwd_dx's arithmetic (dx[b][t][:] = W_ih^T dgi[b][t][:]).  D2 != nullptr (layer 1): the reverse direction's single
// step — the workgroup that owns tile i's last forward step also contracts the reverse step of that tile and stores the sum.
template <int I>
__device__ __forceinline__ void dx2_role(const GruArgs& a, const GruDir& D, const GruDir* D2, const int n_tiles, const int wg, const int nwg,
                                         __bf16* lds) {
  constexpr int NKB = I / 16;
  constexpr int KBW = (NKB >= 4) ? NKB / 4 : 1;
  constexpr int PS = BwdDw2<I>::DXP;
  __bf16 (*dgp)[16][PS] = (__bf16 (*)[16][PS])lds;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const bool active = (w * KBW) < NKB;
  const int sw_li = quad_swz(li);
  bf16x8 At[KBW][6][3];
  auto load_w = [&](const float* Wih) {
#pragma unroll
    for (int kk = 0; kk < KBW; ++kk)
#pragma unroll
      for (int kb = 0; kb < 6; ++kb)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          __bf16 p0, p1, p2;
          split3(active ? Wih[(size_t)(kb * 32 + lq * 8 + j) * I + (w * KBW + kk) * 16 + li] : 0.f, p0, p1, p2);
          At[kk][kb][0][j] = p0; At[kk][kb][1][j] = p1; At[kk][kb][2][j] = p2;
        }
  };
  auto fetch = [&](const GruDir& G, int unit, float4 (&g)[3]) {     // dr, dz, dn of (row li, units w*16 + lq*4 ..): only ISSUES the loads
    const float4* sp = G.stash + ((size_t)unit * 4 + w) * 4 * 64 + lane;
    g[0] = sp[0]; g[1] = sp[64]; g[2] = sp[128];
  };
  auto contract = [&](const float4 (&g)[3], f32x4 (&acc)[KBW]) {
    __syncthreads();    // previous unit's reads are done
#pragma unroll
    for (int gg = 0; gg < 3; ++gg) {
      const float v[4] = {g[gg].x, g[gg].y, g[gg].z, g[gg].w};
      bf16x4 p[3];
      split3_quad(v, p);
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) *(bf16x4*)&dgp[pp][li][(gg * 64 + w * 16 + lq * 4) ^ sw_li] = p[pp];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KBW; ++kk) acc[kk] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (active) {
#pragma unroll
      for (int kb = 0; kb < 6; ++kb) {
        bf16x8 q[3];
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) q[pp] = *(const bf16x8*)&dgp[pp][li][kb * 32 + ((lq * 8) ^ sw_li)];
#pragma unroll
        for (int kk = 0; kk < KBW; ++kk) acc[kk] = mfma_bf16x3<CT_DX>(At[kk][kb], q, acc[kk]);
      }
    }
  };
  auto store = [&](const GruDir& G, int tile, int t, const f32x4 (&acc)[KBW]) {
    const int b = tile * 16 + li;
    if (active && b < a.B) {
#pragma unroll
      for (int kk = 0; kk < KBW; ++kk) {
        float* dst = G.dx + (int64_t)b * G.dx_bs + (int64_t)t * G.dx_ts + (w * KBW + kk) * 16 + lq * 4;
        float4 o = make_float4(acc[kk][0], acc[kk][1], acc[kk][2], acc[kk][3]);
        if (G.dx_accumulate) {
          const float4 p = *(const float4*)dst;
          o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        *(float4*)dst = o;
      }
    }
  };
  const int n_units = n_tiles * D.n_steps;
  // With a reverse step to fold in, the first n_tiles workgroups do only that (two weight loads, two units); the units of the
  // main loop are dealt to the others.  Which workgroup contracts a unit has no influence on its result.
  const bool dedicated = D2 != nullptr && nwg >= 4 * n_tiles;
  const int mwg = dedicated ? wg - n_tiles : wg, mn = dedicated ? nwg - n_tiles : nwg;
  if (mwg >= 0 && mwg < n_units) {
    load_w(D.Wih);
    float4 g[3];
    fetch(D, mwg, g);
    for (int unit = mwg; unit < n_units; unit += mn) {
      const int tile = unit / D.n_steps, s = unit - tile * D.n_steps;
      const float4 gc[3] = {g[0], g[1], g[2]};
      if (unit + mn < n_units) fetch(D, unit + mn, g);          // the next unit's operands arrive under this unit's work
      f32x4 acc[KBW];
      contract(gc, acc);
      if (D2 == nullptr || s != D.n_steps - 1) store(D, tile, D.t_start + D.t_sign * s, acc);     // with the reverse step: below
    }
  }
  if (D2 != nullptr && wg < n_tiles) {
    // the tiles' last forward step, stored as any other unit; then ONE reload of the weights (the reverse direction's) and the reverse
    // step of the same tiles added to what this very thread stored: acc_rev + stored, the arithmetic of the former accumulate launch
    const int tl = D.t_start + D.t_sign * (D.n_steps - 1);
    if (dedicated) load_w(D.Wih);
    for (int tile = wg; tile < n_tiles; tile += nwg) {
      float4 g[3];
      f32x4 acc[KBW];
      fetch(D, tile * D.n_steps + D.n_steps - 1, g);
      contract(g, acc);
      store(D, tile, tl, acc);
    }
    load_w(D2->Wih);
    for (int tile = wg; tile < n_tiles; tile += nwg) {
      float4 g[3];
      f32x4 acc[KBW];
      fetch(*D2, tile, g);                                     // the reverse direction has one step per tile: unit = tile
      contract(g, acc);
      const int b = tile * 16 + li;
      if (active && b < a.B) {
#pragma unroll
        for (int kk = 0; kk < KBW; ++kk) {
          float* dst = D.dx + (int64_t)b * D.dx_bs + (int64_t)tl * D.dx_ts + (w * KBW + kk) * 16 + lq * 4;
          const float4 p = *(const float4*)dst;
          *(float4*)dst = make_float4(acc[kk][0] + p.x, acc[kk][1] + p.y, acc[kk][2] + p.z, acc[kk][3] + p.w);
        }
      }
    }
  }
}

// dX alone (fold batches; the reverse step of layer 1, which accumulates): grid (workgroups, directions, folds), 21.5 KB of LDS
// REV (layer 1 of the latency form, grid y = 1): the reverse direction's single step is folded in (dx2_role's D2)
template <int I, bool REV = false>
__global__ __launch_bounds__(256) void gru_bwd_dx(const GruArgs a, int n_tiles, const FoldCtx fc) {
  __shared__ __attribute__((aligned(16))) __bf16 dgp[3 * 16 * BwdDw2<I>::DXP];
  FOLD_GRU_ARGS;
  (void)agi_; (void)ax_;
  if constexpr (REV) {
    GruDir D2 = a.dir[1];
    fold_dir(D2, fc);
    dx2_role<I>(a, D, &D2, n_tiles, blockIdx.x, gridDim.x, dgp);
  } else {
    dx2_role<I>(a, D, nullptr, n_tiles, blockIdx.x, gridDim.x, dgp);
  }
}

// Layer 0, one model: grid (gdx + ndw, 2, 1): blockIdx.x < gdx -> dX of direction blockIdx.y, else dW of direction blockIdx.y.
// (Layer 1's dX is gru_bwd_dx<128, true>, its dW rides in layer 0's recurrence launch: gru_bwd4.hip.)
template <int I>
__global__ __launch_bounds__(256, 1) void gru_bwd_dxdw(const GruArgs a, const int n_tiles, const int gdx, const FoldCtx fc) {
  static_assert(I == 32, "layer 0 only");
  extern __shared__ __attribute__((aligned(16))) __bf16 dyn_lds[];
  FOLD_GRU_ARGS;
  (void)agi_;
  if ((int)blockIdx.x < gdx) dx2_role<I>(a, D, nullptr, n_tiles, blockIdx.x, gdx, dyn_lds);
  else dw2_role<I>(a, D, ax_, akey_, n_tiles, blockIdx.x - gdx, gridDim.x - gdx, dyn_lds);
}
// dW alone (the single reverse step of layer 1 behind a fused layer-1 backward): grid (ndw, ndir, folds)
template <int I>
__global__ __launch_bounds__(256, 1) void gru_bwd_dw2(const GruArgs a, const int n_tiles, const FoldCtx fc) {
  extern __shared__ __attribute__((aligned(16))) __bf16 dyn_lds[];
  FOLD_GRU_ARGS;
  (void)agi_;
  dw2_role<I>(a, D, ax_, akey_, n_tiles, blockIdx.x, gridDim.x, dyn_lds);
}

// ------------------------------------------------------------------------------------
// One-layer GRU (msig_batch.gru_layers = 1; the hierarchical experiment's second model, main.py:35-40): outputs[:, -1, :] is
// layer 0's output at the last position, and its gradient enters layer 0's backward at that position only.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void feat_from_h0_kernel(const float* __restrict__ h0, float* __restrict__ feat, int TP, const FoldCtx fc) {
  FOLD_BEGIN; FS(h0); FS(feat);
  feat[(size_t)blockIdx.x * 128 + threadIdx.x] = h0[((size_t)blockIdx.x * TP + (TP - 1)) * 128 + threadIdx.x];
}
__global__ __launch_bounds__(256) void dh0_from_dfeat_kernel(const float* __restrict__ dfeat, float* __restrict__ dh0, int TP, const FoldCtx fc) {
  FOLD_BEGIN; FS(dfeat); FS(dh0);
  float4* row = (float4*)(dh0 + (size_t)blockIdx.x * TP * 128);
  const int n4 = TP * 32;
  for (int i = threadIdx.x; i < n4 - 32; i += 256) row[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (threadIdx.x < 32) row[n4 - 32 + threadIdx.x] = ((const float4*)(dfeat + (size_t)blockIdx.x * 128))[threadIdx.x];
}

// ------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------
static void fill_dir(GruDir& g, const float* params, const int64_t* po, int layer, int dir) {
  g.Wih = params + po[MSIG_P_GRU_T(layer, dir, 0)];
  g.Whh = params + po[MSIG_P_GRU_T(layer, dir, 1)];
  g.bih = params + po[MSIG_P_GRU_T(layer, dir, 2)];
  g.bhh = params + po[MSIG_P_GRU_T(layer, dir, 3)];
}

static void setup_layer0(GruArgs& a, const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po) {
  const size_t stash_dir = (size_t)d.NT * d.TP * 16 * 64;   // float4 elements per direction
  a = GruArgs{};
  a.x = w.p<float>(MSIG_WS_P2); a.x_bs = (int64_t)d.TP * 32; a.x_ts = 32;
  a.B = d.B; a.drop_thr = 0; a.drop_key = 0; a.drop_scale = 1.f;
  for (int dir = 0; dir < 2; ++dir) {
    GruDir& g = a.dir[dir];
    fill_dir(g, b->params, po, 0, dir);
    g.t_start = dir ? d.TP - 1 : 0; g.t_sign = dir ? -1 : 1; g.n_steps = d.TP;
    g.h = w.p<float>(MSIG_WS_H0); g.h_bs = (int64_t)d.TP * 128; g.h_ts = 128; g.h_col = dir * 64;
    g.h_last = nullptr; g.hl_bs = 0; g.hl_col = 0;
    g.stash = b->training ? w.p<float4>(MSIG_WS_STASH0) + dir * stash_dir : nullptr;
  }
}

static void setup_layer1(GruArgs& a, const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po) {
  a = GruArgs{};
  a.x = w.p<float>(MSIG_WS_H0); a.x_bs = (int64_t)d.TP * 128; a.x_ts = 128;
  a.B = d.B;
  a.drop_thr = b->training ? b->dropout_thr : 0; a.drop_key = b->key_gru; a.drop_scale = drop_scale(a.drop_thr);
  {  // forward direction: all T' steps
    GruDir& g = a.dir[0];
    fill_dir(g, b->params, po, 1, 0);
    g.t_start = 0; g.t_sign = 1; g.n_steps = d.TP;
    g.h = w.p<float>(MSIG_WS_H1); g.h_bs = (int64_t)d.TP * 64; g.h_ts = 64; g.h_col = 0;
    g.h_last = w.p<float>(MSIG_WS_FEAT); g.hl_bs = 128; g.hl_col = 0;
    g.stash = b->training ? w.p<float4>(MSIG_WS_STASH1) : nullptr;
  }
  {  // reverse direction: only its first step (t = T'-1) reaches outputs[:, -1, :]
    GruDir& g = a.dir[1];
    fill_dir(g, b->params, po, 1, 1);
    g.t_start = d.TP - 1; g.t_sign = -1; g.n_steps = 1;
    g.h = w.p<float>(MSIG_WS_FEAT); g.h_bs = 128; g.h_ts = 0; g.h_col = 64;
    g.h_last = nullptr; g.hl_bs = 0; g.hl_col = 0;
    g.stash = b->training ? w.p<float4>(MSIG_WS_STASH1R) : nullptr;
  }
}

// Latency form of the layer-1 forward (bulk projection + lean recurrence) for underfilled GPUs; the
// workspace holds the gi tensor only below this tile count.  MSIG_GRU_FWD=fused|split overrides.
#define MSIG_LATENCY_TILES 192
// Grid sizes of the latency form's bulk kernels (projection, dX, dW).  Every workgroup first loads its weights (up to 98 KB) and,
// for dW, ends with a 150 KB partial row that colsum has to read again, so the number of workgroups is what one model needs
// to fill the chip — and in a fold batch (blockIdx.z = fold) the chip is shared: the per-fold cap shrinks with the fold count
// (at 15 folds: 3.3 -> see profiles/r02_multi_step_probe.log for the step time with and without this).
// dW: the number of workgroups fixes the summation order of the partials, so it must NOT depend on the fold count (a fold's
// numbers are bit-identical in a fold batch and alone): a constant number of (tile, step) units per workgroup instead.
#ifndef MSIG_DW_UNITS_PER_WG
#define MSIG_DW_UNITS_PER_WG 8
#endif
static int dw_grid(int units) {
  const int n = (units + MSIG_DW_UNITS_PER_WG - 1) / MSIG_DW_UNITS_PER_WG;
  return n < 1 ? 1 : (n < MSIG_DW_WG ? n : MSIG_DW_WG);
}
// The recurrence kernels of the latency form are one dependent chain per workgroup; two of them on one CU share its SIMDs and
// both chains stretch.  While the whole grid fits on the chip with one workgroup per CU, ask for enough dynamic LDS (never
// touched) that a second workgroup cannot become resident on the same CU.
static size_t exclusive_cu_lds(int n_wgs) { return n_wgs <= 256 ? (size_t)(96 * 1024) : 0; }
// Round 3: a single model's caps came down from 1024-2048 to 256 (projection) / 512 (dX): at the reference's B = 64 (960 units per
// direction) a workgroup then amortises its weight setup over ~4 units instead of 1 — gru_fwd_proj 27.9 / 16.0 -> 22.7 / 11.3 us,
// gru_bwd_dx 20.0 / 13.4 -> 14.3 / 11.0 us per launch (profiles/r03_bulk_kernels_bf16.log).  Units are independent: the grid
// does not touch a single bit of the results.
static int bulk_grid(int units, int cap_single, int n_folds, int n_dirs) {
  int cap = cap_single;
  if (n_folds > 1) {
    cap = 2048 / (n_folds * n_dirs);
    if (cap < 32) cap = 32;
    if (cap > cap_single) cap = cap_single;
  }
  return units < cap ? (units < 1 ? 1 : units) : cap;
}
#define MSIG_WS_LAYER0 1     // wave-specialised forward for layer 0 as well (0: gru_fwd_b3<32>)
// Kernel forms: per call (msig_batch.fwd_form / bwd_form, 0 = default).  The default is by batch size unless MSIG_GRU_FWD /
// MSIG_GRU_BWD name another one: read ONCE, at the first launch (concurrent fold threads launch while tests used to mutate the
// environment: getenv per launch was a data race), immutable afterwards.  There is no mutable process-global form any more
// (round 3's msig_set_kernel_form): two host threads may drive different models with different forms.
#define MSIG_L1_WS_TILES_DEFAULT 32
static int g_env_fwd_form = MSIG_FORM_AUTO, g_env_bwd_form = MSIG_FORM_AUTO, g_l1_ws_tiles = -1;
static std::once_flag g_form_env_once;
static void forms_from_env() {
  std::call_once(g_form_env_once, [] {
    const char* f = getenv("MSIG_GRU_FWD");
    if (f && !strcmp(f, "fused")) g_env_fwd_form = MSIG_FWD_B3;
    else if (f && !strcmp(f, "ws")) g_env_fwd_form = MSIG_FWD_WS;
    else if (f && !strcmp(f, "split")) g_env_fwd_form = MSIG_FWD_LATENCY;
    else if (f && !strcmp(f, "fp32")) g_env_fwd_form = MSIG_FWD_FP32;
    const char* l1 = getenv("MSIG_L1_WS_TILES");       // diagnostic: where layer 1 of the automatic forward form changes over
    g_l1_ws_tiles = l1 && atoi(l1) > 0 ? atoi(l1) : MSIG_L1_WS_TILES_DEFAULT;
    const char* b = getenv("MSIG_GRU_BWD");
    if (b && !strcmp(b, "split")) g_env_bwd_form = MSIG_BWD_SPLIT;
    else if (b && (!strcmp(b, "fused") || !strcmp(b, "b3"))) g_env_bwd_form = MSIG_BWD_B3;
    else if (b && !strcmp(b, "b4")) g_env_bwd_form = MSIG_BWD_B4;
    else if (b && !strcmp(b, "b5")) g_env_bwd_form = MSIG_BWD_B5;
    else if (b && !strcmp(b, "b6")) g_env_bwd_form = MSIG_BWD_B6;
  });
}
int msig_check_forms(const msig_batch* b) {
  if (b->fwd_form < 0 || b->fwd_form > MSIG_FWD_WS + 1 || b->bwd_form < 0 || b->bwd_form > MSIG_BWD_B6 + 1) return MSIG_E_FORM;
  return 0;
}
// Below MSIG_LATENCY_TILES batch tiles the forward GRU has two forms PER LAYER, and since round 5 they are the SAME arithmetic (the
// projection of gru_fwd_proj and of gru_fwd_ws's bulk waves, the chain of gru_fwd_rec and of gru_fwd_ws's chain waves: same pieces,
// same scales, same MFMA order, same stash) — bit-identical outputs (tests/test_parity_gpu.py::test_forward_forms_are_bit_identical,
// tools/form_bits_probe.py), so the choice is free to follow the LAUNCH: how many tiles it carries over all its folds.
//   layer 0 (I = 32): gru_fwd_ws always.  Its bulk waves keep up with the chain (0.62 us per step), so the kernel costs what the
//     recurrence alone costs and the projection launch and its gi round trip through HBM go away: 150 us against 146 + 12 us at
//     B = 64, against 281 + 88 us at 15 folds of 64 (profiles/r05_fwd_form_sweep.log).
//   layer 1 (I = 128): the bulk waves' 36 MFMAs per step share each SIMD's matrix pipe with the chain and the step stretches to
//     1.0 us — 241 us however few tiles — while projection + recurrence cost 166 us at 4 tiles, 213 at 24, 233 at 32, 286 at 40:
//     gru_fwd_ws from MSIG_L1_WS_TILES tiles per launch on.
// An explicit form (msig_batch.fwd_form, MSIG_GRU_FWD) still runs both layers in that form.  The BACKWARD forms do round
// differently (the fused kernels group the dW partials by tile, the latency form by unit): there the fold count the choice is made
// for stays msig_multi.form_folds, which a caller pins where a fold's bits must not depend on its companions.
#ifndef MSIG_FOLD_TILES
#define MSIG_FOLD_TILES 12
#endif

#define FWD_MIXED 64                 // internal: automatic choice below MSIG_LATENCY_TILES, per layer and per launch
static int fwd_form(const msig_batch* b, int n_tiles) {
  forms_from_env();
  const int f = b->fwd_form ? b->fwd_form - 1 : g_env_fwd_form;
  if (n_tiles >= MSIG_LATENCY_TILES)                                  // no gi region in the workspace: throughput forms only
    return (f == MSIG_FWD_FP32 || f == MSIG_FWD_B3) ? f : MSIG_FWD_WS;
  return f != MSIG_FORM_AUTO ? f : FWD_MIXED;
}

#ifdef MSIG_STAMPS
static void report_fwd_stamps(const char* tag, unsigned long long* dbg_dev, int nwg, int steps, hipStream_t st) {
  (void)hipStreamSynchronize(st);
  unsigned long long h[8 * 256];
  if (nwg > 256) nwg = 256;
  (void)hipMemcpy(h, dbg_dev, sizeof(unsigned long long) * 8 * nwg, hipMemcpyDeviceToHost);
  double acc[8] = {0};
  for (int i = 0; i < nwg; ++i) for (int j = 0; j < 8; ++j) acc[j] += (double)h[i * 8 + j] / nwg;
  fprintf(stderr, "[stamps %s fwd, cycles per step] ph0 %.0f | ph1 %.0f | ph2 %.0f | ph3 %.0f | ph4 %.0f | ph5 %.0f | ph6 %.0f  (seq: loop/init, x-MFMA, x prefetch, barrier, h-part, gates, stores; rec: acc init, barrier, reads+stores+MFMA, gates+ldsW)\n",
          tag, acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps, acc[4] / steps, acc[5] / steps, acc[6] / steps);
}
#endif

// > 64 KiB of dynamic LDS needs an opt-in attribute per kernel and per DEVICE (a process may drive several): set once per device.
static int ensure_lds_optin() {
  static std::mutex mu;
  static bool done[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= 64) return MSIG_E_SHAPE;
  std::lock_guard<std::mutex> lk(mu);
  if (done[dev]) return 0;
  const hipFuncAttribute A = hipFuncAttributeMaxDynamicSharedMemorySize;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b3<128, false>, A, BwdB3<128>::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b3<32, false>, A, BwdB3<32>::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b3<128, true>, A, BwdB3<128>::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b3<32, true>, A, BwdB3<32>::SMEM)) != hipSuccess) return (int)e;
  { const int rc = gru_bwd_b4_lds_optin(); if (rc) return rc; }
  { const int rc = gru_bwd_b6_lds_optin(); if (rc) return rc; }
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_dxdw<32>, A, BwdDw2<32>::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_dw2<128>, A, BwdDw2<128>::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_dw2<32>, A, BwdDw2<32>::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_fwd_rec<true>, A, 96 * 1024)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_fwd_rec<false>, A, 96 * 1024)) != hipSuccess) return (int)e;
  done[dev] = true;
  return 0;
}

static int bwd_form(const msig_batch* b, int n_tiles, int n_folds);
// Every form-related rejection of a call, BEFORE its first launch (api.hip forward_fc / train_step_fc): a fold batch runs the
// latency form and gru_fwd_ws only.  (Round 4 returned MSIG_E_FORM from launch_gru_fwd, after the front end of a training-mode
// call had already updated the BatchNorm running statistics.)
int msig_check_call_forms(const msig_batch* b, int n_tiles, const FoldCtx& fc) {
  const int rc = msig_check_forms(b);
  if (rc) return rc;
  const int form = fwd_form(b, n_tiles);
  if (fc.stride != 0 && form != MSIG_FWD_LATENCY && form != MSIG_FWD_WS && form != FWD_MIXED) return MSIG_E_FORM;
  return 0;
}
int launch_gru_fwd(const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po, const FoldCtx& fc, hipStream_t st) {
  GruArgs a;
  { const int rc = msig_check_forms(b); if (rc) return rc; }
  const int form = fwd_form(b, d.NT);
  const bool fp32 = form == MSIG_FWD_FP32;
  const bool folds = fc.stride != 0;                 // a fold batch (even of one fold: its arena need not be the first)
  if (folds && form != MSIG_FWD_LATENCY && form != MSIG_FWD_WS && form != FWD_MIXED) return MSIG_E_FORM;   // fold batching: latency form and gru_fwd_ws only
  const int launch_tiles = d.NT * (fc.n > 0 ? fc.n : 1);
  const bool ws0 = form == MSIG_FWD_WS || form == FWD_MIXED;
  const bool ws1 = form == MSIG_FWD_WS || (form == FWD_MIXED && launch_tiles >= g_l1_ws_tiles);
  bool latency = form == MSIG_FWD_LATENCY;           // layer 0 here, layer 1 below
  { const int rc = ensure_lds_optin(); if (rc) return rc; }
#ifdef MSIG_STAMPS
  static unsigned long long* dbg_dev = nullptr;
  if (!dbg_dev) (void)hipMalloc(&dbg_dev, 256 * 8 * sizeof(unsigned long long));
#endif
  setup_layer0(a, b, d, w, po);
  // gru_bwd_b6 recomputes W_hn h + b_hn: gru_fwd_ws then stores two stash vectors per step instead of three — but only inside a
  // fused train-step call (fc.fused_step), where THIS descriptor also resolves the backward form.  As separate calls
  // (msig_forward, then msig_backward) the two descriptors may name different forms, so the stash stays consumable by all of them.
  { const int bf = bwd_form(b, d.NT, fc.form_folds);
    a.stash_skip_hn = (fc.fused_step && form == MSIG_FWD_WS && b->gru_layers != 1 && bf == MSIG_BWD_B6) ? 1 : 0; }
#ifdef MSIG_STAMPS
  a.dbg = dbg_dev;
#endif
  if (latency) {
    a.gi = w.p<float4>(MSIG_WS_GI);
    a.gi_dir_stride = (size_t)d.NT * d.TP * 4 * 3 * 64;
    const int units = d.NT * d.TP;
    { MSIG_K("gru_fwd_proj_l0", st); gru_fwd_proj<32><<<dim3(bulk_grid(units, 256, fc.n, 2), 2, fc.n), 256, 0, st>>>(a, d.NT, fc); }
    MSIG_LAUNCH_CHECK();
    MSIG_K("gru_fwd_rec_l0", st);
    if (b->training) gru_fwd_rec<true><<<dim3(d.NT, 2, fc.n), 256, exclusive_cu_lds(d.NT * 2 * fc.n), st>>>(a, fc);
    else gru_fwd_rec<false><<<dim3(d.NT, 2, fc.n), 256, exclusive_cu_lds(d.NT * 2 * fc.n), st>>>(a, fc);
  } else if (fp32) {
    MSIG_K("gru_fwd_seq_l0", st);
    if (b->training) gru_fwd_seq<32, true><<<dim3(d.NT, 2), 256, 0, st>>>(a);
    else gru_fwd_seq<32, false><<<dim3(d.NT, 2), 256, 0, st>>>(a);
  } else if (ws0 && folds) {
    MSIG_K("gru_fwd_ws_l0", st);
    if (b->training) gru_fwd_ws<32, true, true><<<dim3(d.NT, 2, fc.n), 512, 0, st>>>(a, fc);
    else gru_fwd_ws<32, false, true><<<dim3(d.NT, 2, fc.n), 512, 0, st>>>(a, fc);
  } else if (ws0 && MSIG_WS_LAYER0) {
    MSIG_K("gru_fwd_ws_l0", st);
    if (b->training) gru_fwd_ws<32, true, false><<<dim3(d.NT, 2), 512, 0, st>>>(a, fc);
    else gru_fwd_ws<32, false, false><<<dim3(d.NT, 2), 512, 0, st>>>(a, fc);
  } else {
    MSIG_K("gru_fwd_b3_l0", st);
    if (b->training) gru_fwd_b3<32, true><<<dim3(d.NT, 2), 256, 0, st>>>(a);
    else gru_fwd_b3<32, false><<<dim3(d.NT, 2), 256, 0, st>>>(a);
  }
  MSIG_LAUNCH_CHECK();
#ifdef MSIG_STAMPS
  report_fwd_stamps("L0", dbg_dev, d.NT, d.TP, st);
#endif
  if (b->gru_layers == 1) {            // one-layer model: outputs[:, -1, :] = layer 0 at the last position
    MSIG_K("feat_from_h0", st);
    feat_from_h0_kernel<<<dim3(d.B, 1, fc.n), 128, 0, st>>>(w.p<float>(MSIG_WS_H0), w.p<float>(MSIG_WS_FEAT), d.TP, fc);
    MSIG_LAUNCH_CHECK();
    return 0;
  }
  setup_layer1(a, b, d, w, po);
#ifdef MSIG_STAMPS
  a.dbg = dbg_dev;
#endif
  latency = form == MSIG_FWD_LATENCY || (form == FWD_MIXED && !ws1);
  if (latency) {
    // few batch tiles: bulk projection over all CUs, then the lean recurrence.  The single reverse step of the
    // top layer is direction 1 of the same two launches (one unit per tile in the projection, a one-step
    // recurrence), not a third launch: at this batch size a step is bound by its number of launches.
    a.gi = w.p<float4>(MSIG_WS_GI);
    a.gi_dir_stride = (size_t)d.NT * d.TP * 4 * 3 * 64;
    const int units = d.NT * d.TP;
    { MSIG_K("gru_fwd_proj_l1", st); gru_fwd_proj<128><<<dim3(bulk_grid(units, 256, fc.n, 2), 2, fc.n), 256, 0, st>>>(a, d.NT, fc); }
    MSIG_LAUNCH_CHECK();
    {
      MSIG_K("gru_fwd_rec_l1", st);
      if (b->training) gru_fwd_rec<true><<<dim3(d.NT, 2, fc.n), 256, exclusive_cu_lds(d.NT * 2 * fc.n), st>>>(a, fc);
      else gru_fwd_rec<false><<<dim3(d.NT, 2, fc.n), 256, exclusive_cu_lds(d.NT * 2 * fc.n), st>>>(a, fc);
    }
  } else if (fp32) {
    MSIG_K("gru_fwd_seq_l1", st);
    if (b->training) gru_fwd_seq<128, true><<<dim3(d.NT, 2), 256, 0, st>>>(a);
    else gru_fwd_seq<128, false><<<dim3(d.NT, 2), 256, 0, st>>>(a);
  } else if (ws1 && folds) {
    MSIG_K("gru_fwd_ws_l1", st);
    if (b->training) gru_fwd_ws<128, true, true><<<dim3(d.NT, 2, fc.n), 512, 0, st>>>(a, fc);
    else gru_fwd_ws<128, false, true><<<dim3(d.NT, 2, fc.n), 512, 0, st>>>(a, fc);
  } else if (ws1) {
    MSIG_K("gru_fwd_ws_l1", st);
    if (b->training) gru_fwd_ws<128, true, false><<<dim3(d.NT, 2), 512, 0, st>>>(a, fc);
    else gru_fwd_ws<128, false, false><<<dim3(d.NT, 2), 512, 0, st>>>(a, fc);
  } else {
    MSIG_K("gru_fwd_b3_l1", st);
    if (b->training) gru_fwd_b3<128, true><<<dim3(d.NT, 2), 256, 0, st>>>(a);
    else gru_fwd_b3<128, false><<<dim3(d.NT, 2), 256, 0, st>>>(a);
  }
  MSIG_LAUNCH_CHECK();
#ifdef MSIG_STAMPS
  report_fwd_stamps("L1", dbg_dev, d.NT, d.TP, st);
#endif
  return 0;
}

template <int I>
static int reduce_dw(const GruDir& g, int nwg, float* grads, const int64_t* po, int layer, int dir, ColsumPlan& plan) {
  const int PS = 192 * I + 192 * 64 + 256;
  const bool ok = plan.add(g.part, nwg, PS, 0, 192 * I, grads + po[MSIG_P_GRU_T(layer, dir, 0)]) &&
                  plan.add(g.part, nwg, PS, 192 * I, 192 * 64, grads + po[MSIG_P_GRU_T(layer, dir, 1)]) &&
                  plan.add(g.part, nwg, PS, 192 * I + 192 * 64, 192, grads + po[MSIG_P_GRU_T(layer, dir, 2)]) &&              // b_ih <- dr,dz,dn
                  plan.add(g.part, nwg, PS, 192 * I + 192 * 64, 128, grads + po[MSIG_P_GRU_T(layer, dir, 3)]) &&              // b_hh[r,z] <- dr,dz
                  plan.add(g.part, nwg, PS, 192 * I + 192 * 64 + 192, 64, grads + po[MSIG_P_GRU_T(layer, dir, 3)] + 128);     // b_hh[n]   <- dhn
  return ok ? 0 : MSIG_E_SHAPE;
}

// Fused vs split backward.  The fused kernel (gru_bwd_b3) owns a batch tile for all steps with 108 (216) bf16 MFMAs per step on
// its critical path: best when every CU has a tile (B >= ~3000).  With few tiles (the reference's B = 64 is 4) the recurrence
// latency is everything, so the split form wins: a 36-MFMA-per-step recurrence (gru_bwd_seq4) and bulk dX / dW kernels that
// spread over the otherwise idle CUs.  MSIG_GRU_BWD=b3|split / msig_set_kernel_form override.  MSIG_BWD_FUSED (round 1's
// fp32-dW kernel, removed) is accepted as an alias of MSIG_BWD_B3.
enum { BWD_SPLIT = MSIG_BWD_SPLIT, BWD_B3 = MSIG_BWD_B3, BWD_B4 = MSIG_BWD_B4, BWD_B5 = MSIG_BWD_B5, BWD_B6 = MSIG_BWD_B6 };
#ifndef MSIG_BWD_DEFAULT_FUSED
#define MSIG_BWD_DEFAULT_FUSED BWD_B6      // layer 0: gru_bwd_b6 (+ gru_fwd_ws storing r, z only); layer 1: gru_bwd_b3<128> in every fused form
#endif
static int bwd_form(const msig_batch* b, int n_tiles, int n_folds) {
  forms_from_env();
  const int f = b->bwd_form ? b->bwd_form - 1 : g_env_bwd_form;
  if (f != MSIG_FORM_AUTO) return f == MSIG_BWD_SPLIT ? BWD_SPLIT : (f == MSIG_BWD_B4 ? BWD_B4 : (f == MSIG_BWD_B5 ? BWD_B5 : (f == MSIG_BWD_B6 ? BWD_B6 : BWD_B3)));
  return (n_tiles >= 192 || (n_folds > 1 && n_tiles * n_folds >= MSIG_FOLD_TILES)) ? MSIG_BWD_DEFAULT_FUSED : BWD_SPLIT;
}

int launch_gru_bwd(const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po, ColsumPlan& plan, const FoldCtx& fc, hipStream_t st) {
  GruArgs a, a1;                       // a1: layer 1's arguments, kept for its dW workgroups in layer 0's recurrence launch (latency form)
  int nwg1 = 0;
  bool dw1_pending = false;
  const PartOffsets pof = part_offsets(d);
  float* part1 = w.p<float>(MSIG_WS_GRAD_PART) + pof.l1;
  float* part0 = w.p<float>(MSIG_WS_GRAD_PART) + pof.l0;
  { const int rc = msig_check_forms(b); if (rc) return rc; }
  const int form = bwd_form(b, d.NT, fc.form_folds);
  const bool fused = form != BWD_SPLIT;
  const bool folds = fc.stride != 0;
  { const int rc = ensure_lds_optin(); if (rc) return rc; }
  const int thr = b->training ? b->dropout_thr : 0;
#ifdef MSIG_STAMPS
  static unsigned long long* dbg_dev = nullptr;
  if (!dbg_dev) (void)hipMalloc(&dbg_dev, 2 * 512 * 16 * sizeof(unsigned long long));
#endif
  const bool one_layer = b->gru_layers == 1;
  if (one_layer) {                     // dL/d outputs[:, -1, :] enters layer 0 at the last position; nothing else does
    MSIG_K("dh0_from_dfeat", st);
    dh0_from_dfeat_kernel<<<dim3(d.B, 1, fc.n), 256, 0, st>>>(w.p<float>(MSIG_WS_DFEAT), w.p<float>(MSIG_WS_DH0), d.TP, fc);
    MSIG_LAUNCH_CHECK();
  }
  // ---- layer 1 (forward direction: T' steps; reverse direction: one step) ----
  setup_layer1(a, b, d, w, po);
  const int PS1 = 192 * 128 + 192 * 64 + 256;
  const int nwg_full = pof.gru_rows;                    // each direction's sub-region holds this many partial rows
  for (int dir = 0; dir < 2; ++dir) {
    GruDir& g = a.dir[dir];
    g.dh = w.p<float>(MSIG_WS_DFEAT); g.dh_bs = 128; g.dh_ts = 0; g.dh_col = dir * 64; g.dh_mode = 1;
    g.dx = w.p<float>(MSIG_WS_DH0); g.dx_bs = (int64_t)d.TP * 128; g.dx_ts = 128; g.dx_accumulate = dir;
    g.part = part1 + (size_t)dir * nwg_full * PS1;
  }
  a.x_drop_thr = thr; a.x_drop_key = b->key_gru; a.x_drop_scale = drop_scale(thr);
  if (!one_layer) {
  if (!fused) {
    // latency form: both directions share the recurrence launch (direction 1 is a single step), and dX + dW of the layer are ONE
    // launch (gru_bwd_dxdw): the dX workgroup of a tile's last forward step adds the reverse step's dX itself
    {
      MSIG_K("gru_bwd_seq4_l1", st);
      const int rcs = launch_gru_bwd_seq4(1, a, d.NT, 2, fc, st);
      if (rcs) return rcs;
    }
    MSIG_LAUNCH_CHECK();
    const int units0 = d.NT * d.TP;
    const int nwg = dw_grid(units0);
    // dX of the layer (the reverse direction's single step folded in: the workgroup of a tile's last forward step adds it) is all
    // layer 0's recurrence waits for; the layer's dW rides in THAT launch (gru_bwd4.hip gru_bwd_seq4_dw1), beside the chains
    {
      const int gdx = bulk_grid(units0, 512, fc.n, 1);
      MSIG_K("gru_bwd_dx_l1", st);
      gru_bwd_dx<128, true><<<dim3(gdx, 1, fc.n), 256, 0, st>>>(a, d.NT, fc);
    }
    MSIG_LAUNCH_CHECK();
    a1 = a; nwg1 = nwg; dw1_pending = true;
    for (int dir = 0; dir < 2; ++dir) {
      const int units = d.NT * a.dir[dir].n_steps;          // workgroups beyond a direction's units write no partial row
      int rc = reduce_dw<128>(a.dir[dir], nwg < units ? nwg : units, b->grads, po, 1, dir, plan);
      if (rc) return rc;
    }
  } else
  for (int dir = 0; dir < 2; ++dir) {   // separate launches: the reverse step ACCUMULATES into DH0[:, T'-1]
    GruArgs one = a;
    one.dir[0] = a.dir[dir];
    const int units = d.NT * one.dir[0].n_steps;
    int nwg;
    if (fused && dir == 0) {
      nwg = d.NT < 256 ? d.NT : 256;
#ifdef MSIG_STAMPS
      one.dbg = dbg_dev;
#endif
#ifdef MSIG_B4_L1
      constexpr bool b4_l1 = true;
#else
      constexpr bool b4_l1 = false;       // layer 1 stays on gru_bwd_b3: see launch_gru_bwd_b4
#endif
      if (form == BWD_B4 && b4_l1) {
        MSIG_K("gru_bwd_b4_l1", st);
        const int rc4 = launch_gru_bwd_b4(128, folds, one, d.NT, nwg, 1, fc, st);
        if (rc4) return rc4;
      } else {
        MSIG_K("gru_bwd_b3_l1", st);
        if (folds) gru_bwd_b3<128, true><<<dim3(nwg, 1, fc.n), 256, BwdB3<128>::SMEM, st>>>(one, d.NT, fc);
        else gru_bwd_b3<128, false><<<dim3(nwg, 1), 256, BwdB3<128>::SMEM, st>>>(one, d.NT, fc);
      }
      MSIG_LAUNCH_CHECK();
#ifdef MSIG_STAMPS
      {
        (void)hipStreamSynchronize(st);
        unsigned long long h[8 * 256];
        (void)hipMemcpy(h, dbg_dev, sizeof(unsigned long long) * 8 * nwg, hipMemcpyDeviceToHost);
        double acc[8] = {0};
        for (int i = 0; i < nwg; ++i) for (int j = 0; j < 8; ++j) acc[j] += (double)h[i * 8 + j] / nwg;
        fprintf(stderr, "[stamps L1, cycles per tile (first tile of each WG), %d steps] top %.0f | dhMFMA %.0f | gates+ldsW %.0f | issue loads %.0f | dx %.0f | dW %.0f | barrier %.0f | vmcnt wait %.0f\n",
                d.TP, acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]);
      }
#endif
    } else {
      {
        MSIG_K(dir ? "gru_bwd_seq4_l1rev" : "gru_bwd_seq4_l1", st);
        const int rcs = launch_gru_bwd_seq4(1, one, d.NT, 1, fc, st);
        if (rcs) return rcs;
      }
      MSIG_LAUNCH_CHECK();
      const int gdx = bulk_grid(units, 512, fc.n, 1);
      { MSIG_K(dir ? "gru_bwd_dx_l1rev" : "gru_bwd_dx_l1", st); gru_bwd_dx<128><<<dim3(gdx, 1, fc.n), 256, 0, st>>>(one, d.NT, fc); }
      MSIG_LAUNCH_CHECK();
      nwg = dw_grid(units);
      if (nwg > units) nwg = units;
      { MSIG_K(dir ? "gru_bwd_dw_l1rev" : "gru_bwd_dw_l1", st); gru_bwd_dw2<128><<<dim3(nwg, 1, fc.n), 256, BwdDw2<128>::SMEM, st>>>(one, d.NT, fc); }
      MSIG_LAUNCH_CHECK();
    }
    int rc = reduce_dw<128>(one.dir[0], nwg, b->grads, po, 1, dir, plan);
    if (rc) return rc;
  }
  }       // !one_layer
  // ---- layer 0 (both directions, T' steps); upstream grad = DH0 with the dropout mask ----
  setup_layer0(a, b, d, w, po);
  const int PS0 = 192 * 32 + 192 * 64 + 256;
  const int thr0 = one_layer ? 0 : thr;           // no layer above: no inter-layer dropout on the upstream gradient
  a.drop_thr = thr0; a.drop_key = b->key_gru; a.drop_scale = drop_scale(thr0);
  a.x_drop_thr = 0; a.x_drop_key = 0; a.x_drop_scale = 1.f;
  for (int dir = 0; dir < 2; ++dir) {
    GruDir& g = a.dir[dir];
    g.dh = w.p<float>(MSIG_WS_DH0); g.dh_bs = (int64_t)d.TP * 128; g.dh_ts = 128; g.dh_col = dir * 64; g.dh_mode = 0;
    g.dx = w.p<float>(MSIG_WS_DX0) + (size_t)dir * d.B * d.TP * 32; g.dx_bs = (int64_t)d.TP * 32; g.dx_ts = 32;
    g.dx_accumulate = 0;
    g.part = part0 + (size_t)dir * nwg_full * PS0;
  }
  int nwg0;
  if (fused) {
    nwg0 = d.NT < 128 ? d.NT : 128;
#ifdef MSIG_STAMPS
    a.dbg = dbg_dev;
#endif
    if (form == BWD_B6) {
      MSIG_K("gru_bwd_b6_l0", st);
      const int rc = launch_gru_bwd_b6(folds, a, d.NT, nwg0, 2, fc, st);
      if (rc) return rc;
    } else if (form == BWD_B4 || form == BWD_B5) {
      MSIG_K(form == BWD_B5 ? "gru_bwd_b5_l0" : "gru_bwd_b4_l0", st);
      const int rc = launch_gru_bwd_b4(32, folds, a, d.NT, nwg0, 2, fc, st, form == BWD_B5);
      if (rc) return rc;
    } else {
      MSIG_K("gru_bwd_b3_l0", st);
      if (folds) gru_bwd_b3<32, true><<<dim3(nwg0, 2, fc.n), 256, BwdB3<32>::SMEM, st>>>(a, d.NT, fc);
      else gru_bwd_b3<32, false><<<dim3(nwg0, 2), 256, BwdB3<32>::SMEM, st>>>(a, d.NT, fc);
    }
    MSIG_LAUNCH_CHECK();
#ifdef MSIG_STAMPS
    {
      (void)hipStreamSynchronize(st);
      unsigned long long h[8 * 256];
      (void)hipMemcpy(h, dbg_dev, sizeof(unsigned long long) * 8 * 2 * nwg0, hipMemcpyDeviceToHost);
      double acc[8] = {0};
      for (int i = 0; i < 2 * nwg0; ++i) for (int j = 0; j < 8; ++j) acc[j] += (double)h[i * 8 + j] / (2 * nwg0);
      fprintf(stderr, "[stamps L0, cycles per tile, %d steps] top %.0f | dhMFMA %.0f | gates+ldsW %.0f | issue loads %.0f | dx %.0f | dW %.0f | barrier %.0f | vmcnt wait %.0f\n",
              d.TP, acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]);
    }
#endif
  } else {
#ifdef MSIG_STAMPS
    a.dbg = dbg_dev;
#endif
    {
      MSIG_K(dw1_pending ? "gru_bwd_seq4_l0+dw_l1" : "gru_bwd_seq4_l0", st);
      const int rcs = dw1_pending ? launch_gru_bwd_seq4_dw1(a, a1, d.NT, nwg1, fc, st) : launch_gru_bwd_seq4(0, a, d.NT, 2, fc, st);
      if (rcs) return rcs;
    }
    MSIG_LAUNCH_CHECK();
#ifdef MSIG_STAMPS
    {
      (void)hipStreamSynchronize(st);
      unsigned long long h[8 * 256];
      const int nw = d.NT < 256 ? d.NT : 256;
      (void)hipMemcpy(h, dbg_dev, sizeof(unsigned long long) * 8 * nw, hipMemcpyDeviceToHost);
      double acc[8] = {0};
      for (int i = 0; i < nw; ++i) for (int j = 0; j < 8; ++j) acc[j] += (double)h[i * 8 + j] / nw;
      fprintf(stderr, "[stamps L0 bwd_seq, cycles per step] loop %.0f | loads+gates+stores %.0f | ldsW %.0f | barrier %.0f | MFMA+carry %.0f\n",
              acc[0] / d.TP, acc[1] / d.TP, acc[2] / d.TP, acc[3] / d.TP, acc[4] / d.TP);
    }
#endif
    a.drop_thr = 0; a.drop_scale = 1.f;   // layer-0 input (P2) has no dropout (masks are branch-free: thr 0 == keep all, scale 1)
    const int units0 = d.NT * d.TP;
    nwg0 = dw_grid(units0);
    if (fc.n == 1) {
      const int gdx0 = bulk_grid(units0, 128, 1, 2);        // (128 dX + 120 dW) x 2 directions of 74 KB LDS: one round at two workgroups per CU
      MSIG_K("gru_bwd_dxdw_l0", st);
      gru_bwd_dxdw<32><<<dim3(gdx0 + nwg0, 2, 1), 256, BwdDw2<32>::SMEM, st>>>(a, d.NT, gdx0, fc);
    } else {
      const int gdx0 = bulk_grid(units0, 512, fc.n, 2);
      { MSIG_K("gru_bwd_dx_l0", st); gru_bwd_dx<32><<<dim3(gdx0, 2, fc.n), 256, 0, st>>>(a, d.NT, fc); }
      MSIG_LAUNCH_CHECK();
      MSIG_K("gru_bwd_dw_l0", st);
      gru_bwd_dw2<32><<<dim3(nwg0, 2, fc.n), 256, BwdDw2<32>::SMEM, st>>>(a, d.NT, fc);
    }
    MSIG_LAUNCH_CHECK();
  }
  for (int dir = 0; dir < 2; ++dir) {
    int rc = reduce_dw<32>(a.dir[dir], nwg0, b->grads, po, 0, dir, plan);
    if (rc) return rc;
  }
  return 0;
}
