// gru_bwd_b6 — the fused layer-0 GRU backward with two waves per SIMD (gru_bwd_b5's division of labour) that RECOMPUTES
// W_hn h_{t-1} + b_hn instead of reading it from the stash, so that the forward pass (gru_fwd_ws<32>) stores two stash vectors
// per step (r, z) instead of three: 1.0 GB less written there and 1.0 GB less read here per B = 8192 step (PMC: 4.53 -> 3.52, 6.06 -> 5.05).
//
// gru_bwd_b5's phase stamps (profiles/r03_b5_chain_experiments.log) say where a step's time goes: the four CHAIN waves
// (recurrence + the whole gate math) are the critical path — ~3100 cycles per step against 1728 matrix cycles per SIMD — and the
// four BULK waves (dX / dW) wait 500-670 cycles per step at the barrier.  So what moves here moves from the chain to the bulk waves:
//   * the staging of h_{t-1} (three-piece split + plane stores: 25 of the chain's 99 dh-independent operations) goes to the bulk
//     waves 6,7, which already stage x — both now THREE steps ahead, from plain global loads issued a whole iteration before their use,
//     into a ring of FOUR [x | h_prev] plane buffers (the gate-gradient planes keep their ring of two);
//   * W_hn h_{t-1} + b_hn of step s+2 is contracted by the bulk waves 4,5 during iteration s (24 MFMAs per wave, one per slot of
//     its dX stream: two 16-unit blocks each — the MFMA output layout IS the chain wave's register layout of those 16 units)
//     from the planes staged an iteration earlier, and handed to chain wave w through a lane-linear 2 x 4 KiB LDS ring for the
//     gate math of iteration s+1.  Same
//     operands, same split, same six-term order as gru_fwd_ws's chain: the recomputed vector is bit-identical to the one the
//     forward pass no longer stores.
// The chain's staged operands shrink to four pieces (r, z, h_{t-1}, upstream dh) per wave and step.
// LDS: gate-gradient planes 2 x 27 648 B | [x | h_prev] planes 4 x 9 216 B | hn ring 2 x 4 096 B | staging 2 x 16 384 B | bulk A's hn
// weight pieces 2 x 12 288 B = 157 696 B.
#include "gru_args.h"
#include "gru_bwd4.h"
#include "gru_bwd_pipe.h"

// ROLE 0: waves 0-3, CHAIN: recurrence + gate math of the wave's 16 units -> gate-gradient planes
// ROLE 1: waves 4,5, BULK A: hn of unit blocks w and w + 2, dX of one 16-column block, the three dW tiles of 32 n-gate units
// ROLE 2: waves 6,7, BULK B: the six dW tiles of the r resp. z gate, staging of x and h_prev three steps ahead
template <bool FOLDS, int ROLE>
__device__ __forceinline__ void bwd6_run(const GruArgs& a, const GruDir& D, const float* __restrict__ ax_, const uint32_t dkey, const int n_tiles) {
  using G = BwdB6;
  constexpr int I = 32;
  constexpr bool CH = ROLE == 0, BA = ROLE == 1, BB = ROLE == 2;
  constexpr int SD = G::SD, SX = G::SX, DGP = G::DGP, XHP = G::XHP, DGBUF = G::DGBUF, XHBUF = G::XHBUF, XH0 = G::XH0;
  extern __shared__ __attribute__((aligned(16))) __bf16 ring[];
  constexpr int NT = BA ? 3 : (BB ? 6 : 0), NTA = NT > 0 ? NT : 1;           // dW tiles (32 x 32) of this wave
  constexpr int NST = G::NST, SLOTB = G::SLOTB, NPIECE = 4;                 // chain: r, z, h_prev, upstream dh
  const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3, li = lane & 15, lq = lane >> 4;     // w: wave index within its group of four
  const int u0 = w * 16 + lq * 4;

  // ---- resident A operands, split once ----
  //   recurrence  A[i = li][k] = W_hh[k][w*16 + li]  (k over the 192 gate rows)     dX  A[i = li][k] = W_ih[k][w*16 + li]
  //   hn          A[i = li][k] = W_hh[128 + w*16 + li][k]  (k over the 64 state columns: gru_fwd_ws's Ah[2])
  bf16x8 AhB[CH ? 6 : 1][3], AiB[BA ? 6 : 1][3];
  if constexpr (CH || BA) {
#pragma unroll
    for (int kb = 0; kb < 6; ++kb)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 p0, p1, p2;
        if constexpr (CH) {
          split3(D.Whh[(size_t)(kb * 32 + lq * 8 + j) * 64 + w * 16 + li], p0, p1, p2);
          AhB[kb][0][j] = p0; AhB[kb][1][j] = p1; AhB[kb][2][j] = p2;
        } else {
          split3(D.Wih[(size_t)(kb * 32 + lq * 8 + j) * I + w * 16 + li], p0, p1, p2);
          AiB[kb][0][j] = p0; AiB[kb][1][j] = p1; AiB[kb][2][j] = p2;
        }
      }
#pragma unroll
    for (int kb = 0; kb < 6; ++kb)
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) { if constexpr (CH) PIN_ACC(AhB[kb][pp]); else PIN_ACC(AiB[kb][pp]); }
  }
  // hn is bulk A's job: wave w contracts unit blocks w and w + 2.  (One block on each of the four bulk waves would balance the two
  // SIMD pairs' matrix time, but bulk B — six dW tiles, the staging and its two load sets — has no registers left for the h operand:
  // 41 .. 69 spilled registers in every arrangement tried.)  Bulk A's registers are full as well (dX weights + accumulators fill the
  // AccVGPRs, fragments the arch VGPRs; resident hn weights were spilled and reloaded every step), so the pieces of W_hn live in
  // LDS — 2 waves x 2 blocks x 6 KiB behind the staging ring, lane-linear, written and read by the same wave (no barrier) — and
  // are fetched a group ahead of the MFMAs that use them.
  f32x4 b_hn[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  char* const whn = (char*)ring + G::WHN0 + (w & 1) * 12288 + lane * 16;
  if constexpr (BA) {
#pragma unroll
    for (int ub = 0; ub < 2; ++ub) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const float* wr = D.Whh + (size_t)(128 + (w + 2 * ub) * 16 + li) * 64 + kb * 32 + lq * 8;
        bf16x8 P[3];
#pragma unroll
        for (int j = 0; j < 8; ++j) { __bf16 p0, p1, p2; split3(wr[j], p0, p1, p2); P[0][j] = p0; P[1][j] = p1; P[2][j] = p2; }
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) *(bf16x8*)(whn + ((ub * 2 + kb) * 3 + pp) * 1024) = P[pp];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) b_hn[ub][e] = D.bhh[128 + (w + 2 * ub) * 16 + lq * 4 + e];
    }
  }
  // ---- persistent dW accumulators: tile t = (A block of the gate-gradient planes [dr|dz|dhn|dn], B block of [x | h_prev]) ----
  int aoff[NTA], boff[NTA];
  aoff[0] = boff[0] = 0;
  if constexpr (BA) {                      // 32 n-gate units: dW_ih <- dn . x, dW_hh <- dhn . h_prev
    aoff[0] = 192 + 32 * w; boff[0] = 0;
    aoff[1] = 128 + 32 * w; boff[1] = 32;
    aoff[2] = 128 + 32 * w; boff[2] = 64;
  } else if constexpr (BB) {               // wave 2: the r gate, wave 3: the z gate; units lo / hi x columns x, h lo, h hi
    const int gc = (w - 2) * 64;
#pragma unroll
    for (int t = 0; t < 6; ++t) { aoff[t] = gc + 32 * (t / 3); boff[t] = 32 * (t % 3); }
  }
  f32x16 accW[NTA];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[t][r] = 0.f;
  float bacc[4][4];                         // chain: bias gradients of this lane's (row, 4 units): [dr, dz, dhn, dn]
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) bacc[g][e] = 0.f;

  const int n_steps = D.n_steps, t_start = D.t_start, t_sign = D.t_sign;
  const int dthr = a.drop_thr;
  const float dscale = a.drop_scale;
  const int64_t h_bs = D.h_bs, h_ts = D.h_ts, dh_bs = D.dh_bs, dh_ts = D.dh_ts, x_bs = a.x_bs, x_ts = a.x_ts;
  const int64_t dx_bs = D.dx_bs, dx_ts = D.dx_ts;
  const int dh_col = D.dh_col;
  const int64_t hstep = (int64_t)t_sign * h_ts, ustep = (int64_t)t_sign * dh_ts;
  const int64_t dxstep = (int64_t)t_sign * dx_ts;

  // ---- per-lane LDS offsets (elements) ----
  const int sw_li = quad_swz(li);
  const int rd_row = li * SD + ((lq * 8) ^ sw_li);                       // gate-gradient planes: B[k = 8 lq + j][n = li] of a 32-column k block
  const int wr_dg = li * SD + (u0 ^ sw_li);                              // this lane's 4-unit chunk of each gate
  const int rd_h = li * SX + ((I + lq * 8) ^ sw_li);                     // [x | h_prev] planes: h_prev k block 0 (k block 1: + 32)
  int tr_dg[2], tr_xh[2];                  // transposed reads of a 32-column block (gru_bwd4.hip)
  {
    const int g2 = lane >> 5, half = (lane >> 4) & 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * g2 + 4 * h + (li >> 2), sw = ((4 - (2 * g2 + h)) & 3) * 8;
      tr_dg[h] = row * SD + ((16 * half + 4 * (li & 3)) ^ sw);
      tr_xh[h] = row * SX + ((16 * half + 4 * (li & 3)) ^ sw);
    }
  }
  // bulk B staging roles (128 threads): one float4 of the x tile (16 rows x 32), two of the h_prev tile (16 rows x 64)
  const int tb = tid & 127;
  const int sx_row = tb >> 3, sx_c4 = tb & 7;
  const int sx_off = sx_row * SX + ((4 * sx_c4) ^ quad_swz(sx_row));
  int sh_row[2], sh_c4[2], sh_off[2];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int idx = tb + 128 * v;
    sh_row[v] = idx >> 4; sh_c4[v] = idx & 15;
    sh_off[v] = sh_row[v] * SX + ((I + 4 * sh_c4[v]) ^ quad_swz(sh_row[v]));
  }
  // chain: staged operands (LDS-DMA ring, as gru_bwd_b4) and the hn ring
  char* const stg = (char*)ring + G::STG0 + lane * 16 + w * (NPIECE * 1024);                    // this lane's 16 bytes of piece 0, slot 0
  const uint32_t stg_m0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(ring) + G::STG0 + w * (NPIECE * 1024));
  char* const hnh = (char*)ring + G::HNH0 + lane * 16;                                          // + 4096 * (step & 1) + 1024 * unit block
  auto xh_buf = [](int s) { return XH0 + (s & 3) * XHBUF; };                                   // element offset of step s's [x | h_prev] planes

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int tl = t_start + t_sign * (n_steps - 1);                 // time index of processing step 0 (the last time step)
    const int b = tile * 16 + li;
    const bool valid = b < a.B;
    const int bl = valid ? b : a.B - 1;
    const float vmask = valid ? 1.0f : 0.0f;
    const float sc_u = dscale * vmask;
    const int row0 = min(tile * 16, a.B - 1);
    auto uniform = [](const void* p) -> const char* {          // a wave-uniform pointer the compiler cannot prove uniform -> SGPR pair
      const uint64_t v = (uint64_t)(uintptr_t)p;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
      return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    // ---- chain: DMA sources (uniform base stepped by the scalar unit + a per-lane byte offset fixed per tile) ----
    [[maybe_unused]] const char* sp_b = uniform(D.stash + ((size_t)((size_t)tile * n_steps + (n_steps - 1)) * 4 + w) * 4 * 64);
    [[maybe_unused]] const uint32_t sp_off = (uint32_t)lane * 16;
    [[maybe_unused]] const char* hq_b = uniform(D.h + D.h_col + (int64_t)row0 * h_bs + (int64_t)(n_steps > 1 ? tl - t_sign : tl) * h_ts);
    [[maybe_unused]] const uint32_t hq_off = (uint32_t)(((int64_t)(bl - row0) * h_bs + u0) * 4);
    [[maybe_unused]] const char* uq_b = uniform(D.dh + D.dh_col + (int64_t)row0 * dh_bs + (int64_t)tl * dh_ts);
    [[maybe_unused]] const uint32_t uq_off = (uint32_t)(((int64_t)(bl - row0) * dh_bs + u0) * 4);
    [[maybe_unused]] uint32_t ue = (uint32_t)((int64_t)bl * dh_bs + (int64_t)tl * dh_ts + dh_col + u0);      // element index of the upstream gradient (dropout mask)
    [[maybe_unused]] float4 hcur = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (CH) hcur = *(const float4*)(D.h + D.h_col + u0 + (int64_t)bl * h_bs + (int64_t)tl * h_ts);   // h_t of processing step 0
    [[maybe_unused]] float* dxq = D.dx + lq * 4 + (int64_t)b * dx_bs + (int64_t)tl * dx_ts;                  // bulk A; only dereferenced when valid
    auto dma = [&](const uint32_t voff, const char* sbase, const uint32_t lds_dst) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    };
    // piece i of the operands of processing step (n_steps - 1 - s) -> slot; the bases then move on to the next older time step
    auto load_piece = [&](int i, int s, int slot) {
      const uint32_t dst = stg_m0 + slot * SLOTB;
      if (i == 0) dma(sp_off, sp_b, dst);
      if (i == 1) { dma(sp_off, sp_b + 64 * 16, dst + 1024); if (s > 0) sp_b -= 4 * 4 * 64 * 16; }
      if (i == 2) { dma(hq_off, hq_b, dst + 2048); if (s > 1) hq_b -= hstep * 4; }
      if (i == 3) { dma(uq_off, uq_b, dst + 3072); if (s > 0) uq_b -= ustep * 4; }
    };
    auto issue_loads = [&](int s, int slot) {
#pragma unroll
      for (int i = 0; i < NPIECE; ++i) load_piece(i, s, slot);
    };
    struct Staged { float4 r4, z4, hn4, hp4, up4; uint32_t ue; float hkeep; };
    int cons_left = n_steps;               // steps not yet consumed: the LAST one (time step 0) has h_{-1} = 0
    int cons_step = 0;                     // processing step whose operands are read next
    auto read_staged = [&](Staged& L, int slot) {
      const char* q = stg + slot * SLOTB;
      L.r4 = *(const float4*)q; L.z4 = *(const float4*)(q + 1024); L.hp4 = *(const float4*)(q + 2048); L.up4 = *(const float4*)(q + 3072);
      L.hn4 = *(const float4*)(hnh + 4096 * (cons_step & 1) + 1024 * w);
      L.ue = ue; ue -= (uint32_t)ustep;
      L.hkeep = cons_left == 1 ? 0.0f : 1.0f;
      --cons_left; ++cons_step;
    };
    auto clamp0 = [](int s) { return s > 0 ? s : 0; };

    // ---- bulk B: x and h_prev of a processing step, global -> registers (clamped past the sequence's end: staged, never used) ----
    // The steady-state loads are inline asm: the compiler SINKS an ordinary load of the (restrict, read-only) input to its use an
    // iteration later — first version of this kernel: the x load was issued right in front of its first use, 1475 cycles per step in
    // bulk B's first phase — and an asm it cannot move.  It cannot count it either: the wait is by hand (xh_wait), and the loads
    // still in flight when the loop ends are drained before their registers can be reused.
    struct XH { f32x4 x, h[2]; };
    auto gload = [](const float* p) -> f32x4 { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; };
    const int sxb = min(tile * 16 + sx_row, a.B - 1);
    const float* sx_p = ax_ + (int64_t)sxb * x_bs + 4 * sx_c4;
    const float* sh_p[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) sh_p[v] = D.h + D.h_col + (int64_t)min(tile * 16 + sh_row[v], a.B - 1) * h_bs + 4 * sh_c4[v];
    auto load_xh = [&](XH& R, int s, auto asyncc) {     // s: processing step
      constexpr bool ASYNC = decltype(asyncc)::value;
      const int sc = s < n_steps ? s : n_steps - 1, t = tl - t_sign * sc, tp = sc < n_steps - 1 ? t - t_sign : t;
      if constexpr (ASYNC) {
        R.x = gload(sx_p + (int64_t)t * x_ts);
#pragma unroll
        for (int v = 0; v < 2; ++v) R.h[v] = gload(sh_p[v] + (int64_t)tp * h_ts);
      } else {
        R.x = *(const f32x4*)(sx_p + (int64_t)t * x_ts);
#pragma unroll
        for (int v = 0; v < 2; ++v) R.h[v] = *(const f32x4*)(sh_p[v] + (int64_t)tp * h_ts);
      }
    };
    auto xh_wait = [](XH& R) {             // R's three loads have landed (the three issued after them may still be in flight)
      asm volatile("s_waitcnt vmcnt(3)" : "+v"(R.x), "+v"(R.h[0]), "+v"(R.h[1]) :: "memory");
    };
    auto zero_last = [&](XH& R, int s) {   // h_{-1} = 0: the last processing step has no previous state (prologue form)
      if (s >= n_steps - 1) { R.h[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; R.h[1] = R.h[0]; }
    };
    float hk = 1.0f;                       // steady state: the same as a factor applied where the values are consumed
    auto stage_now = [&](const XH& R, int s) {        // prologue form: split and store in one go
      const int xb = xh_buf(s);
      bf16x4 p[3];
      split3_quad(R.x, p);
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) *(bf16x4*)&ring[xb + pp * XHP + sx_off] = p[pp];
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        split3_quad(R.h[v], p);
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) *(bf16x4*)&ring[xb + pp * XHP + sh_off[v]] = p[pp];
      }
    };
    // ---- bulk: W_hn h_prev + b_hn of processing step s from its planes -> hn ring (gru_fwd_ws's contraction, term for term) ----
    auto hn_block = [&](int s) {                     // prologue form (bulk A): reads, 2 x 12 MFMAs, stores
      const __bf16* pb = ring + xh_buf(s) + rd_h;
      bf16x8 ho[2][3], Al[2][2][3];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) {
          ho[kb][pp] = *(const bf16x8*)&pb[pp * XHP + kb * 32];
#pragma unroll
          for (int ub = 0; ub < 2; ++ub) Al[ub][kb][pp] = *(const bf16x8*)(whn + ((ub * 2 + kb) * 3 + pp) * 1024);
        }
#pragma unroll
      for (int ub = 0; ub < 2; ++ub) {
        f32x4 acc = b_hn[ub];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) acc = mfma_bf16x3<CT_FWD_REC>(Al[ub][kb], ho[kb], acc);
        *(float4*)(hnh + 4096 * (s & 1) + 1024 * (w + 2 * ub)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      }
    };

    // ================= the chain's gate math as two queues of single operations (gru_bwd4.hip) =================
    float cN[4], cZ[4], cR[4], rv[4], zv[4], upm[4] = {0.f, 0.f, 0.f, 0.f}, hpv[4], t0[4], t1[4], t2[4], t3[4], dhv[4];
    float dhz[4] = {0.f, 0.f, 0.f, 0.f};
    float dgv[4][4];                       // [0 dr, 1 dz, 2 dhn, 3 dn][e] — plane column order
    SplitPair spg[2], spx[3][2];
    uint32_t wd_u = 0;
    f32x4 dh_next = {0.f, 0.f, 0.f, 0.f};
    // Q1 (chain): what does not depend on dh — C (coefficients, 15 stages x 4 elements), U (dropout mask of the upstream gradient)
    constexpr int NC_ = 60, NU_ = 14, NQ1 = NC_ + NU_;
    auto q1 = [&](auto kc, Staged& L) {
      constexpr int K = decltype(kc)::value;
      if constexpr (K < NC_) {
        constexpr int S = K / 4, e = K % 4;
        const float r_ = f4e<e>(L.r4), z_ = f4e<e>(L.z4), hh_ = f4e<e>(L.hn4), hc_ = f4e<e>(hcur);
        if constexpr (S == 0) { hpv[e] = f4e<e>(L.hp4) * L.hkeep; PINV(hpv[e]); }                         // h_{-1} = 0
        if constexpr (S == 1) { t0[e] = 1.0f - z_; PINV(t0[e]); }                                        // omz
        if constexpr (S == 2) { t1[e] = __builtin_fmaf(-z_, hpv[e], hc_); PINV(t1[e]); }                 // h_t - z h_{t-1}
        if constexpr (S == 3) { t2[e] = __builtin_fmaxf(t0[e], 1e-30f); PINV(t2[e]); }
        if constexpr (S == 4) { t2[e] = __builtin_amdgcn_rcpf(t2[e]); PINV(t2[e]); }
        if constexpr (S == 5) { t3[e] = 1.0f - r_; PINV(t3[e]); }
        if constexpr (S == 6) { t3[e] = r_ * t3[e]; PINV(t3[e]); }
        if constexpr (S == 7) { cR[e] = hh_ * t3[e]; PINV(cR[e]); }                                      // dr = dn * (W_hn h + b_hn) r (1 - r)
        if constexpr (S == 8) { t3[e] = z_ * t0[e]; PINV(t3[e]); }                                       // z (1 - z)
        if constexpr (S == 9) { t1[e] = t1[e] * t2[e]; PINV(t1[e]); }
        if constexpr (S == 10) { t1[e] = __builtin_amdgcn_fmed3f(t1[e], -1.0f, 1.0f); PINV(t1[e]); }     // n_t recovered from h (gru_n_from_h)
        if constexpr (S == 11) { t2[e] = __builtin_fmaf(-t1[e], t1[e], 1.0f); PINV(t2[e]); }             // 1 - n^2
        if constexpr (S == 12) { cN[e] = t0[e] * t2[e]; PINV(cN[e]); }                                   // dn = dh (1 - z)(1 - n^2)
        if constexpr (S == 13) { t2[e] = hpv[e] - t1[e]; PINV(t2[e]); }
        if constexpr (S == 14) { cZ[e] = t2[e] * t3[e]; rv[e] = r_; zv[e] = z_; PINV(cZ[e]); }           // dz = dh (h_{t-1} - n) z (1 - z)
      } else {
        constexpr int S = K - NC_;
        if constexpr (S == 0) wd_u = (L.ue >> 2) ^ dkey;         // fmix32((elem >> 2) ^ key), one statement per slot
        if constexpr (S == 1) wd_u ^= wd_u >> 16;
        if constexpr (S == 2) wd_u *= 0x85EBCA6Bu;
        if constexpr (S == 3) wd_u ^= wd_u >> 13;
        if constexpr (S == 4) wd_u *= 0xC2B2AE35u;
        if constexpr (S == 5) wd_u ^= wd_u >> 16;
        if constexpr (S < 6) PINV(wd_u);
        if constexpr (S >= 6 && S < 10) { upm[S - 6] = drop_mul(wd_u, S - 6, dthr, sc_u); PINV(upm[S - 6]); }
        if constexpr (S >= 10) { upm[S - 10] = upm[S - 10] * f4e<S - 10>(L.up4); PINV(upm[S - 10]); }
      }
    };
    // Q2 (chain): what depends on dh — DM (6 stages x 4 elements), BA (bias sums), then per gate: split (22) + 3 plane stores
    constexpr int NDM = 24, NBA = 16, NG1 = 2 * SPLIT_STAGES + 3, NQ2 = NDM + NBA + 4 * NG1;
    auto q2 = [&](auto kc, const int nb) {
      constexpr int K = decltype(kc)::value;
      if constexpr (K < NDM) {
        constexpr int S = K / 4, e = K % 4;
        if constexpr (S == 0) { dhv[e] = dh_next[e] + upm[e]; PINV(dhv[e]); }
        if constexpr (S == 1) { dhz[e] = dhv[e] * zv[e]; PINV(dhz[e]); }
        if constexpr (S == 2) { dgv[3][e] = dhv[e] * cN[e]; PINV(dgv[3][e]); }
        if constexpr (S == 3) { dgv[1][e] = dhv[e] * cZ[e]; PINV(dgv[1][e]); }
        if constexpr (S == 4) { dgv[0][e] = dgv[3][e] * cR[e]; PINV(dgv[0][e]); }
        if constexpr (S == 5) { dgv[2][e] = dgv[3][e] * rv[e]; PINV(dgv[2][e]); }
      } else if constexpr (K < NDM + NBA) {
        constexpr int g = (K - NDM) / 4, e = (K - NDM) % 4;
        bacc[g][e] += dgv[g][e];
        PINV(bacc[g][e]);
      } else {
        constexpr int gi = (K - NDM - NBA) / NG1, S = (K - NDM - NBA) % NG1;
        constexpr int g = gi == 0 ? 1 : (gi == 1 ? 3 : (gi == 2 ? 0 : 2));       // dz, dn first (ready first), then dr, dhn
        if constexpr (S < 2 * SPLIT_STAGES) {
          constexpr int st = S / 2, p = S % 2;
          if constexpr (st == 0) { spg[p].a = dgv[g][2 * p]; spg[p].b = dgv[g][2 * p + 1]; }
          split_stage<st>(spg[p]);
        } else {
          constexpr int pp = S - 2 * SPLIT_STAGES;
          *(uint2*)&ring[nb + wr_dg + pp * DGP + g * 64] = make_uint2(spg[0].P[pp], spg[1].P[pp]);
        }
      }
    };
    // QX (bulk B): split + plane stores of the three staged float4 (x, h_prev lo, h_prev hi) of XH R into plane buffer xb
    constexpr int NX1 = 2 * SPLIT_STAGES + 3, NQX = 3 * NX1;
    auto qx = [&](auto kc, const XH& R, const int xb) {
      constexpr int K = decltype(kc)::value, v = K / NX1, S0 = K % NX1;
      if constexpr (S0 < 2 * SPLIT_STAGES) {
        constexpr int st = S0 / 2, p = S0 % 2;
        if constexpr (st == 0) {
          const f32x4& q = v == 0 ? R.x : R.h[v == 0 ? 0 : v - 1];
          if constexpr (v == 0) { spx[v][p].a = q[2 * p]; spx[v][p].b = q[2 * p + 1]; }
          else { spx[v][p].a = q[2 * p] * hk; spx[v][p].b = q[2 * p + 1] * hk; }      // h_{-1} = 0 (x * 0 with x finite: h is a GRU state)
        }
        split_stage<st>(spx[v][p]);
      } else {
        constexpr int pp = S0 - 2 * SPLIT_STAGES;
        const int off = v == 0 ? sx_off : sh_off[v == 0 ? 0 : v - 1];
        *(uint2*)&ring[xb + off + pp * XHP] = make_uint2(spx[v][0].P[pp], spx[v][1].P[pp]);
      }
    };

    int cur = 0, nxt = DGBUF;              // gate-gradient plane buffers (element offsets)
    STAMP_DECL;
    Staged L;                              // chain: the operands of the step whose gate gradients are computed next
    XH RA, RB;                             // bulk B: x / h_prev of the steps staged next, two register sets: the loads of step
                                           // j+4 are issued at the top of iteration j and consumed in the MFMA gaps of iteration j+1
    int slot_c = 0, steps_issued = 0;
    // ================= prologue =================
    if constexpr (CH) {
#pragma unroll
      for (int q = 0; q < NST; ++q) { issue_loads(clamp0(n_steps - 1 - q), q); ++steps_issued; }
    }
    if constexpr (BB) {                    // planes of processing steps 0, 1, 2; the loads of step 3 stay in flight
#pragma unroll
      for (int s = 0; s < 3; ++s) { load_xh(RA, s, std::false_type{}); zero_last(RA, s); stage_now(RA, s); }
      FENCE();
      load_xh(RA, 3, std::true_type{});
    }
    lds_barrier();                         // A: the planes of steps 0 .. 2 are complete
    if constexpr (BA) {
      hn_block(0);
      if (n_steps > 1) hn_block(1);
    }
    lds_barrier();                         // B: hn of steps 0, 1 in the ring
    if constexpr (CH) {
      WAIT_VM((NST - 1) * NPIECE);         // step 0 has landed
      read_staged(L, 0);
      sfor<NQ1>([&](auto k) { q1(k, L); });
      hcur = L.hp4;
      sfor<NQ2>([&](auto k) { q2(k, cur); });
      dh_next = (f32x4){0.f, 0.f, 0.f, 0.f};
      FENCE();                             // every read of slot 0 is complete before the slot is refilled
      issue_loads(clamp0(n_steps - 1 - steps_issued), 0); ++steps_issued;
      slot_c = 1 % NST;
    }
    lds_barrier();                         // C: the gate-gradient planes of step 0

    // ================= one iteration.  FULL: index j (step j's planes are in `cur` / xh_buf(j)):
    //   chain: recurrence of step j -> dh of step j+1; gate math of step j+1 -> planes into `nxt`
    //   bulk : hn of step j+2 -> hn ring; dX / dW of step j; bulk B: planes of step j+3, loads of step j+4
    // !FULL (the last step): dX / dW only =================
    auto step = [&](auto fullc, const int j, XH& Rc, XH& Rl) {       // Rc: consumed (step j+3), Rl: loaded (step j+4)
      constexpr bool FULL = decltype(fullc)::value;
      STAMP(0);
      if constexpr (CH) {
        if constexpr (FULL) {
          const int s_ld = clamp0(n_steps - 1 - steps_issued);
          const __bf16* pb = ring + cur + rd_row;
          bf16x8 q[6][3];
          auto rd_rec = [&](auto kbc) {
            constexpr int kb = decltype(kbc)::value;
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) q[kb][pp] = *(const bf16x8*)&pb[pp * DGP + kb * 32];          // columns [dr|dz|dhn] = 0..191
          };
          WAIT_VM((NST - 1) * NPIECE);
          read_staged(L, slot_c);
          sfor<3>(rd_rec);
          FENCE();
          constexpr int PRE = 2, RF0 = 8;
          sfor<PRE>([&](auto k) { q1(k, L); });
          FENCE();
          STAMP(1);
          f32x4 ah0 = {0.f, 0.f, 0.f, 0.f}, ah1 = {0.f, 0.f, 0.f, 0.f};
          sfor<36>([&](auto sc) {
            constexpr int s = decltype(sc)::value, kb = s / 6, t = s % 6;
            if constexpr (kb & 1) { ah1 = mf16<t, CT_BWD_REC>(AhB[kb], q[kb], ah1); PINA(ah1); } else { ah0 = mf16<t, CT_BWD_REC>(AhB[kb], q[kb], ah0); PINA(ah0); }
            FENCE();
            if constexpr (t == 0 && kb + 3 < 6) rd_rec(ic<kb + 3>{});
            // refill the consumed slot: every staged value has been touched by a pinned operation or by the pins below
            if constexpr (s == RF0 - 1) { PINV(L.r4.x); PINV(L.z4.x); PINV(L.hp4.x); PINV(L.up4.x); PINV(L.hn4.x); }
            if constexpr (s >= RF0 && s < RF0 + NPIECE) load_piece(s - RF0, s_ld, slot_c);
            if constexpr (PRE + 2 * s < NQ1) q1(ic<PRE + 2 * s>{}, L);
            if constexpr (PRE + 2 * s + 1 < NQ1) q1(ic<PRE + 2 * s + 1>{}, L);
            FENCE();
          });
          static_assert(NQ1 <= PRE + 72, "the coefficients must be complete before the dependent part starts");
#pragma unroll
          for (int e = 0; e < 4; ++e) dh_next[e] = dhz[e] + ah0[e] + ah1[e];
          hcur = L.hp4;                                     // this step's h_{t-1} is the next processed step's h_t
          ++steps_issued; slot_c = slot_c + 1 == NST ? 0 : slot_c + 1;
          FENCE();
          STAMP(2);
          sfor<NQ2>([&](auto k) { q2(k, nxt); });
          STAMP(3);
        }
        lds_barrier();
        STAMP(5);
        { const int o = cur; cur = nxt; nxt = o; }
        return;
      }
      // ---------------- bulk ----------------
      if constexpr (BB && FULL) { load_xh(Rl, j + 4, std::true_type{}); hk = j + 3 >= n_steps - 1 ? 0.0f : 1.0f; FENCE(); }
      const __bf16* pb = ring + cur + rd_row;
      const int xcur = xh_buf(j);
      // transposed reads for the dW tiles.  Bulk B: B blocks x, h lo, h hi (18 reads), then two A blocks (12 reads).  Bulk A keeps
      // two B sets: x and h lo here (12 + 12 reads), h hi replaces x during tile 1.
      constexpr int NBR = BA ? 2 : 3, NFR = 6 * NBR + 12;
      bf16x8 Af[2][3], Bf[NBR][3];
      auto frag_read = [&](auto nc) {
        constexpr int n = decltype(nc)::value;
        if constexpr (n < 6 * NBR) {
          constexpr int blk = n / 6, pp = (n % 6) / 2, h = n % 2;
          put_half<h>(Bf[blk][pp], lds_tr_read4(ring + xcur + tr_xh[h] + 32 * blk + pp * XHP));
        } else if constexpr (n < NFR) {
          constexpr int m = n - 6 * NBR, blk = m / 6, pp = (m % 6) / 2, h = m % 2;
          put_half<h>(Af[blk][pp], lds_tr_read4(ring + cur + tr_dg[h] + aoff[BB ? 3 * blk : blk] + pp * DGP));
        }
      };
      [[maybe_unused]] const int xnew = xh_buf(j + 3);
      [[maybe_unused]] f32x4 ax;
      if constexpr (BB) {
        sfor<NFR>(frag_read);
        FENCE();
        if constexpr (FULL) { xh_wait(Rc); FENCE(); }
      }
      if constexpr (BA) {
        ax = (f32x4){0.f, 0.f, 0.f, 0.f};      // ONE accumulation chain (a dependent chain of this MFMA issues back to back): the AccVGPRs
                                               // are full — dX weights 72, dW accumulators 48, this and the hn accumulator 4 + 4
        bf16x8 qd[3][3];                                // operands of k block kb+2 are read under the MFMAs of kb: a k block's six MFMAs
                                                        // are 96 cycles, an LDS round trip under this kernel's load ~250
        auto rd_dx = [&](auto kbc) {                    // gate rows [r|z|n] <-> columns [dr|dz| . |dn]
          constexpr int kb = decltype(kbc)::value, col0 = kb < 4 ? kb * 32 : 192 + (kb - 4) * 32;
#pragma unroll
          for (int pp = 0; pp < 3; ++pp) qd[kb % 3][pp] = *(const bf16x8*)&pb[pp * DGP + col0];
        };
        // hn of step j+2 rides in the dX stream, one hn MFMA per slot of the first 24 (past the sequence's end: computed from stale
        // planes into a ring slot nothing reads any more): four groups of six (unit block, k block), each group's three weight pieces
        // fetched a group ahead into a ring of two.  The dW fragment reads move behind them (slots 24 .. 35, two per slot).
        // What a wave of this kernel pays for is MFMA issue — two waves of a SIMD alternating on the matrix pipe get ~25 cycles per
        // 16x16x32 MFMA, whatever the operand prefetch depth or the order of the two streams (profiles/r03_b6_experiments.log).
        bf16x8 hoq[2][3], Alq[2][3];
        f32x4 hacc = b_hn[0];
        auto rd_al = [&](auto gc) {                     // group g = 2 ub + kb: unit block w first, then w + 2
          constexpr int g = decltype(gc)::value, ub = g / 2, kb = g % 2;
#pragma unroll
          for (int pp = 0; pp < 3; ++pp) Alq[g & 1][pp] = *(const bf16x8*)(whn + ((ub * 2 + kb) * 3 + pp) * 1024);
        };
        if constexpr (FULL) {
          const __bf16* ph = ring + xh_buf(j + 2) + rd_h;
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) hoq[kb][pp] = *(const bf16x8*)&ph[pp * XHP + kb * 32];
          rd_al(ic<0>{});
        }
        rd_dx(ic<0>{});
        rd_dx(ic<1>{});
        FENCE();
        sfor<36>([&](auto sc) {
          constexpr int s = decltype(sc)::value, kb = s / 6, t = s % 6;
          if constexpr (FULL && s < 24) {
            constexpr int g = s / 6, ub = g / 2, hkb = g % 2;
            hacc = mf16<t, CT_FWD_REC>(Alq[g & 1], hoq[hkb], hacc);
            PINA(hacc);
            FENCE();
            if constexpr (t == 0 && g + 1 < 4) rd_al(ic<g + 1>{});
            if constexpr (s % 12 == 11) {          // unit block ub complete
              *(float4*)(hnh + 4096 * ((j + 2) & 1) + 1024 * (w + 2 * ub)) = make_float4(hacc[0], hacc[1], hacc[2], hacc[3]);
              hacc = b_hn[1];
            }
          }
          ax = mf16<t, CT_DX>(AiB[kb], qd[kb % 3], ax);
          PINA(ax);
          FENCE();
          if constexpr (t == 0 && kb + 2 < 6) rd_dx(ic<kb + 2>{});
          if constexpr (FULL) {
            if constexpr (s >= 24) { frag_read(ic<2 * (s - 24)>{}); frag_read(ic<2 * (s - 24) + 1>{}); }
          } else {
            if constexpr (s >= 2 && s < 2 + NFR) frag_read(ic<s - 2>{});
          }
          FENCE();
        });
        static_assert(NFR == 24, "bulk A: 24 fragment reads in the last 12 dX slots");
      }
      STAMP(3);
      sfor<6 * NT>([&](auto sc) {
        constexpr int s = decltype(sc)::value, tI = s / 6, t = s % 6;
        constexpr int ai = BA ? (tI == 0 ? 0 : 1) : tI / 3;
        constexpr int bi = BA ? (tI == 1 ? 1 : 0) : tI % 3;
        accW[tI] = mf32<t, CT_DW>(Af[ai], Bf[bi], accW[tI]);
        PINA(accW[tI]);
        FENCE();
        if constexpr (BA && tI == 1) put_half<t % 2>(Bf[0][t / 2], lds_tr_read4(ring + xcur + tr_xh[t % 2] + 64 + (t / 2) * XHP));   // h hi for tile 2
        if constexpr (BA && s == 1) {
          if (valid) *(float4*)(dxq + w * 16) = make_float4(ax[0], ax[1], ax[2], ax[3]);
          dxq -= dxstep;
        }
        if constexpr (BB && FULL) {                     // the planes of step j+3 ride in the gaps: six operations per 32x32 slot
          sfor<6>([&](auto oc) {
            constexpr int K = 6 * s + decltype(oc)::value;
            if constexpr (K < NQX) qx(ic<K>{}, Rc, xnew);
          });
        }
        FENCE();
      });
      static_assert(NQX <= 36 * 6, "not enough MFMA gaps for the staging of a step");
      STAMP(4);
      lds_barrier();
      STAMP(5);
      { const int o = cur; cur = nxt; nxt = o; }
    };
    const int n_full = n_steps - 1;
    for (int j = 0; j < n_full; j += 2) {
      step(std::true_type{}, j, RA, RB);
      if (j + 1 < n_full) step(std::true_type{}, j + 1, RB, RA);
    }
    if constexpr (BB) WAIT_VM(0);          // the loads past the last step land before their registers are reused
    step(std::false_type{}, n_full, RA, RB);       // the last step: dX / dW only (ends on a barrier: every ring is free again)
    if constexpr (CH) WAIT_VM(0);          // the clamped re-loads past the last step must not land in the next tile's slots
#ifdef MSIG_STAMPS
    constexpr int rec = ROLE;
    if (a.dbg && lane == 0 && w == (BB ? 2 : 0) && tile == (int)blockIdx.x)
      for (int i = 0; i < 8; ++i) a.dbg[(((size_t)rec * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + i] = ph_[i];
#endif
  }

  // ---- partial: [dW_ih 192*I][dW_hh 192*64][db 256 = dr,dz,dn,dhn] ----
  float* Pp = D.part + (size_t)blockIdx.x * (192 * I + 192 * 64 + 256);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ao = aoff[t], bo = boff[t];
    const int row0 = ao < 192 ? ao : ao - 64;             // W rows: [r|z] as they are; dhn (128..191) -> n rows of W_hh; dn (192..255) -> n rows of W_ih
    const bool ih = bo < I;
    float* base = ih ? Pp + (size_t)row0 * I + bo : Pp + 192 * I + (size_t)row0 * 64 + (bo - I);
    const int ld = ih ? I : 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) base[(size_t)(8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)) * ld + (lane & 31)] = accW[t][r];
  }
  // bias gradients: fold the 16 batch rows through LDS (the loop ends on a barrier).  Scratch columns are [dr|dz|dhn|dn]; the
  // partial wants [dr|dz|dn|dhn].
  float* scratch = (float*)ring;
  constexpr int RSB = 272;
  if constexpr (CH) {
#pragma unroll
    for (int g = 0; g < 4; ++g) *(float4*)&scratch[li * RSB + g * 64 + u0] = make_float4(bacc[g][0], bacc[g][1], bacc[g][2], bacc[g][3]);
  }
  lds_barrier();
  if constexpr (CH) {                      // the chain waves are threads 0..255
    float bsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) bsum += scratch[r * RSB + tid];
    Pp[192 * I + 192 * 64 + (tid < 128 ? tid : (tid < 192 ? tid + 64 : tid - 64))] = bsum;
  }
}

// waves 0-3 chain, 4,5 bulk A, 6,7 bulk B (wave-uniform branches; every side executes the same number of s_barrier)
template <bool FOLDS>
__global__ __launch_bounds__(512, 1) void gru_bwd_b6(const GruArgs a, int n_tiles, const FoldCtx fc) {
  FOLD_GRU_ARGS_IF(FOLDS);
  (void)axkey_;
  if (threadIdx.x < 256) bwd6_run<FOLDS, 0>(a, D, ax_, akey_, n_tiles);
  else if (threadIdx.x < 384) bwd6_run<FOLDS, 1>(a, D, ax_, akey_, n_tiles);
  else bwd6_run<FOLDS, 2>(a, D, ax_, akey_, n_tiles);
}

int gru_bwd_b6_lds_optin() {
  const hipFuncAttribute A = hipFuncAttributeMaxDynamicSharedMemorySize;
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b6<false>, A, BwdB6::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b6<true>, A, BwdB6::SMEM)) != hipSuccess) return (int)e;
  return 0;
}

int launch_gru_bwd_b6(bool folds, const GruArgs& a, int n_tiles, int nwg, int ndir, const FoldCtx& fc, hipStream_t st) {
  const dim3 grid(nwg, ndir, folds ? fc.n : 1);
  if (folds) gru_bwd_b6<true><<<grid, 512, BwdB6::SMEM, st>>>(a, n_tiles, fc);
  else gru_bwd_b6<false><<<grid, 512, BwdB6::SMEM, st>>>(a, n_tiles, fc);
  MSIG_LAUNCH_CHECK();
#ifdef MSIG_STAMPS
  if (a.dbg) {
    (void)hipStreamSynchronize(st);
    static unsigned long long h[3 * 2 * 256 * 8];
    const int per = ndir * nwg;
    (void)hipMemcpy(h, a.dbg, sizeof(unsigned long long) * 8 * 3 * per, hipMemcpyDeviceToHost);
    const char* names[3] = {"chain wave 0", "bulk wave 4 (hn, dX, 3 dW tiles)", "bulk wave 6 (6 dW tiles, x / h staging)"};
    for (int role = 0; role < 3; ++role) {
      double acc[8] = {0};
      for (int i = 0; i < per; ++i) for (int j = 0; j < 8; ++j) acc[j] += (double)h[((size_t)role * per + i) * 8 + j] / per;
      const double steps = a.dir[0].n_steps;
      if (role == 0)
        fprintf(stderr, "[stamps b6 %s, cycles per step (first tile)] loop top %.0f | wait + staged reads %.0f | recurrence %.0f | dependent gate math + stores %.0f | barrier %.0f\n",
                names[role], acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps, acc[5] / steps);
      else
        fprintf(stderr, "[stamps b6 %s, cycles per step (first tile)] loop top %.0f | hn + fragment reads + dX %.0f | dW %.0f | barrier %.0f\n",
                names[role], acc[0] / steps, acc[3] / steps, acc[4] / steps, acc[5] / steps);
    }
  }
#endif
  return 0;
}
