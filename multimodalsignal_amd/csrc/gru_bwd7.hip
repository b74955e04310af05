// gru_bwd_b7 — the fused LAYER-1 GRU backward (I = 128) with two waves per SIMD.
//
// gru_bwd_b5 / b6's division of labour (four chain waves: recurrence + gate math; four bulk waves: dX / dW) does not fit layer 1 as
// one workgroup: the resident state of a 16-row tile is W_hh 288 + W_ih 576 + dW accumulators 576 registers per lane-slot, 1440 of the
// CU's 2048 — at eight waves 76 registers per wave would be left where the chain needs ~130 and a bulk wave ~100.  So the tile's
// dX / dW COLUMNS are cut in two and each half gets its own workgroup (blockIdx.y = half) that runs the WHOLE chain — recurrence and
// gate math are a sixth of the matrix work; both halves compute identical gate gradients — and half of the bulk work:
//   chain waves 0-3 : W_hh 72 AccVGPRs; recurrence with the dh-independent gate math and the h_prev staging in its gaps, then the
//                     dependent gate math, the splits and the plane stores (gru_bwd_b5's chain; layer 1: no per-step upstream gradient);
//   bulk wave w     : dX of column block 4 half + w (72 AccVGPRs) and up to five of the half's 18 dW tiles (80): the A blocks
//                     [dr lo, dr hi, dz lo, dz hi, n lo, n hi] x the B blocks {x cols 64 half .. +31, .. +63, h cols 32 half .. +31};
//                     it stages its 16 x 16 piece of the half's x columns (inter-layer dropout mask, split, plane stores) one step
//                     ahead from an inline-asm global load issued a whole iteration before its use (gru_bwd6.hip).
// Price: the stash, h_prev and the first upstream gradient are read by both halves (+2 GB per B = 8192 step).
// MEASURED AND NOT THE DEFAULT (MSIG_GRU_BWD=b7 selects it; parity-green): 1.94-1.97 ms per launch against 1.79 ms for gru_bwd_b3<128>
// (profiles/r03_b7_experiment.log).  The chain waves finish a step in ~2640 cycles and wait ~1050 at the barrier for the bulk waves
// (3700: dX 1750 for 36 MFMAs, dW 1650 for 30): an LDS round trip takes ~300 cycles in this kernel, every dW tile starts on an
// s_waitcnt lgkmcnt(0), and the registers that deeper operand rings would need are not there (A blocks two tiles ahead: 7-8 spilled
// registers; fragments early in the dX stream: 19, 2.37 ms).
// Planes per step: gate gradients [dr|dz|dhn|dn] 16 x 256 (row stride 288) and [x (the half's 64 columns) | h_prev (64)] 16 x 128
// (row stride 160 = 80 dwords = 16 * 5), three bf16 pieces each, ring of two; operand staging ring of three steps.
#include "gru_args.h"
#include "gru_bwd4.h"
#include "gru_bwd_pipe.h"

// ROLE 0: chain waves; ROLE 1: bulk waves
template <bool FOLDS, int ROLE>
__device__ __forceinline__ void bwd7_run(const GruArgs& a, const GruDir& D, const float* ax_, const uint32_t xkey, const int n_tiles) {
  using G = BwdB7;
  constexpr int I = 128, XC = 64;                           // x columns of the layer / of a half
  constexpr bool CH = ROLE == 0, BK = ROLE == 1;
  constexpr int SD = G::SD, SX = G::SX, DGP = G::DGP, XHP = G::XHP, BUFE = G::BUFE;
  extern __shared__ __attribute__((aligned(16))) __bf16 ring[];     // [2][ dg: 3 pieces x 16 x SD | xh: 3 pieces x 16 x SX ]
  constexpr int NST = G::NST, SLOTB = G::SLOTB, NPIECE = 4;         // chain: r, z, W_hn h + b_hn, h_prev
  constexpr int NT = 5;                                             // dW tiles of a bulk wave (wave 3: three)
  const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3, li = lane & 15, lq = lane >> 4;
  const int hf = blockIdx.y;                                        // column half of this workgroup
  const int u0 = w * 16 + lq * 4;

  // ---- resident A operands, split once: recurrence A[i = li][k] = W_hh[k][w*16 + li];  dX A[i = li][k] = W_ih[k][(4 hf + w)*16 + li] ----
  bf16x8 AB[6][3];
#pragma unroll
  for (int kb = 0; kb < 6; ++kb) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 p0, p1, p2;
      if constexpr (CH) split3(D.Whh[(size_t)(kb * 32 + lq * 8 + j) * 64 + w * 16 + li], p0, p1, p2);
      else split3(D.Wih[(size_t)(kb * 32 + lq * 8 + j) * I + (4 * hf + w) * 16 + li], p0, p1, p2);
      AB[kb][0][j] = p0; AB[kb][1][j] = p1; AB[kb][2][j] = p2;
    }
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) PIN_ACC(AB[kb][pp]);
  }
  // ---- bulk: dW tiles.  Tiles 0..2 use B set 0, tiles 3,4 B set 1 (wave 3 has no tiles 3,4).  Plane columns: gate gradients
  //      [dr 0..63 | dz 64..127 | dhn 128..191 | dn 192..255]; [x | h_prev]: x 0..63, h_prev 64..127 ----
  int aoff[NT], boff[NT];
  {
    const int bset0 = w < 2 ? 0 : XC + 32 * hf, bset1 = 32;          // waves 0,1: x lo then x hi; wave 2: h then x hi; wave 3: h
    const int a0 = (w & 1) ? 3 : 0;                                  // first A block of the B-set-0 tiles: blocks 0-2 or 3-5
    auto acol = [](int blk, bool on_h) { return blk < 4 ? 32 * blk : (on_h ? 128 : 192) + 32 * (blk - 4); };   // n rows: dn with x, dhn with h
#pragma unroll
    for (int t = 0; t < 3; ++t) { aoff[t] = acol(a0 + t, w >= 2); boff[t] = bset0; }
    const int a1 = w == 0 ? 0 : (w == 1 ? 2 : 4);                    // B set 1 = x hi: A blocks 0,1 | 2,3 | 4,5
#pragma unroll
    for (int t = 3; t < 5; ++t) { aoff[t] = acol(a1 + (t - 3), false); boff[t] = bset1; }
  }
  const bool has45 = w < 3;                                          // wave-uniform
  f32x16 accW[NT];
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
  const int xthr = a.x_drop_thr;
  const float xscale = a.x_drop_scale;
  const int64_t h_bs = D.h_bs, h_ts = D.h_ts, x_bs = a.x_bs, x_ts = a.x_ts, dx_bs = D.dx_bs, dx_ts = D.dx_ts;
  const int64_t hstep = (int64_t)t_sign * h_ts, dxstep = (int64_t)t_sign * dx_ts;

  // ---- per-lane LDS offsets (elements, relative to a ring buffer) ----
  const int sw_li = quad_swz(li);
  const int rd_row = li * SD + ((lq * 8) ^ sw_li);                       // gate-gradient planes: B[k = 8 lq + j][n = li] of a 32-column k block
  const int wr_dg = li * SD + (u0 ^ sw_li);                              // this lane's 4-unit chunk of each gate
  const int wr_h = 3 * DGP + li * SX + ((XC + u0) ^ sw_li);              // chain: its 4 h_prev columns
  int tr_dg[2], tr_xh[2];                  // transposed reads of a 32-column block (gru_bwd4.hip)
  {
    const int g2 = lane >> 5, half = (lane >> 4) & 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * g2 + 4 * h + (li >> 2), sw = ((4 - (2 * g2 + h)) & 3) * 8;
      tr_dg[h] = row * SD + ((16 * half + 4 * (li & 3)) ^ sw);
      tr_xh[h] = 3 * DGP + row * SX + ((16 * half + 4 * (li & 3)) ^ sw);
    }
  }
  // bulk staging role (256 threads): one float4 of the half's x tile (16 rows x 64 columns)
  const int tb = tid & 255, sx_row = tb >> 4, sx_c4 = tb & 15;
  const int sx_off = 3 * DGP + sx_row * SX + ((4 * sx_c4) ^ quad_swz(sx_row));
  // chain: staged operands (LDS-DMA ring, as gru_bwd_b4)
  char* const stg = (char*)ring + G::STG0 + lane * 16 + w * (NPIECE * 1024);
  const uint32_t stg_m0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(ring) + G::STG0 + w * (NPIECE * 1024));

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int tl = t_start + t_sign * (n_steps - 1);                 // time index of processing step 0 (the last time step)
    const int b = tile * 16 + li;
    const bool valid = b < a.B;
    const int bl = valid ? b : a.B - 1;
    const float vmask = valid ? 1.0f : 0.0f;
    const int row0 = min(tile * 16, a.B - 1);
    auto uniform = [](const void* p) -> const char* {          // a wave-uniform pointer the compiler cannot prove uniform -> SGPR pair
      const uint64_t v = (uint64_t)(uintptr_t)p;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
      return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    [[maybe_unused]] const char* sp_b = uniform(D.stash + ((size_t)((size_t)tile * n_steps + (n_steps - 1)) * 4 + w) * 4 * 64);
    [[maybe_unused]] const uint32_t sp_off = (uint32_t)lane * 16;
    [[maybe_unused]] const char* hq_b = uniform(D.h + D.h_col + (int64_t)row0 * h_bs + (int64_t)(n_steps > 1 ? tl - t_sign : tl) * h_ts);
    [[maybe_unused]] const uint32_t hq_off = (uint32_t)(((int64_t)(bl - row0) * h_bs + u0) * 4);
    [[maybe_unused]] float4 hcur = make_float4(0.f, 0.f, 0.f, 0.f), up_first = hcur;
    if constexpr (CH) {
      hcur = *(const float4*)(D.h + D.h_col + u0 + (int64_t)bl * h_bs + (int64_t)tl * h_ts);          // h_t of processing step 0
      up_first = *(const float4*)(D.dh + D.dh_col + u0 + (int64_t)bl * D.dh_bs);                       // dh_mode 1: the upstream gradient enters at the first processed step only
    }
    [[maybe_unused]] float* dxq = D.dx + (4 * hf + w) * 16 + lq * 4 + (int64_t)b * dx_bs + (int64_t)tl * dx_ts;      // bulk; only dereferenced when valid
    auto dma = [&](const uint32_t voff, const char* sbase, const uint32_t lds_dst) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    };
    auto load_piece = [&](int i, int s, int slot) {
      const uint32_t dst = stg_m0 + slot * SLOTB;
      if (i == 0) dma(sp_off, sp_b, dst);
      if (i == 1) dma(sp_off, sp_b + 64 * 16, dst + 1024);
      if (i == 2) { dma(sp_off, sp_b + 192 * 16, dst + 2048); if (s > 0) sp_b -= 4 * 4 * 64 * 16; }
      if (i == 3) { dma(hq_off, hq_b, dst + 3072); if (s > 1) hq_b -= hstep * 4; }
    };
    auto issue_loads = [&](int s, int slot) {
#pragma unroll
      for (int i = 0; i < NPIECE; ++i) load_piece(i, s, slot);
    };
    struct Staged { float4 r4, z4, hn4, hp4; float hkeep; };
    int cons_left = n_steps;               // steps not yet consumed: the LAST one (time step 0) has h_{-1} = 0
    auto read_staged = [&](Staged& L, int slot) {
      const char* q = stg + slot * SLOTB;
      L.r4 = *(const float4*)q; L.z4 = *(const float4*)(q + 1024); L.hn4 = *(const float4*)(q + 2048); L.hp4 = *(const float4*)(q + 3072);
      L.hkeep = cons_left == 1 ? 0.0f : 1.0f;
      --cons_left;
    };
    auto clamp0 = [](int s) { return s > 0 ? s : 0; };

    // ---- bulk: the x piece of a processing step, global -> registers (clamped past the sequence's end: staged, never used) ----
    struct XR { f32x4 x; uint32_t xe; };
    auto gload = [](const float* p) -> f32x4 { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; };
    const int sxb = min(tile * 16 + sx_row, a.B - 1);
    const int64_t sx_e0 = (int64_t)sxb * x_bs + XC * hf + 4 * sx_c4;           // element index within x at time 0 (the dropout mask hashes it)
    auto load_x = [&](XR& R, int s, auto asyncc) {
      constexpr bool ASYNC = decltype(asyncc)::value;
      const int sc = s < n_steps ? s : n_steps - 1, t = tl - t_sign * sc;
      const int64_t e = sx_e0 + (int64_t)t * x_ts;
      R.xe = (uint32_t)e;
      if constexpr (ASYNC) R.x = gload(ax_ + e); else R.x = *(const f32x4*)(ax_ + e);
    };
    auto x_wait = [](XR& R) {               // R's load has landed (younger in the queue: at most the next step's load)
      asm volatile("s_waitcnt vmcnt(1)" : "+v"(R.x) :: "memory");
    };

    // ================= the chain's gate math as two queues of single operations (gru_bwd4.hip) =================
    float cN[4], cZ[4], cR[4], rv[4], zv[4], hpv[4], t0[4], t1[4], t2[4], t3[4], dhv[4];
    float dhz[4] = {0.f, 0.f, 0.f, 0.f};
    float dgv[4][4];                       // [0 dr, 1 dz, 2 dhn, 3 dn][e] — plane column order
    SplitPair sph[2], spg[2], spx[2];
    uint32_t wd_x = 0;
    f32x4 dh_next = {0.f, 0.f, 0.f, 0.f};
    // Q1 (chain): C (coefficients, 15 stages x 4 elements), HS (split + store of h_prev)
    constexpr int NC_ = 60, NHS = 2 * SPLIT_STAGES + 3, NQ1 = NC_ + NHS;
    auto q1 = [&](auto kc, Staged& L, const int nb) {
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
        if constexpr (S < 2 * SPLIT_STAGES) {
          constexpr int st = S / 2, p = S % 2;
          if constexpr (st == 0) { sph[p].a = hpv[2 * p]; sph[p].b = hpv[2 * p + 1]; }
          split_stage<st>(sph[p]);
        } else {
          constexpr int pp = S - 2 * SPLIT_STAGES;
          *(uint2*)&ring[nb + wr_h + pp * XHP] = make_uint2(sph[0].P[pp], sph[1].P[pp]);
        }
      }
    };
    // Q2 (chain): DM (6 stages x 4 elements), BA (bias sums), then per gate: split (22) + 3 plane stores
    constexpr int NDM = 24, NBA = 16, NG1 = 2 * SPLIT_STAGES + 3, NQ2 = NDM + NBA + 4 * NG1;
    auto q2 = [&](auto kc, const int nb) {
      constexpr int K = decltype(kc)::value;
      if constexpr (K < NDM) {
        constexpr int S = K / 4, e = K % 4;
        if constexpr (S == 0) { dhv[e] = dh_next[e]; PINV(dhv[e]); }
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
    // QX (bulk): inter-layer dropout mask (14 operations), split (22) and plane stores (3) of the staged x piece
    constexpr int NXM = 14, NQX = NXM + 2 * SPLIT_STAGES + 3;
    auto qx = [&](auto kc, XR& R, const int nb) {
      constexpr int K = decltype(kc)::value;
      if constexpr (K < NXM) {
        if constexpr (K == 0) wd_x = (R.xe >> 2) ^ xkey;           // fmix32((elem >> 2) ^ key), one statement per slot
        if constexpr (K == 1) wd_x ^= wd_x >> 16;
        if constexpr (K == 2) wd_x *= 0x85EBCA6Bu;
        if constexpr (K == 3) wd_x ^= wd_x >> 13;
        if constexpr (K == 4) wd_x *= 0xC2B2AE35u;
        if constexpr (K == 5) wd_x ^= wd_x >> 16;
        if constexpr (K < 6) PINV(wd_x);
        if constexpr (K >= 6 && K < 10) { t0[K - 6] = drop_mul(wd_x, K - 6, xthr, xscale); PINV(t0[K - 6]); }
        if constexpr (K >= 10) { R.x[K - 10] *= t0[K - 10]; PINV(R.x[K - 10]); }
      } else if constexpr (K < NXM + 2 * SPLIT_STAGES) {
        constexpr int st = (K - NXM) / 2, p = (K - NXM) % 2;
        if constexpr (st == 0) { spx[p].a = R.x[2 * p]; spx[p].b = R.x[2 * p + 1]; }
        split_stage<st>(spx[p]);
      } else {
        constexpr int pp = K - NXM - 2 * SPLIT_STAGES;
        *(uint2*)&ring[nb + sx_off + pp * XHP] = make_uint2(spx[0].P[pp], spx[1].P[pp]);
      }
    };

    int cur = 0, nxt = BUFE;
    STAMP_DECL;
    Staged L;                              // chain: the operands of the step whose gate gradients are computed next
    XR RA, RB;                             // bulk: the x piece of the steps staged next, two register sets
    int slot_c = 0, steps_issued = 0;
    // ================= prologue: planes of processing step 0 =================
    if constexpr (CH) {
#pragma unroll
      for (int q = 0; q < NST; ++q) { issue_loads(clamp0(n_steps - 1 - q), q); ++steps_issued; }
      WAIT_VM((NST - 1) * NPIECE);         // step 0 has landed
      read_staged(L, 0);
      dh_next = (f32x4){up_first.x * vmask, up_first.y * vmask, up_first.z * vmask, up_first.w * vmask};
      sfor<NQ1>([&](auto k) { q1(k, L, cur); });
      hcur = L.hp4;
      sfor<NQ2>([&](auto k) { q2(k, cur); });
      dh_next = (f32x4){0.f, 0.f, 0.f, 0.f};
      FENCE();                             // every read of slot 0 is complete before the slot is refilled
      issue_loads(clamp0(n_steps - 1 - steps_issued), 0); ++steps_issued;
      slot_c = 1 % NST;
    } else {
      load_x(RA, 0, std::false_type{});
      sfor<NQX>([&](auto k) { qx(k, RA, cur); });
      FENCE();
      load_x(RA, 1, std::true_type{});     // stays in flight until iteration 0 consumes it
    }
    lds_barrier();

    // ================= one iteration.  FULL: index j (step j's planes are in `cur`):
    //   chain: recurrence of step j -> dh of step j+1; gate math of step j+1 -> planes into `nxt`
    //   bulk : dX / dW of step j; x piece of step j+1 -> `nxt`; load of the x piece of step j+2
    // !FULL (the last step): dX / dW only =================
    auto step = [&](auto fullc, const int j, XR& Rc, XR& Rl) {
      constexpr bool FULL = decltype(fullc)::value;
      STAMP(0);
      const __bf16* pb = ring + cur + rd_row;
      if constexpr (CH) {
        if constexpr (FULL) {
          const int s_ld = clamp0(n_steps - 1 - steps_issued);
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
          constexpr int PRE = 13, RF0 = 8;
          sfor<PRE>([&](auto k) { q1(k, L, nxt); });
          FENCE();
          STAMP(1);
          f32x4 ah0 = {0.f, 0.f, 0.f, 0.f}, ah1 = {0.f, 0.f, 0.f, 0.f};
          sfor<36>([&](auto sc) {
            constexpr int s = decltype(sc)::value, kb = s / 6, t = s % 6;
            if constexpr (kb & 1) { ah1 = mf16<t>(AB[kb], q[kb], ah1); PINA(ah1); } else { ah0 = mf16<t>(AB[kb], q[kb], ah0); PINA(ah0); }
            FENCE();
            if constexpr (t == 0 && kb + 3 < 6) rd_rec(ic<kb + 3>{});
            if constexpr (s == RF0 - 1) { PINV(L.r4.x); PINV(L.z4.x); PINV(L.hn4.x); PINV(L.hp4.x); }   // the slot's reads have returned
            if constexpr (s >= RF0 && s < RF0 + NPIECE) load_piece(s - RF0, s_ld, slot_c);
            if constexpr (PRE + 2 * s < NQ1) q1(ic<PRE + 2 * s>{}, L, nxt);
            if constexpr (PRE + 2 * s + 1 < NQ1) q1(ic<PRE + 2 * s + 1>{}, L, nxt);
            FENCE();
          });
          static_assert(NQ1 <= PRE + 72, "the dh-independent part must be complete before the dependent part starts");
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
      if constexpr (FULL) { load_x(Rl, j + 2, std::true_type{}); FENCE(); }
      bf16x8 Af[2][3], Bf[2][3];
      // transposed fragment reads: B set 0, B set 1, the A block of tile 0 (18 reads, in the second half of the dX stream); the A
      // block of tile t+1 is fetched in tile t's first MFMA slot (other register set)
      auto frag_read = [&](auto nc) {
        constexpr int n = decltype(nc)::value;
        if constexpr (n < 12) {
          constexpr int bs = n / 6, pp = (n % 6) / 2, h = n % 2;
          put_half<h>(Bf[bs][pp], lds_tr_read4(ring + cur + tr_xh[h] + boff[bs == 0 ? 0 : 3] + pp * XHP));
        } else if constexpr (n < 18) {
          constexpr int m = n - 12, pp = m / 2, h = m % 2;
          put_half<h>(Af[0][pp], lds_tr_read4(ring + cur + tr_dg[h] + aoff[0] + pp * DGP));
        }
      };
      f32x4 ax = {0.f, 0.f, 0.f, 0.f};
      bf16x8 qd[3][3];                                  // operands of k block kb+2 are read under the MFMAs of kb
      auto rd_dx = [&](auto kbc) {                      // gate rows [r|z|n] <-> columns [dr|dz| . |dn]
        constexpr int kb = decltype(kbc)::value, col0 = kb < 4 ? kb * 32 : 192 + (kb - 4) * 32;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) qd[kb % 3][pp] = *(const bf16x8*)&pb[pp * DGP + col0];
      };
      rd_dx(ic<0>{});
      rd_dx(ic<1>{});
      FENCE();
      sfor<36>([&](auto sc) {
        constexpr int s = decltype(sc)::value, kb = s / 6, t = s % 6;
        ax = mf16<t>(AB[kb], qd[kb % 3], ax);
        PINA(ax);
        FENCE();
        if constexpr (t == 0 && kb + 2 < 6) rd_dx(ic<kb + 2>{});
        if constexpr (s >= 12 && s < 30) frag_read(ic<s - 12>{});       // (earlier in the stream they cost 19 spilled registers)
        FENCE();
      });
      STAMP(3);
      if constexpr (FULL) { x_wait(Rc); FENCE(); }
      sfor<6 * NT>([&](auto sc) {
        constexpr int s = decltype(sc)::value, tI = s / 6, t = s % 6;
        if (tI < 3 || has45) {
          accW[tI] = mf32<t>(Af[tI & 1], Bf[tI < 3 ? 0 : 1], accW[tI]);
          PINA(accW[tI]);
        }
        FENCE();
        if constexpr (tI + 1 < NT && t == 0) {          // the A block of the next tile into the other register set, all six reads in the tile's
                                                        // first slot: issued one per slot, the last one came back after the next tile's first
                                                        // MFMA wanted it (s_waitcnt lgkmcnt(0) at every tile: 56 cycles per 32x32x16 MFMA)
          sfor<6>([&](auto rc) {
            constexpr int r = decltype(rc)::value, pp = r / 2, h = r % 2;
            put_half<h>(Af[(tI + 1) & 1][pp], lds_tr_read4(ring + cur + tr_dg[h] + aoff[tI + 1] + pp * DGP));
          });
        }
        if constexpr (s == 1) {
          if (valid) *(float4*)dxq = make_float4(ax[0], ax[1], ax[2], ax[3]);
          dxq -= dxstep;
        }
        if constexpr (FULL) {                           // the x piece of step j+1 rides in the gaps: six operations per 32x32 slot
          sfor<6>([&](auto oc) {
            constexpr int K = 6 * s + decltype(oc)::value;
            if constexpr (K < NQX) qx(ic<K>{}, Rc, nxt);
          });
        }
        FENCE();
      });
      static_assert(NQX <= 6 * 3 * 6, "the x staging must fit the slots every bulk wave executes");
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
    WAIT_VM(0);                            // chain: the clamped re-loads past the last step; bulk: the load past the last step
    step(std::false_type{}, n_full, RA, RB);       // the last step: dX / dW only (ends on a barrier: the ring is free again)
#ifdef MSIG_STAMPS
    if (a.dbg && lane == 0 && w == 0 && hf == 0 && tile == (int)blockIdx.x)
      for (int i = 0; i < 8; ++i) a.dbg[((size_t)ROLE * gridDim.x + blockIdx.x) * 8 + i] = ph_[i];
#endif
  }

  // ---- partial: [dW_ih 192*I][dW_hh 192*64][db 256 = dr,dz,dn,dhn]; the two halves write disjoint columns of the same row ----
  float* Pp = D.part + (size_t)blockIdx.x * (192 * I + 192 * 64 + 256);
  if constexpr (BK) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t >= 3 && !has45) break;
      const int ao = aoff[t], bo = boff[t];
      const int row0 = ao < 192 ? ao : ao - 64;             // W rows: [r|z] as they are; dhn (128..191) -> n rows of W_hh; dn (192..255) -> n rows of W_ih
      const bool ih = bo < XC;
      float* base = ih ? Pp + (size_t)row0 * I + XC * hf + bo : Pp + 192 * I + (size_t)row0 * 64 + (bo - XC);
      const int ld = ih ? I : 64;
#pragma unroll
      for (int r = 0; r < 16; ++r) base[(size_t)(8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)) * ld + (lane & 31)] = accW[t][r];
    }
  }
  // bias gradients (identical in both halves: half 0 writes them): fold the 16 batch rows through LDS (the loop ends on a barrier).
  // Scratch columns are [dr|dz|dhn|dn]; the partial wants [dr|dz|dn|dhn].
  float* scratch = (float*)ring;
  constexpr int RSB = 272;
  if constexpr (CH) {
#pragma unroll
    for (int g = 0; g < 4; ++g) *(float4*)&scratch[li * RSB + g * 64 + u0] = make_float4(bacc[g][0], bacc[g][1], bacc[g][2], bacc[g][3]);
  }
  lds_barrier();
  if constexpr (CH) {                      // the chain waves are threads 0..255
    if (hf == 0) {
      float bsum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) bsum += scratch[r * RSB + tid];
      Pp[192 * I + 192 * 64 + (tid < 128 ? tid : (tid < 192 ? tid + 64 : tid - 64))] = bsum;
    }
  }
}

// grid (workgroups, 2 column halves, folds); waves 0-3 chain, 4-7 bulk (wave-uniform branch; both sides execute the same number of s_barrier)
template <bool FOLDS>
__global__ __launch_bounds__(512, 1) void gru_bwd_b7(const GruArgs a, int n_tiles, const FoldCtx fc) {
  GruDir Dv_; const float* ax_ = a.x;
  uint32_t xkey = a.x_drop_key;
  if constexpr (FOLDS) {
    Dv_ = a.dir[0]; fold_dir(Dv_, fc);
    FOLD_BEGIN; FS(ax_);
    xkey = fc.key_gru[blockIdx.z];
  }
  const GruDir& D = FOLDS ? Dv_ : a.dir[0];
  if (threadIdx.x < 256) bwd7_run<FOLDS, 0>(a, D, ax_, xkey, n_tiles);
  else bwd7_run<FOLDS, 1>(a, D, ax_, xkey, n_tiles);
}

int gru_bwd_b7_lds_optin() {
  const hipFuncAttribute A = hipFuncAttributeMaxDynamicSharedMemorySize;
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b7<false>, A, BwdB7::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b7<true>, A, BwdB7::SMEM)) != hipSuccess) return (int)e;
  return 0;
}

int launch_gru_bwd_b7(bool folds, const GruArgs& a, int n_tiles, int nwg, const FoldCtx& fc, hipStream_t st) {
  const dim3 grid(nwg, 2, folds ? fc.n : 1);
  if (folds) gru_bwd_b7<true><<<grid, 512, BwdB7::SMEM, st>>>(a, n_tiles, fc);
  else gru_bwd_b7<false><<<grid, 512, BwdB7::SMEM, st>>>(a, n_tiles, fc);
  MSIG_LAUNCH_CHECK();
#ifdef MSIG_STAMPS
  if (a.dbg) {
    (void)hipStreamSynchronize(st);
    static unsigned long long h[2 * 256 * 8];
    (void)hipMemcpy(h, a.dbg, sizeof(unsigned long long) * 8 * 2 * nwg, hipMemcpyDeviceToHost);
    for (int role = 0; role < 2; ++role) {
      double acc[8] = {0};
      for (int i = 0; i < nwg; ++i) for (int j = 0; j < 8; ++j) acc[j] += (double)h[((size_t)role * nwg + i) * 8 + j] / nwg;
      const double steps = a.dir[0].n_steps;
      if (role == 0)
        fprintf(stderr, "[stamps b7 chain wave 0, cycles per step (first tile)] loop top %.0f | wait + staged reads + 13 ops %.0f | recurrence %.0f | dependent gate math + stores %.0f | barrier %.0f\n",
                acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps, acc[5] / steps);
      else
        fprintf(stderr, "[stamps b7 bulk wave 4, cycles per step (first tile)] loop top %.0f | dX + fragment reads %.0f | dW + x staging %.0f | barrier %.0f\n",
                acc[0] / steps, acc[3] / steps, acc[4] / steps, acc[5] / steps);
    }
  }
#endif
  return 0;
}
