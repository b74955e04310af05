// gru_bwd_b4 — the fused GRU backward (BPTT recurrence, dX, dW_ih / dW_hh / db of one layer in one kernel, every contraction on
// split-bf16 MFMA) as ONE software-pipelined instruction stream per wave: the throughput form above 192 batch tiles.
//
// Round 2's gru_bwd_b3 ran a step as recurrence -> gate math -> dX / dW one after the other at one wave per SIMD: its matrix
// pipe was 42-45 % busy and 39 % of its wave cycles were issue stalls.  What a lone wave can issue in the gaps of its own MFMA
// stream was measured with hand-placed asm (tools/gen_mfma_gap_fill.py, profiles/r03_mfma_gap_fill_microbench.log): beside
// v_mfma_f32_16x16x32_bf16 two plain VALU instructions per MFMA are free (16.4 -> 16.9 cycles), beside 32x32x16 six (32.1 ->
// 32.8); one ds_read per MFMA costs ~1 cycle; v_dot2c_f32_bf16 does NOT hide (+10 cycles each: the round-2 split sequence is
// the wrong one inside an MFMA stream), neither do more than one ds_write_b64 per 32 cycles.  So this kernel
//   * contracts dW on v_mfma_f32_32x32x16_bf16: K = 16 is exactly the 16 batch rows of ONE step (no pairing of steps, a ring
//     of TWO plane buffers instead of three), output tiles of 32 gate units x 32 input columns, both operands fetched
//     column-major from the row-major planes by ds_read_b64_tr_b16 (tools/dw32_check.hip);
//   * computes everything of the next step's gate math that does not depend on dh (n recovered from h, the gate-derivative
//     coefficients, the dropout mask of the upstream gradient, the split of h_prev / x) in the gaps of the RECURRENCE MFMAs,
//     and the rest (dh -> dr, dz, dn, dhn, their three-piece split with and / sub / perm, the plane stores) in the gaps of the
//     dX / dW MFMAs of the step already in LDS — placed by hand, one sched_barrier-fenced slot per MFMA;
//   * issues its LDS operand reads and the global loads of later steps from fixed slots of the same stream.
// Layer 0 (I = 32): the four waves have two roles so that MFMA time balances: waves 0,1 = recurrence + dX of one 16-column
// block + the three dW tiles of 32 n-gate units; waves 2,3 = recurrence + the six dW tiles of the r resp. z gate.
// Layer 1 (I = 128): every wave = recurrence + dX of two column blocks + nine dW tiles (one and a half 32-column blocks).
// Plane columns: [dr | dz | dhn | dn] and [x (I) | h_prev (64)], three bf16 pieces each.  Row strides are 16 * odd dwords (the
// four rows a transposed 32-lane access touches fall into disjoint 16-bank windows) and 8-element chunks are XORed with
// g(row >> 2), g = [0,3,2,1] (row reads by ds_read_b128 and the producers' ds_write_b64 are conflict-free as well).
#include "gru_args.h"
#include "gru_bwd4.h"

#include "gru_bwd_pipe.h"
#include "gru_dw2.h"

// ROLE 0: layer-0 waves 0,1 (recurrence + dX of one 16-column block + 3 dW tiles: the n-gate units)
// ROLE 1: layer-0 waves 2,3 (recurrence + 6 dW tiles: the r resp. z gate; they also stage the x tile)
// ROLE 2: layer 1, every wave (recurrence + dX of two column blocks + 9 dW tiles)
// ROLE 3: the recurrence of the LATENCY form (few batch tiles: gru_bwd_seq4): recurrence + gate gradients only, every wave; the
//         gate gradients leave as fp32 through the stash for the bulk dX / dW kernels (gru_bwd_dx / gru_bwd_dw), exactly as
//         gru_bwd_seq leaves them.  I = 32 <=> dh_mode 0 (layer 0), I = 128 <=> dh_mode 1 (layer 1, both directions).
// gru_bwd_b5 (layer 0, 512 threads = TWO waves per SIMD with different jobs, as gru_fwd_ws does for the forward pass):
// ROLE 4: waves 0-3, the CHAIN: recurrence + the whole gate math of the wave's 16 units (ROLE 3's flow, with the planes of all four
//         gates, the h_prev planes and the bias sums, and nothing through the stash); no dX / dW
// ROLE 5: waves 4,5, BULK: dX of one 16-column block + the three dW tiles of the n gate (ROLE 0 without recurrence and gate math)
// ROLE 6: waves 6,7, BULK: the six dW tiles of the r resp. z gate; they stage the x tile (ROLE 1 without recurrence and gate math)
//         The bulk waves contract step j from `cur` while the chain waves turn step j into the planes of step j+1 in `nxt`.
template <int I, bool FOLDS, int ROLE>
__device__ __forceinline__ void bwd4_run(const GruArgs& a, const GruDir& D, const float* __restrict__ ax_, const uint32_t dkey, const uint32_t xkey,
                                         const int n_tiles, const int tile0 = blockIdx.x, const int tile_stride = gridDim.x) {
  using G = BwdB4<ROLE == 3 ? 32 : I>;      // ROLE 3 keeps gate-gradient planes only: the smaller geometry and three staging slots for both layers
  constexpr bool L1K = I == 128;
  constexpr int SD = G::SD, SX = G::SX, DGP = G::DGP, XHP = G::XHP, BUFE = G::BUFE;
  extern __shared__ __attribute__((aligned(16))) __bf16 ring[];       // [2][ dg: 3 pieces x 16 x SD | xh: 3 pieces x 16 x SX ]
  constexpr bool SEQ = ROLE == 3;
  constexpr bool CH8 = ROLE == 4, BK8 = ROLE == 5 || ROLE == 6;  // chain / bulk waves of the 8-wave form
  constexpr bool SERIAL = SEQ || CH8;                             // the dependent gate math follows the recurrence directly (no dX / dW in this wave)
  constexpr bool HAS_DX = ROLE == 0 || ROLE == 2 || ROLE == 5, HAS_X = ROLE == 1 || ROLE == 2 || ROLE == 6;
  constexpr int NDX = HAS_DX ? (L1K ? 2 : 1) : 0;                // dX column blocks (16 columns) of this wave
  constexpr int NDXA = NDX > 0 ? NDX : 1;
  constexpr int NT = (ROLE == 0 || ROLE == 5) ? 3 : ((ROLE == 1 || ROLE == 6) ? 6 : (ROLE == 2 ? 9 : 0));        // dW tiles (32 x 32) of this wave
  constexpr int NTA = NT > 0 ? NT : 1;
  constexpr int NSTORE = SEQ ? 4 : NDX;                           // global stores per step (dX blocks / the four gate-gradient vectors)
  constexpr int NXV = L1K ? 2 : 1;                                // float4 pieces of the x tile per staging thread
  constexpr int NAF = 2, NBF = ROLE == 2 ? 2 : 3;                 // fragment register sets
  const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) & 3, li = lane & 15, lq = lane >> 4;     // w: wave index within its group of four (bulk waves of the 8-wave form: 0..3 like the waves of gru_bwd_b4 whose work they do)
  const int u0 = w * 16 + lq * 4;

  // ---- resident A operands of the 16x16x32 contractions, split once: six 32-wide k blocks over the 192 gate rows [r|z|n] ----
  //   recurrence  A[i = li][k] = W_hh[k][w*16 + li]        dX  A[i = li][k] = W_ih[k][cb*16 + li]
  bf16x8 AhB[6][3], AiB[NDXA][6][3];
#pragma unroll
  for (int kb = 0; kb < 6; ++kb)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 p0, p1, p2;
      if constexpr (!BK8) {
        split3(D.Whh[(size_t)(kb * 32 + lq * 8 + j) * 64 + w * 16 + li], p0, p1, p2);
        AhB[kb][0][j] = p0; AhB[kb][1][j] = p1; AhB[kb][2][j] = p2;
      }
      if constexpr (HAS_DX) {
#pragma unroll
        for (int kk = 0; kk < NDX; ++kk) {
          const int cb = L1K ? (2 * w + kk) : w;
          split3(D.Wih[(size_t)(kb * 32 + lq * 8 + j) * I + cb * 16 + li], p0, p1, p2);
          AiB[kk][kb][0][j] = p0; AiB[kk][kb][1][j] = p1; AiB[kk][kb][2][j] = p2;
        }
      }
    }
#pragma unroll
  for (int kb = 0; kb < 6; ++kb)
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) {
      if constexpr (!BK8) PIN_ACC(AhB[kb][pp]);
      if constexpr (HAS_DX) {
#pragma unroll
        for (int kk = 0; kk < NDX; ++kk) PIN_ACC(AiB[kk][kb][pp]);
      }
    }
  // ---- persistent dW accumulators: NT tiles of 32 gate units x 32 input columns; tile t = (A block, B block) ----
  //   A block = 32 consecutive columns of the gate-gradient planes [dr|dz|dhn|dn], B block = 32 columns of [x | h_prev]
  int aoff[NTA], boff[NTA];
  aoff[0] = boff[0] = 0;
  if constexpr (ROLE == 0 || ROLE == 5) {    // 32 n-gate units: dW_ih <- dn . x, dW_hh <- dhn . h_prev
    aoff[0] = 192 + 32 * w; boff[0] = 0;
    aoff[1] = 128 + 32 * w; boff[1] = 32;
    aoff[2] = 128 + 32 * w; boff[2] = 64;
  } else if constexpr (ROLE == 1 || ROLE == 6) {   // wave 2: the r gate, wave 3: the z gate; units lo / hi x columns x, h lo, h hi
    const int gc = (w - 2) * 64;
#pragma unroll
    for (int t = 0; t < 6; ++t) { aoff[t] = gc + 32 * (t / 3); boff[t] = 32 * (t % 3); }
  } else if constexpr (ROLE == 2) {          // layer 1: one full 32-column block (6 unit blocks) + half of another (3 unit blocks)
    const int cF = w == 0 ? 0 : (w == 1 ? 64 : (w == 2 ? 96 : 160));
    const int cH = w < 2 ? 32 : 128;
    const int nF = cF < I ? 192 : 128, nH = cH < I ? 192 : 128;     // the n-gate rows pair dn with x columns, dhn with h columns
#pragma unroll
    for (int t = 0; t < 4; ++t) { aoff[t] = 32 * t; boff[t] = cF; }
    aoff[4] = nF; boff[4] = cF; aoff[5] = nF + 32; boff[5] = cF;
    if ((w & 1) == 0) { aoff[6] = 0; aoff[7] = 32; aoff[8] = 64; } else { aoff[6] = 96; aoff[7] = nH; aoff[8] = nH + 32; }
    boff[6] = boff[7] = boff[8] = cH;
  }
  f32x16 accW[NTA];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[t][r] = 0.f;
  float bacc[4][4];                           // bias gradients of this lane's (row, 4 units): [dr, dz, dhn, dn]
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) bacc[g][e] = 0.f;

  constexpr int dh_mode = L1K ? 1 : 0;
  const int n_steps = D.n_steps, t_start = D.t_start, t_sign = D.t_sign;
  const int dthr = dh_mode == 0 ? a.drop_thr : 0, xthr = a.x_drop_thr;
  const float dscale = a.drop_scale, xscale = a.x_drop_scale;
  const int64_t h_bs = D.h_bs, h_ts = D.h_ts, dh_bs = D.dh_bs, dh_ts = D.dh_ts, x_bs = a.x_bs, x_ts = a.x_ts;
  const int64_t dx_bs = D.dx_bs, dx_ts = D.dx_ts;
  const int dh_col = D.dh_col;
  const float* hbase = D.h + D.h_col + u0;
  const float* dhbase = D.dh + D.dh_col + u0;
  float* dxbase = D.dx + lq * 4;
  const int64_t hstep = (int64_t)t_sign * h_ts, ustep = (dh_mode == 0) ? (int64_t)t_sign * dh_ts : 0;
  const int64_t xstep = (int64_t)t_sign * x_ts, dxstep = (int64_t)t_sign * dx_ts;

  // ---- per-lane LDS offsets (elements, relative to a ring buffer) ----
  const int sw_li = quad_swz(li);
  const int rd_row = li * SD + ((lq * 8) ^ sw_li);                       // row reads: B[k = 8 lq + j][n = li] of a 32-column k block
  const int wr_dg = li * SD + (u0 ^ sw_li);                              // this lane's 4-unit chunk of each gate
  const int wr_h = 3 * DGP + li * SX + ((I + u0) ^ sw_li);
  // transposed reads of a 32-column block: lane 32 g + 16 half + i supplies row 8 g + 4 h + (i >> 2), columns 16 half + 4 (i & 3)
  int tr_dg[2], tr_xh[2];
  {
    const int g2 = lane >> 5, half = (lane >> 4) & 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 8 * g2 + 4 * h + (li >> 2), sw = ((4 - (2 * g2 + h)) & 3) * 8;
      tr_dg[h] = row * SD + ((16 * half + 4 * (li & 3)) ^ sw);
      tr_xh[h] = 3 * DGP + row * SX + ((16 * half + 4 * (li & 3)) ^ sw);
    }
  }
  // x tile staging: layer 0 — waves 2,3 (ROLE 1), one float4 each; layer 1 — all threads, two float4 each
  int xrow_off[NXV];
#pragma unroll
  for (int v = 0; v < NXV; ++v) {
    const int idx = L1K ? tid + 256 * v : (tid & 127), row = idx / (I / 4), c4 = idx - row * (I / 4);
    xrow_off[v] = 3 * DGP + row * SX + ((4 * c4) ^ quad_swz(row));
  }

  // ---- operand prefetch: an LDS-DMA staging ring ----
  // The kernel moves its algorithmic bytes (6.06 GB per launch for layer 0) with ONE workgroup per CU, so the bytes it keeps in
  // flight set its bandwidth: a step's loads are 22 KB per CU; with them issued one iteration ahead into two register sets
  // 29 KB in flight gave 3.3 TB/s = 1.82 ms whatever the instruction stream did (the gate math hidden in the MFMA gaps or not),
  // and four register sets gained nothing because hipcc's vmcnt bookkeeping collapses to vmcnt(1) / vmcnt(0) in the unrolled,
  // branchy loop they need (profiles/r03_bwd4_prefetch.log).  So the per-step operands (stash r, z, W_hn h + b_hn; h_{t-1};
  // upstream dh; the x piece) go global -> LDS by global_load_lds_dwordx4 from inline asm — no destination registers for the
  // compiler to copy or wait on, any depth — into NST slots behind the plane ring, and come back by ds_read_b128 at the start
  // of the iteration that consumes them, behind a COUNTED s_waitcnt vmcnt(N) (N = the DMAs and dX stores issued after the
  // consumed step's).  A lane reads back exactly the 16 bytes its own DMA wrote (wave-linear image: M0 base + 16 lane), so no
  // barrier is involved.  NST = 3: three steps (66 KB per CU) in flight.
  struct Staged {
    float4 r4, z4, hn4, hp4, up4, xv[NXV];
    uint32_t ue, xe[NXV]; float hkeep;
  };
  // Layer 1 (experimental, -DMSIG_B4_L1) loads its operands into registers instead: its scratch reloads would each wait for
  // vmcnt(0), i.e. for every DMA in flight.
  constexpr bool USE_DMA = ROLE != 2;
  constexpr int NPC = G::NPC, NST = G::NST, SLOTB = G::SLOTB;     // pieces (1 KiB per wave) per step slot, slots, bytes per slot
  constexpr int NPIECE = BK8 ? (HAS_X ? NXV : 0) : (L1K ? 4 : 5) + (HAS_X ? NXV : 0);       // DMA instructions of a step issued by this wave
  // 8-wave form: the chain waves use five of the six pieces of their block; the x pieces of bulk waves 6,7 take the sixth of blocks 0,1
  constexpr int XPIECE0 = L1K ? 4096 : 5120;                      // byte offset of the first x piece inside a wave's block
  const int stg_w = BK8 ? w - 2 : w;
  char* const stg = (char*)ring + G::STG0 + lane * 16 + stg_w * (NPC * 1024);                   // this lane's 16 bytes of piece 0, slot 0
  const uint32_t stg_m0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(ring) + G::STG0 + stg_w * (NPC * 1024));   // wave-uniform LDS byte address

  for (int tile = tile0; tile < n_tiles; tile += tile_stride) {
    // ---- per-tile pointers ----
    const int tl = t_start + t_sign * (n_steps - 1);                 // time index of the last step (processed first)
    const int b = tile * 16 + li;
    const bool valid = b < a.B;
    const int bl = valid ? b : a.B - 1;
    const float vmask = valid ? 1.0f : 0.0f;
    const float sc_u = dscale * vmask;
    const float4* sp = D.stash + ((size_t)((size_t)tile * n_steps + (n_steps - 1)) * 4 + w) * 4 * 64 + lane;
    const float* hq = hbase + (int64_t)bl * h_bs + (int64_t)(n_steps > 1 ? tl - t_sign : tl) * h_ts;   // h_{t-1} of the last step
    float4 hcur = *(const float4*)(hbase + (int64_t)bl * h_bs + (int64_t)tl * h_ts);                   // h_t of the last step
    const float* uq = dhbase + (int64_t)bl * dh_bs + (int64_t)(dh_mode == 0 ? tl : 0) * dh_ts;
    uint32_t ue = (uint32_t)((int64_t)bl * dh_bs + (int64_t)(dh_mode == 0 ? tl : 0) * dh_ts + dh_col + u0);
    const float* xq[NXV]; uint32_t xe[NXV];
#pragma unroll
    for (int v = 0; v < NXV; ++v) {
      const int idx = L1K ? tid + 256 * v : (tid & 127), row = idx / (I / 4), c4 = idx - row * (I / 4);
      const int bb = min(tile * 16 + row, a.B - 1);
      const int64_t x0 = (int64_t)bb * x_bs + (int64_t)tl * x_ts + 4 * c4;
      xq[v] = ax_ + x0;
      xe[v] = (uint32_t)x0;
    }
    float* dxq = dxbase + (int64_t)b * dx_bs + (int64_t)tl * dx_ts;       // only dereferenced when valid
    [[maybe_unused]] float4* wp = (float4*)sp;                            // ROLE 3: the stash slots of the step whose gate gradients are written
    // DMA source addressing: uniform bases of the tile's first row / this wave's stash block at the first processed step, lane offsets
    auto uniform = [](const void* p) -> const char* {          // a wave-uniform pointer the compiler cannot prove uniform -> SGPR pair
      const uint64_t v = (uint64_t)(uintptr_t)p;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
      return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    const int row0 = min(tile * 16, a.B - 1);
    const char* sp_b = uniform(D.stash + ((size_t)((size_t)tile * n_steps + (n_steps - 1)) * 4 + w) * 4 * 64);
    const uint32_t sp_off = (uint32_t)lane * 16;
    const char* hq_b = uniform(D.h + D.h_col + (int64_t)row0 * h_bs + (int64_t)(n_steps > 1 ? tl - t_sign : tl) * h_ts);
    const uint32_t hq_off = (uint32_t)(((int64_t)(bl - row0) * h_bs + u0) * 4);
    const char* uq_b = uniform(D.dh + D.dh_col + (int64_t)row0 * dh_bs + (int64_t)(dh_mode == 0 ? tl : 0) * dh_ts);
    const uint32_t uq_off = (uint32_t)(((int64_t)(bl - row0) * dh_bs + u0) * 4);
    const char* xq_b = uniform(ax_ + (int64_t)row0 * x_bs + (int64_t)tl * x_ts);
    uint32_t xq_off[NXV];
#pragma unroll
    for (int v = 0; v < NXV; ++v) {
      const int idx = L1K ? tid + 256 * v : (tid & 127), row = idx / (I / 4), c4 = idx - row * (I / 4);
      const int bb = min(tile * 16 + row, a.B - 1);
      xq_off[v] = (uint32_t)(((int64_t)(bb - row0) * x_bs + 4 * c4) * 4);
    }
    (void)sp; (void)hq; (void)xq; (void)sp_b; (void)hq_b; (void)uq_b; (void)xq_b;

    // piece i of the loads of time step s -> slot `slot` (a DMA: only ISSUES); the pointers address step s and move on to s-1
    // source address = wave-uniform 64-bit base (SGPR pair, stepped by the scalar unit) + this lane's 32-bit byte offset (fixed per tile):
    // no vector instruction per step for addressing
    auto dma = [&](const uint32_t voff, const char* sbase, const uint32_t lds_dst) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    };
    auto load_piece = [&](int i, int s, int slot) {
      const uint32_t dst = stg_m0 + slot * SLOTB;
      constexpr int X0 = BK8 ? 0 : (L1K ? 4 : 5);      // index of the first x piece
      if constexpr (!BK8) {
        if (i == 0) dma(sp_off, sp_b, dst);
        if (i == 1) dma(sp_off, sp_b + 64 * 16, dst + 1024);
        if (i == 2) { dma(sp_off, sp_b + 192 * 16, dst + 2048); if (s > 0) sp_b -= 4 * 4 * 64 * 16; }
        if (i == 3) { dma(hq_off, hq_b, dst + 3072); if (s > 1) hq_b -= hstep * 4; }
        if constexpr (!L1K) {
          if (i == 4) { dma(uq_off, uq_b, dst + 4096); if (s > 0) uq_b -= ustep * 4; }
        }
      }
      if constexpr (HAS_X) {
#pragma unroll
        for (int v = 0; v < NXV; ++v)
          if (i == X0 + v) { dma(xq_off[v], xq_b, dst + XPIECE0 + 1024 * v); }
        if (i == NPIECE - 1 && s > 0) xq_b -= xstep * 4;
      }
    };
    auto issue_loads = [&](int s, int slot) {
#pragma unroll
      for (int i = 0; i < NPIECE; ++i) load_piece(i, s, slot);
    };
    // the staged operands of the step in `slot` -> registers; the consumption-side element indices move on with it
    int cons_left = n_steps;               // steps not yet consumed: the LAST one (time step 0) has h_{-1} = 0
    auto read_staged = [&](Staged& L, int slot) {
      const char* q = stg + slot * SLOTB;
      if constexpr (!BK8) {
        L.r4 = *(const float4*)q; L.z4 = *(const float4*)(q + 1024); L.hn4 = *(const float4*)(q + 2048); L.hp4 = *(const float4*)(q + 3072);
        if constexpr (!L1K) { L.up4 = *(const float4*)(q + 4096); L.ue = ue; ue -= (uint32_t)ustep; }
      }
      if constexpr (HAS_X) {
#pragma unroll
        for (int v = 0; v < NXV; ++v) { L.xv[v] = *(const float4*)(q + XPIECE0 + 1024 * v); L.xe[v] = xe[v]; xe[v] -= (uint32_t)xstep; }
      }
      L.hkeep = cons_left == 1 ? 0.0f : 1.0f;
      --cons_left;
    };
    // register form (!USE_DMA): piece i of time step s straight into L (ordinary loads the compiler counts); pointers move on with it
    auto reg_piece = [&](Staged& L, int i, int s) {
      if (i == 0) L.r4 = sp[0];
      if (i == 1) L.z4 = sp[64];
      if (i == 2) { L.hn4 = sp[192]; if (s > 0) sp -= 4 * 4 * 64; }
      if (i == 3) { L.hp4 = *(const float4*)hq; if (s > 1) hq -= hstep; }
      if constexpr (HAS_X) {
#pragma unroll
        for (int v = 0; v < NXV; ++v)
          if (i == 4 + v) { L.xv[v] = *(const float4*)xq[v]; if (s > 0) xq[v] -= xstep; }
      }
    };
    auto begin_step_regs = [&](Staged& L) {            // consumption-side scalars of the step whose operands are in L
#pragma unroll
      for (int v = 0; v < NXV; ++v) { L.xe[v] = xe[v]; xe[v] -= (uint32_t)xstep; }
      L.hkeep = cons_left == 1 ? 0.0f : 1.0f;
      --cons_left;
    };
    auto clamp0 = [](int s) { return s > 0 ? s : 0; };

    // ================= the gate math of one step as two queues of single operations =================
    // State of the step whose gate gradients are computed next (its h_t is hcur, its loads are in L):
    float cN[4], cZ[4], cR[4], rv[4], zv[4], upm[4] = {0.f, 0.f, 0.f, 0.f}, hpv[4], t0[4], t1[4], t2[4], t3[4], dhv[4];
    float dhz[4] = {0.f, 0.f, 0.f, 0.f};
    float dgv[4][4];                       // [0 dr, 1 dz, 2 dhn, 3 dn][e] — plane column order
    SplitPair sph[2], spx[NXV][2], spg[2];
    uint32_t wd_u = 0, wd_x[NXV];
    f32x4 dh_next = {0.f, 0.f, 0.f, 0.f};
    // ---- Q1: everything that does not depend on dh.  C (coefficients, 15 stages x 4 elements), U (layer 0: dropout mask of the
    //      upstream gradient), HS (split + store of h_prev), XS (mask, split + store of the x tile pieces) ----
    constexpr int NC_ = BK8 ? 0 : 60, NU_ = (L1K || BK8) ? 0 : 14, NHS = (SEQ || BK8) ? 0 : 2 * SPLIT_STAGES + 3, NXM = L1K ? 14 : 0, NXS1 = NXM + 2 * SPLIT_STAGES + 3;
    constexpr int NXS = HAS_X ? NXV * NXS1 : 0;
    constexpr int NQ1 = NC_ + NU_ + NHS + NXS;
    auto q1 = [&](auto kc, Staged& L, const int nb) {       // nb: ring buffer (element offset) the planes of this step go to
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
        if constexpr (S == 14) { cZ[e] = t2[e] * t3[e]; rv[e] = r_; zv[e] = z_; PINV(cZ[e]); }       // dz = dh (h_{t-1} - n) z (1 - z)
      } else if constexpr (K < NC_ + NU_) {
        if constexpr (!L1K) {
          constexpr int S = K - NC_;
          // fmix32((elem >> 2) ^ key), one statement per slot; then per element: keep iff byte >= thr
          if constexpr (S == 0) wd_u = (L.ue >> 2) ^ dkey;
          if constexpr (S == 1) wd_u ^= wd_u >> 16;
          if constexpr (S == 2) wd_u *= 0x85EBCA6Bu;
          if constexpr (S == 3) wd_u ^= wd_u >> 13;
          if constexpr (S == 4) wd_u *= 0xC2B2AE35u;
          if constexpr (S == 5) wd_u ^= wd_u >> 16;
          if constexpr (S < 6) PINV(wd_u);
          if constexpr (S >= 6 && S < 10) { upm[S - 6] = drop_mul(wd_u, S - 6, dthr, sc_u); PINV(upm[S - 6]); }
          if constexpr (S >= 10) { upm[S - 10] = upm[S - 10] * f4e<S - 10>(L.up4); PINV(upm[S - 10]); }
        }
      } else if constexpr (K < NC_ + NU_ + NHS) {
        constexpr int S = K - NC_ - NU_;
        if constexpr (S < 2 * SPLIT_STAGES) {
          constexpr int st = S / 2, p = S % 2;
          if constexpr (st == 0) { sph[p].a = hpv[2 * p]; sph[p].b = hpv[2 * p + 1]; }
          split_stage<st>(sph[p]);
        } else {
          constexpr int pp = S - 2 * SPLIT_STAGES;
          *(uint2*)&ring[nb + wr_h + pp * XHP] = make_uint2(sph[0].P[pp], sph[1].P[pp]);
        }
      } else if constexpr (K < NQ1) {
        if constexpr (HAS_X) {
          constexpr int v = (K - NC_ - NU_ - NHS) / NXS1, S0 = (K - NC_ - NU_ - NHS) % NXS1;
          if constexpr (S0 < NXM) {          // layer 1: the input is the dropped layer-0 output
            if constexpr (S0 == 0) wd_x[v] = (L.xe[v] >> 2) ^ xkey;
            if constexpr (S0 == 1) wd_x[v] ^= wd_x[v] >> 16;
            if constexpr (S0 == 2) wd_x[v] *= 0x85EBCA6Bu;
            if constexpr (S0 == 3) wd_x[v] ^= wd_x[v] >> 13;
            if constexpr (S0 == 4) wd_x[v] *= 0xC2B2AE35u;
            if constexpr (S0 == 5) wd_x[v] ^= wd_x[v] >> 16;
            if constexpr (S0 < 6) PINV(wd_x[v]);
            if constexpr (S0 >= 6 && S0 < 10) { t0[S0 - 6] = drop_mul(wd_x[v], S0 - 6, xthr, xscale); PINV(t0[S0 - 6]); }   // t0 is free once C is through
            if constexpr (S0 >= 10 && S0 < 14) { f4mul<S0 - 10>(L.xv[v], t0[S0 - 10]); }
          } else if constexpr (S0 < NXM + 2 * SPLIT_STAGES) {
            constexpr int st = (S0 - NXM) / 2, p = (S0 - NXM) % 2;
            if constexpr (st == 0) { spx[v][p].a = f4e<2 * p>(L.xv[v]); spx[v][p].b = f4e<2 * p + 1>(L.xv[v]); }
            split_stage<st>(spx[v][p]);
          } else {
            constexpr int pp = S0 - NXM - 2 * SPLIT_STAGES;
            *(uint2*)&ring[nb + xrow_off[v] + pp * XHP] = make_uint2(spx[v][0].P[pp], spx[v][1].P[pp]);
          }
        }
      }
    };
    // ---- Q2: what depends on dh.  DM (6 stages x 4 elements), BA (bias sums), then per gate: split (22) + 3 plane stores ----
    //      ROLE 3: no bias sums (gru_bwd_dw makes them from the stash), no dn planes (only dX reads them); the four fp32 vectors go
    //      to the stash slots of the step, [dr, dz, dn, dhn], at the top of the NEXT iteration while its operand reads are in flight
    //      (stash_store).  Measured per launch at B = 64 (profiles/r03_bench_B64_kernels.log): stores at the end of the step, in
    //      front of the barrier, 238 / 226 us (layer 0 / 1); threaded through the next recurrence's MFMA slots 271 / 244 us
    constexpr int NDM = BK8 ? 0 : 24, NBA = (SEQ || BK8) ? 0 : 16, NG1 = 2 * SPLIT_STAGES + 3, NGS = BK8 ? 0 : (SEQ ? 3 : 4);
    constexpr int NQ2 = NDM + NBA + NGS * NG1;
    auto q2 = [&](auto kc, const int nb) {
      constexpr int K = decltype(kc)::value;
      if constexpr (K < NDM) {
        constexpr int S = K / 4, e = K % 4;
        if constexpr (S == 0) { dhv[e] = L1K ? dh_next[e] : dh_next[e] + upm[e]; PINV(dhv[e]); }
        if constexpr (S == 1) { dhz[e] = dhv[e] * zv[e]; PINV(dhz[e]); }
        if constexpr (S == 2) { dgv[3][e] = dhv[e] * cN[e]; PINV(dgv[3][e]); }
        if constexpr (S == 3) { dgv[1][e] = dhv[e] * cZ[e]; PINV(dgv[1][e]); }
        if constexpr (S == 4) { dgv[0][e] = dgv[3][e] * cR[e]; PINV(dgv[0][e]); }
        if constexpr (S == 5) { dgv[2][e] = dgv[3][e] * rv[e]; PINV(dgv[2][e]); }
      } else if constexpr (K < NDM + NBA) {
        constexpr int g = (K - NDM) / 4, e = (K - NDM) % 4;
        bacc[g][e] += dgv[g][e];
        PINV(bacc[g][e]);
      } else if constexpr (K < NQ2) {
        constexpr int gi = (K - NDM - NBA) / NG1, S = (K - NDM - NBA) % NG1;
        constexpr int g = SEQ ? (gi == 0 ? 1 : (gi == 1 ? 0 : 2))                // ROLE 3: dz, dr, dhn
                              : (gi == 0 ? 1 : (gi == 1 ? 3 : (gi == 2 ? 0 : 2)));     // dz, dn first (ready first), then dr, dhn
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

    [[maybe_unused]] auto stash_store = [&](int v) {        // ROLE 3: stash slot v <- dr, dz, dn, dhn of the step last computed
      const int g = v == 0 ? 0 : (v == 1 ? 1 : (v == 2 ? 3 : 2));
      wp[64 * v] = make_float4(dgv[g][0], dgv[g][1], dgv[g][2], dgv[g][3]);
      if (v == 3) wp -= 4 * 4 * 64;
    };
    int cur = 0, nxt = BUFE;
    STAMP_DECL;
    // ---- prologue: gate gradients of the first processed step (dh = upstream gradient only) ----
    Staged L;                              // the operands of the step whose gate gradients are computed next
    int slot_c = 0, steps_issued = 0;      // slot of the next step to consume; steps whose loads have been issued
    [[maybe_unused]] float4 up_first = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (L1K) up_first = *(const float4*)uq;          // dh_mode 1: the upstream gradient enters at the first processed step only
    if constexpr (USE_DMA) {
#pragma unroll
      for (int q = 0; q < NST; ++q) { issue_loads(clamp0(n_steps - 1 - q), q); ++steps_issued; }
      WAIT_VM((NST - 1) * NPIECE);           // step 0 has landed (the DMAs of steps 1 .. NST-1 may still be in flight)
      read_staged(L, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4 + NXV; ++i) reg_piece(L, i, n_steps - 1);
      begin_step_regs(L);
      steps_issued = 1;
    }
    {
      if constexpr (L1K) dh_next = (f32x4){up_first.x * vmask, up_first.y * vmask, up_first.z * vmask, up_first.w * vmask};
      sfor<NQ1>([&](auto k) { q1(k, L, cur); });
      if constexpr (!BK8) hcur = L.hp4;
      sfor<NQ2>([&](auto k) { q2(k, cur); });
      dh_next = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    FENCE();                               // every read of slot 0 is complete (its values have been used) before the slot is refilled
    if constexpr (USE_DMA) {
      issue_loads(clamp0(n_steps - 1 - steps_issued), 0); ++steps_issued;
    } else {
#pragma unroll
      for (int i = 0; i < 4 + NXV; ++i) reg_piece(L, i, clamp0(n_steps - 1 - steps_issued));
      ++steps_issued;
    }
    slot_c = 1 % NST;
    lds_barrier();

    // ---- one step of the pipeline.  FULL: processing index j (step j is in `cur`): recurrence of step j -> dh of step j+1,
    //      gate math of step j+1 (operands staged in slot slot_c) -> planes into `nxt`, dX / dW of step j from `cur`; the slot is
    //      refilled with the operands of step j+1+NST as soon as it has been read.
    //      !FULL (the last step): dX / dW of the step in `cur` only. ----
    auto step = [&](auto fullc, const int j) {
      constexpr bool FULL = decltype(fullc)::value;
      constexpr int R4 = ROLE == 5 ? 0 : (ROLE == 6 ? 1 : ROLE);    // the bulk waves of the 8-wave form do the dX / dW work of ROLE 0 / 1
      const int s_ld = clamp0(n_steps - 1 - steps_issued);
      const __bf16* pb = ring + cur + rd_row;
      bf16x8 Af[NAF][3], Bf[NBF][3];
      // transposed reads for the dW tiles of ROLE 0 / 1: B blocks x, h lo, h hi (18 reads), then two A blocks (12 reads)
      auto frag_read = [&](auto nc) {
        constexpr int n = decltype(nc)::value;
        if constexpr (n < 18) {
          constexpr int blk = n / 6, pp = (n % 6) / 2, h = n % 2;
          put_half<h>(Bf[blk][pp], lds_tr_read4(ring + cur + tr_xh[h] + 32 * blk + pp * XHP));
        } else if constexpr (n < 30) {
          constexpr int m = n - 18, blk = m / 6, pp = (m % 6) / 2, h = m % 2;
          put_half<h>(Af[blk][pp], lds_tr_read4(ring + cur + tr_dg[h] + aoff[R4 == 1 ? 3 * blk : blk] + pp * DGP));
        }
      };
      constexpr int PRE = 16, NR1 = (FULL && !BK8) ? PRE + 72 : 0;        // Q1 operations consumed by the recurrence phase
      constexpr int RF0 = 8;                                    // first recurrence slot that issues a refill DMA
      auto STAGED_DONE = [&]() {                                // a use of every staged register: the compiler waits for their ds_reads here
        if constexpr (!BK8) {
          PINV(L.r4.x); PINV(L.z4.x); PINV(L.hn4.x); PINV(L.hp4.x);
          if constexpr (!L1K) PINV(L.up4.x);
        }
        if constexpr (HAS_X) {
#pragma unroll
          for (int v = 0; v < NXV; ++v) PINV(L.xv[v].x);
        }
      };
      STAMP(0);
      bf16x8 q[6][3];                                           // recurrence operands; dX reuses the [dr|dz] blocks (kb 0..3)
      if constexpr (BK8) {
        // bulk waves of the 8-wave form: no recurrence.  ROLE 6 fetches the x piece of step j+1 (staged in slot slot_c) and refills the slot
        if constexpr (FULL && HAS_X) {
          WAIT_VM((NST - 1) * NPIECE);
          read_staged(L, slot_c);
          FENCE();
        }
        if constexpr (ROLE == 6) { sfor<30>(frag_read); FENCE(); }
        if constexpr (FULL && HAS_X) {
          STAGED_DONE();
#pragma unroll
          for (int i = 0; i < NPIECE; ++i) load_piece(i, s_ld, slot_c);
          ++steps_issued; slot_c = slot_c + 1 == NST ? 0 : slot_c + 1;
          FENCE();
        }
      } else if constexpr (FULL) {
        // ---------------- R: recurrence ----------------
        auto rd_rec = [&](auto kbc) {
          constexpr int kb = decltype(kbc)::value;
#pragma unroll
          for (int pp = 0; pp < 3; ++pp) q[kb][pp] = *(const bf16x8*)&pb[pp * DGP + kb * 32];          // columns [dr|dz|dhn] = 0..191
        };
        // the step consumed now was issued NST steps of DMAs ago; younger in the queue: (NST - 1) steps of DMAs and, for the
        // waves that store, the NSTORE stores of each of the last min(j, NST) iterations (conservatively none while j < NST)
        if constexpr (USE_DMA) {
          if (NSTORE > 0 && j >= NST) WAIT_VM((NST - 1) * NPIECE + NST * NSTORE); else WAIT_VM((NST - 1) * NPIECE);
          read_staged(L, slot_c);
        } else {
          begin_step_regs(L);
        }
        sfor<3>(rd_rec);
        FENCE();
        if constexpr (SEQ) {               // the gate gradients of the step computed last iteration leave while the operand reads are in flight
#pragma unroll
          for (int v = 0; v < 4; ++v) stash_store(v);
          FENCE();
        }
        sfor<PRE>([&](auto k) { q1(k, L, nxt); });
        FENCE();
        STAMP(1);
        f32x4 ah0 = {0.f, 0.f, 0.f, 0.f}, ah1 = {0.f, 0.f, 0.f, 0.f};
        sfor<36>([&](auto sc) {
          constexpr int s = decltype(sc)::value, kb = s / 6, t = s % 6;
          if constexpr (kb & 1) { ah1 = mf16<t, CT_BWD_REC>(AhB[kb], q[kb], ah1); PINA(ah1); } else { ah0 = mf16<t, CT_BWD_REC>(AhB[kb], q[kb], ah0); PINA(ah0); }
          FENCE();
          if constexpr (t == 0 && kb + 3 < 6) rd_rec(ic<kb + 3>{});
          if constexpr (ROLE == 1 && s >= 3 && s < 33) frag_read(ic<s - 3>{});        // they read `cur`, complete since the barrier
          // refill the consumed slot: its reads were issued at the top of the phase and have returned (every staged value has been
          // touched by a pinned operation or by STAGED_DONE below before the first DMA is issued)
          if constexpr (USE_DMA) {
            if constexpr (s == RF0 - 1) { STAGED_DONE(); }
            if constexpr (s >= RF0 && s < RF0 + NPIECE) load_piece(s - RF0, s_ld, slot_c);
          }
          q1(ic<PRE + 2 * s>{}, L, nxt);
          q1(ic<PRE + 2 * s + 1>{}, L, nxt);
          FENCE();
        });
        static_assert(NC_ + NU_ <= PRE + 72, "the coefficients must be complete before the dependent part starts");
#pragma unroll
        for (int e = 0; e < 4; ++e) dh_next[e] = dhz[e] + ah0[e] + ah1[e];
        hcur = L.hp4;                                     // this step's h_{t-1} is the next processed step's h_t
        if constexpr (USE_DMA) { ++steps_issued; slot_c = slot_c + 1 == NST ? 0 : slot_c + 1; }
        FENCE();
        STAMP(2);
      } else {
        if constexpr (ROLE == 1) { sfor<30>(frag_read); FENCE(); }
      }
      if constexpr (SERIAL) {
        // latency form / chain waves: nothing of this wave's to hide the dependent part behind (dX / dW are bulk kernels on the other
        // CUs resp. the work of the bulk waves beside it): it follows the recurrence directly — gate gradients, their split, the plane
        // stores (ROLE 3: and the four stash vectors); then what is left of Q1 (chain waves: the tail of the h_prev staging)
        if constexpr (FULL) {
          sfor<NQ2>([&](auto k) { q2(k, nxt); });
          if constexpr (NQ1 > NR1) sfor<NQ1 - NR1>([&](auto k) { q1(ic<NR1 + decltype(k)::value>{}, L, nxt); });
          STAMP(3);
        }
        if constexpr (FULL || CH8) {
          lds_barrier();
          STAMP(5);
          { const int o = cur; cur = nxt; nxt = o; }
        }
        return;
      }
      // ---------------- G: dX / dW of step j, with the dependent gate math of step j+1 in the gaps ----------------
      // gop(k): k-th operation of the phase: Q2 first, then the rest of Q1
      auto gop = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        if constexpr (FULL) {
          if constexpr (K < NQ2) q2(kc, nxt);
          else if constexpr (NR1 + (K - NQ2) < NQ1) q1(ic<NR1 + (K - NQ2)>{}, L, nxt);
        }
      };
      constexpr int NG16 = 36 * NDX;                    // 16x16 slots (2 operations each), then 6 * NT 32x32 slots (6 each)
      static_assert(SERIAL || NQ2 + (NQ1 - (BK8 ? 0 : PRE + 72)) <= 2 * NG16 + 36 * NT, "not enough MFMA gaps for the gate math of a step");
      auto tail_mem = [&](auto) {};
      f32x4 ax[NDXA][2];
      if constexpr (HAS_DX) {
#pragma unroll
        for (int kk = 0; kk < NDX; ++kk) ax[kk][0] = ax[kk][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 qx[2][3];                                // operands of k block kb+1 are read under the MFMAs of kb
        auto rd_dx = [&](auto kbc) {                    // gate rows [r|z|n] <-> columns [dr|dz| . |dn]
          constexpr int kb = decltype(kbc)::value, col0 = kb < 4 ? kb * 32 : 192 + (kb - 4) * 32;
#pragma unroll
          for (int pp = 0; pp < 3; ++pp) qx[kb & 1][pp] = *(const bf16x8*)&pb[pp * DGP + col0];
        };
        // a FULL step still holds the recurrence's operands of the [dr|dz] columns (k blocks 0..3 are the same columns for dX):
        // only the two dn blocks are read here (12 fewer ds_read_b128 per step)
        constexpr int KB0 = (FULL && !BK8) ? 4 : 0;
        rd_dx(ic<KB0>{});
        FENCE();
        sfor<NG16>([&](auto sc) {
          constexpr int s = decltype(sc)::value, kb = s / (6 * NDX), kk = (s / 6) % NDX, t = s % 6;
          if constexpr (kb < KB0) ax[kk][kb & 1] = mf16<t, CT_DX>(AiB[kk][kb], q[kb], ax[kk][kb & 1]);
          else ax[kk][kb & 1] = mf16<t, CT_DX>(AiB[kk][kb], qx[kb & 1], ax[kk][kb & 1]);
          PINA(ax[kk][kb & 1]);
          FENCE();
          if constexpr (t == 0 && kk == 0 && kb + 1 < 6 && kb + 1 > KB0) rd_dx(ic<kb + 1>{});
          if constexpr (R4 == 0 && s >= 2 && s < 32) frag_read(ic<s - 2>{});
          if constexpr (ROLE == 2 && s >= 40 && s < 58) {
            // layer 1: the two B blocks (12 reads) and the first A block (6 reads) during the dX stream
            constexpr int n = s - 40;
            if constexpr (n < 12) {
              constexpr int blk = n / 6, pp = (n % 6) / 2, h = n % 2;
              put_half<h>(Bf[blk][pp], lds_tr_read4(ring + cur + tr_xh[h] + boff[blk == 0 ? 0 : 6] + pp * XHP));
            } else {
              constexpr int m = n - 12, pp = m / 2, h = m % 2;
              put_half<h>(Af[0][pp], lds_tr_read4(ring + cur + tr_dg[h] + aoff[0] + pp * DGP));
            }
          }
          if constexpr (!USE_DMA && FULL && s < 4) reg_piece(L, s, s_ld);     // r, z, hn, h_prev of step j+2: the coefficients of step j+1 are done
          gop(ic<2 * s>{});
          gop(ic<2 * s + 1>{});
          tail_mem(sc);
          FENCE();
        });
      }
      STAMP(3);
      // dW: NT tiles x 6 MFMAs
      sfor<6 * NT>([&](auto sc) {
        constexpr int s = decltype(sc)::value, tI = s / 6, t = s % 6;
        constexpr int ai = R4 == 0 ? (tI == 0 ? 0 : 1) : (R4 == 1 ? tI / 3 : (tI & 1));
        constexpr int bi = R4 == 0 ? tI : (R4 == 1 ? tI % 3 : (tI < 6 ? 0 : 1));
        accW[tI] = mf32<t, CT_DW>(Af[ai], Bf[bi], accW[tI]);
        PINA(accW[tI]);
        FENCE();
        if constexpr (ROLE == 2 && tI + 1 < NT) {        // the A block of the next tile, one transposed read per slot, other register set
          constexpr int pp = t / 2, h = t % 2;
          put_half<h>(Af[(tI + 1) & 1][pp], lds_tr_read4(ring + cur + tr_dg[h] + aoff[tI + 1] + pp * DGP));
        }
        if constexpr (HAS_DX && s == 1) {
          if (valid) {
#pragma unroll
            for (int kk = 0; kk < NDX; ++kk)
              *(float4*)(dxq + (L1K ? (2 * w + kk) : w) * 16) = make_float4(ax[kk][0][0] + ax[kk][1][0], ax[kk][0][1] + ax[kk][1][1],
                                                                         ax[kk][0][2] + ax[kk][1][2], ax[kk][0][3] + ax[kk][1][3]);
          }
          dxq -= dxstep;
        }
        if constexpr (!USE_DMA && FULL && HAS_X) {
          // the x pieces of step j+2, once the staging of step j+1's x is through (the gate math is NQ2 + NQ1 - NR1 operations)
          constexpr int XS_DONE = (NQ2 + NQ1 - NR1 - 2 * NG16 + 5) / 6 + 1;
          static_assert(XS_DONE + NXV < 6 * NT, "no slot left for the x loads");
          if constexpr (s >= XS_DONE && s < XS_DONE + NXV) reg_piece(L, 4 + (s - XS_DONE), s_ld);
          if constexpr (s == XS_DONE + NXV) ++steps_issued;
        }
        sfor<6>([&](auto oc) { gop(ic<2 * NG16 + 6 * s + decltype(oc)::value>{}); });
        tail_mem(ic<NG16 + s>{});
        FENCE();
      });
      STAMP(4);
      lds_barrier();
      STAMP(5);
      { const int o = cur; cur = nxt; nxt = o; }
    };
    const int n_full = n_steps - 1;
    for (int j = 0; j < n_full; ++j) step(std::true_type{}, j);
    if constexpr (!SEQ) step(std::false_type{}, n_full); // the last step: dX / dW only (ends on a barrier: the ring is free again)
    else {
#pragma unroll
      for (int v = 0; v < 4; ++v) stash_store(v);        // the last step's gate gradients
      lds_barrier();
    }
    if constexpr (USE_DMA) WAIT_VM(0);                   // the clamped re-loads past the last step must not land in the next tile's slots
#ifdef MSIG_STAMPS
    // record slot: gru_bwd_b4 — waves 0 / 2 (the two roles); gru_bwd_b5 — chain wave 0, bulk wave 4 (ROLE 5), bulk wave 6 (ROLE 6)
    constexpr int rec8 = ROLE == 4 ? 0 : (ROLE == 5 ? 1 : 2);
    const bool rec_on = (CH8 || BK8) ? (w == (ROLE == 6 ? 2 : 0)) : (w == 0 || w == 2);
    const int rec = (CH8 || BK8) ? rec8 : (w >> 1);
    if (a.dbg && lane == 0 && rec_on && tile == (int)blockIdx.x)
      for (int i = 0; i < 8; ++i) a.dbg[(((size_t)rec * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + i] = ph_[i];
#endif
  }

  if constexpr (!SEQ) {
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
  // bias gradients: fold the 16 batch rows through LDS (every wave is past its last read of the ring: the loop ends on a
  // barrier).  Scratch columns are [dr|dz|dhn|dn]; the partial wants [dr|dz|dn|dhn].
  float* scratch = (float*)ring;
  constexpr int RSB = 272;
  if constexpr (!BK8) {
#pragma unroll
    for (int g = 0; g < 4; ++g) *(float4*)&scratch[li * RSB + g * 64 + u0] = make_float4(bacc[g][0], bacc[g][1], bacc[g][2], bacc[g][3]);
  }
  lds_barrier();
  if constexpr (!BK8) {                      // (8-wave form: the chain waves are threads 0..255)
    float bsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) bsum += scratch[r * RSB + tid];
    Pp[192 * I + 192 * 64 + (tid < 128 ? tid : (tid < 192 ? tid + 64 : tid - 64))] = bsum;
  }
  }
}

template <int I, bool FOLDS>
__global__ __launch_bounds__(256, 1) void gru_bwd_b4(const GruArgs a, int n_tiles, const FoldCtx fc) {
  FOLD_GRU_ARGS_IF(FOLDS);
  if constexpr (I == 128) {
    bwd4_run<I, FOLDS, 2>(a, D, ax_, akey_, axkey_, n_tiles);
  } else {
    // two wave roles (wave-uniform branch; both sides execute the same number of s_barrier)
    if (threadIdx.x < 128) bwd4_run<I, FOLDS, 0>(a, D, ax_, akey_, axkey_, n_tiles);
    else bwd4_run<I, FOLDS, 1>(a, D, ax_, akey_, axkey_, n_tiles);
  }
}

// Layer 0 with two waves per SIMD: waves 0-3 chain, 4,5 and 6,7 bulk (wave-uniform branches; every side executes the same number
// of s_barrier).
template <bool FOLDS>
__global__ __launch_bounds__(512, 1) void gru_bwd_b5(const GruArgs a, int n_tiles, const FoldCtx fc) {
  FOLD_GRU_ARGS_IF(FOLDS);
  if (threadIdx.x < 256) bwd4_run<32, FOLDS, 4>(a, D, ax_, akey_, axkey_, n_tiles);
  else if (threadIdx.x < 384) bwd4_run<32, FOLDS, 5>(a, D, ax_, akey_, axkey_, n_tiles);
  else bwd4_run<32, FOLDS, 6>(a, D, ax_, akey_, axkey_, n_tiles);
}

// The recurrence of the latency form: grid (tiles, directions, folds), always fold-aware (a single model is a batch of one fold).
template <int I>
__global__ __launch_bounds__(256, 1) void gru_bwd_seq4(const GruArgs a, int n_tiles, const FoldCtx fc) {
  FOLD_GRU_ARGS_IF(true);
  bwd4_run<I, true, 3>(a, D, ax_, akey_, axkey_, n_tiles);
}

// Layer 0's recurrence of the latency form WITH layer 1's dW in the same launch.  dW of layer 1 needs what gru_bwd_seq4<128> left
// (the gate gradients in the stash) and nothing of layer 0; layer 0's recurrence needs layer 1's dX and nothing of its dW — and it
// is one dependent chain per tile on n_tiles x 2 CUs for a quarter of the step while the other CUs idle.  So the dW workgroups
// (gru_dw2.h dw2_role<128>, the same units per workgroup, the same partial rows: no bit changes) ride along: the launch costs what
// the recurrence costs as long as the dW work fits beside it (one model: 8 + 240 workgroups, one round; a fold batch of five: 40 +
// 1200 on 216 free CUs, 83 us of 244), and layer 1's bulk launch shrinks to its dX (one model: 31 -> 20 us).
// Roles follow the DISPATCH order (x fastest, then y, then z), not the grid coordinates: the first n_tiles x 2 x folds workgroups
// are the chains — every fold's, so that none of them queues behind other folds' dW — the rest dW.
__global__ __launch_bounds__(256, 1) void gru_bwd_seq4_dw1(const GruArgs a, const GruArgs a1, const int n_tiles, const int nwg1, const FoldCtx fc) {
  extern __shared__ __attribute__((aligned(16))) __bf16 dw_lds[];
  const int lin = (int)blockIdx.x + (int)gridDim.x * ((int)blockIdx.y + 2 * (int)blockIdx.z);
  const int n_chain = n_tiles * 2 * (int)gridDim.z;
  if (lin < n_chain) {
    const int tile = lin % n_tiles, dir = (lin / n_tiles) & 1, fold = lin / (2 * n_tiles);
    const int64_t foff_ = (int64_t)fc.slot[fold] * fc.stride;
    const uint32_t key = fc.key_gru[fold];
    GruDir Dv = a.dir[dir];
    FS(Dv.Wih); FS(Dv.Whh); FS(Dv.bih); FS(Dv.bhh); FS(Dv.h); FS(Dv.h_last); FS(Dv.stash); FS(Dv.dh); FS(Dv.dx); FS(Dv.part);
    const float* ax_ = a.x;
    FS(ax_);
    bwd4_run<32, true, 3>(a, Dv, ax_, key, key, n_tiles, tile, n_tiles);
  } else {
    const int m = lin - n_chain;
    const int wg = m % nwg1, dir = (m / nwg1) & 1, fold = m / (2 * nwg1);
    const int64_t foff_ = (int64_t)fc.slot[fold] * fc.stride;
    const uint32_t key = fc.key_gru[fold];
    GruDir Dv = a1.dir[dir];
    FS(Dv.Wih); FS(Dv.Whh); FS(Dv.bih); FS(Dv.bhh); FS(Dv.h); FS(Dv.h_last); FS(Dv.stash); FS(Dv.dh); FS(Dv.dx); FS(Dv.part);
    const float* ax_ = a1.x;
    FS(ax_);
    dw2_role<128>(a1, Dv, ax_, key, n_tiles, wg, nwg1, dw_lds);
  }
}

int gru_bwd_b4_lds_optin() {
  const hipFuncAttribute A = hipFuncAttributeMaxDynamicSharedMemorySize;
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b4<32, false>, A, BwdB4<32>::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b4<32, true>, A, BwdB4<32>::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b5<false>, A, BwdB4<32>::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b5<true>, A, BwdB4<32>::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_seq4<32>, A, BwdB4<32>::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_seq4<128>, A, BwdB4<32>::SMEM)) != hipSuccess) return (int)e;
  static_assert(BwdDw2<128>::SMEM <= BwdB4<32>::SMEM, "the dW role runs in the recurrence's LDS allocation");
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_seq4_dw1, A, BwdB4<32>::SMEM)) != hipSuccess) return (int)e;
#ifdef MSIG_B4_L1
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b4<128, false>, A, BwdB4<128>::SMEM)) != hipSuccess) return (int)e;
  if ((e = hipFuncSetAttribute((const void*)gru_bwd_b4<128, true>, A, BwdB4<128>::SMEM)) != hipSuccess) return (int)e;
#endif
  return 0;
}

int launch_gru_bwd_seq4(int dh_mode, const GruArgs& a, int n_tiles, int ndir, const FoldCtx& fc, hipStream_t st) {
  const dim3 grid(n_tiles, ndir, fc.n);
  if (dh_mode == 0) gru_bwd_seq4<32><<<grid, 256, BwdB4<32>::SMEM, st>>>(a, n_tiles, fc);
  else gru_bwd_seq4<128><<<grid, 256, BwdB4<32>::SMEM, st>>>(a, n_tiles, fc);
  MSIG_LAUNCH_CHECK();
#ifdef MSIG_STAMPS
  if (a.dbg && a.dir[0].n_steps > 1) {
    (void)hipStreamSynchronize(st);
    static unsigned long long h[8];
    (void)hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost);          // workgroup 0, direction 0, waves 0,1
    const double steps = a.dir[0].n_steps;
    fprintf(stderr, "[stamps seq4 dh_mode %d, cycles per step] loop top %.0f | wait + staged reads + 16 ops %.0f | recurrence %.0f | dependent gate math + stores %.0f | barrier %.0f\n",
            dh_mode, h[0] / steps, h[1] / steps, h[2] / steps, h[3] / steps, h[5] / steps);
  }
#endif
  return 0;
}

// layer 0's recurrence + layer 1's dW (a1: layer 1's argument block, nwg1 dW workgroups per direction): grid (n_tiles + nwg1, 2, fc.n)
int launch_gru_bwd_seq4_dw1(const GruArgs& a, const GruArgs& a1, int n_tiles, int nwg1, const FoldCtx& fc, hipStream_t st) {
  gru_bwd_seq4_dw1<<<dim3(n_tiles + nwg1, 2, fc.n), 256, BwdB4<32>::SMEM, st>>>(a, a1, n_tiles, nwg1, fc);
  MSIG_LAUNCH_CHECK();
  return 0;
}

int launch_gru_bwd_b4(int I, bool folds, const GruArgs& a, int n_tiles, int nwg, int ndir, const FoldCtx& fc, hipStream_t st, bool waves8) {
  const dim3 grid(nwg, ndir, folds ? fc.n : 1);
  if (I == 32 && waves8) {
    if (folds) gru_bwd_b5<true><<<grid, 512, BwdB4<32>::SMEM, st>>>(a, n_tiles, fc);
    else gru_bwd_b5<false><<<grid, 512, BwdB4<32>::SMEM, st>>>(a, n_tiles, fc);
    MSIG_LAUNCH_CHECK();
#ifdef MSIG_STAMPS
    if (a.dbg) {
      (void)hipStreamSynchronize(st);
      static unsigned long long h[3 * 2 * 256 * 8];
      const int per = ndir * nwg;
      (void)hipMemcpy(h, a.dbg, sizeof(unsigned long long) * 8 * 3 * per, hipMemcpyDeviceToHost);
      const char* names[3] = {"chain wave 0", "bulk wave 4 (dX + 3 dW tiles)", "bulk wave 6 (6 dW tiles, x staging)"};
      for (int role = 0; role < 3; ++role) {
        double acc[8] = {0};
        for (int i = 0; i < per; ++i) for (int j = 0; j < 8; ++j) acc[j] += (double)h[((size_t)role * per + i) * 8 + j] / per;
        const double steps = a.dir[0].n_steps;
        if (role == 0)
          fprintf(stderr, "[stamps b5 %s, cycles per step (first tile)] loop top %.0f | wait + staged reads + 16 ops %.0f | recurrence %.0f | dependent gate math + stores %.0f | barrier %.0f\n",
                  names[role], acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps, acc[5] / steps);
        else
          fprintf(stderr, "[stamps b5 %s, cycles per step (first tile)] loop top %.0f | x fetch / fragment reads + dX %.0f | dW %.0f | barrier %.0f\n",
                  names[role], acc[0] / steps, acc[3] / steps, acc[4] / steps, acc[5] / steps);
      }
    }
#endif
    return 0;
  } else if (I == 32) {
    if (folds) gru_bwd_b4<32, true><<<grid, 256, BwdB4<32>::SMEM, st>>>(a, n_tiles, fc);
    else gru_bwd_b4<32, false><<<grid, 256, BwdB4<32>::SMEM, st>>>(a, n_tiles, fc);
  } else {
    // Layer 1 (ROLE 2) is NOT shipped: its 216 resident weight registers + 144 accumulator registers leave ~150 for a pipelined
    // gate math that needs ~240.  With the DMA ring every scratch reload's compiler-made vmcnt(0) drains the ring: 2.36 ms; with
    // register loads instead (USE_DMA = false, what the code does now) 67 dwords of scratch per lane: 2.59 ms — against 1.72 ms
    // for gru_bwd_b3<128> (profiles/r03_bwd4_stamps.log).  Parity-green both ways; make EXTRA=-DMSIG_B4_L1 builds it.
#ifdef MSIG_B4_L1
    if (folds) gru_bwd_b4<128, true><<<grid, 256, BwdB4<128>::SMEM, st>>>(a, n_tiles, fc);
    else gru_bwd_b4<128, false><<<grid, 256, BwdB4<128>::SMEM, st>>>(a, n_tiles, fc);
#else
    return MSIG_E_SHAPE;
#endif
  }
  MSIG_LAUNCH_CHECK();
#ifdef MSIG_STAMPS
  if (a.dbg) {
    (void)hipStreamSynchronize(st);
    static unsigned long long h[2 * 2 * 256 * 8];
    const int per = ndir * nwg;
    (void)hipMemcpy(h, a.dbg, sizeof(unsigned long long) * 8 * 2 * per, hipMemcpyDeviceToHost);
    for (int role = 0; role < (I == 32 ? 2 : 1); ++role) {
      double acc[8] = {0};
      for (int i = 0; i < per; ++i) for (int j = 0; j < 8; ++j) acc[j] += (double)h[((size_t)role * per + i) * 8 + j] / per;
      const double steps = a.dir[0].n_steps;
      fprintf(stderr, "[stamps b4 I=%d waves %s, cycles per step (first tile)] loop top %.0f | reads + vmcnt wait + 16 ops %.0f | recurrence %.0f | dX %.0f | dW %.0f | barrier %.0f\n",
              I, role ? "2,3" : "0,1", acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps, acc[4] / steps, acc[5] / steps);
    }
  }
#endif
  return 0;
}
