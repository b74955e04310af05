// dW of the latency form (gru_bwd_dw2 / gru_bwd_dxdw in gru.hip, and the dW workgroups that ride in gru_bwd_seq4's launch in
// gru_bwd4.hip): the plane geometry of gru_bwd_b3, whose recipe it uses, and the role itself.  Commentary: gru.hip, "Bulk kernels of
// the latency form".
#pragma once
#include "gru_args.h"

#ifndef LDS_AS
#define LDS_AS __attribute__((address_space(3)))
#endif
__device__ __forceinline__ bf16x4 lds_tr_read(const __bf16* p) {      // ds_read_b64_tr_b16; EXEC must be all ones
  typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 v4;
  const v4 r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS v4*)p);
  return __builtin_bit_cast(bf16x4, r);
}

template <int I> struct BwdB3 {
  static constexpr bool L1K = (I == 128);
  static constexpr int SD = 272;                       // gate-gradient plane row stride (bf16 elements): 136 dwords = 8 * 17
  static constexpr int SX = L1K ? 208 : 112;           // [x | h_prev] plane row stride: 104 = 8 * 13 / 56 = 8 * 7 dwords
  static constexpr int DGP = 16 * SD, XHP = 16 * SX;   // elements per piece plane
  static constexpr int BUFE = 3 * DGP + 3 * XHP;       // elements per ring buffer (36 864 B / 46 080 B)
  static constexpr int SMEM = 3 * BUFE * 2;            // three buffers
};

template <int I> struct BwdDw2 {
  using G = BwdB3<I>;
  static constexpr int SMEM = 2 * G::BUFE * 2;          // two plane buffers (the units of a pair): 73 728 B / 92 160 B
  static constexpr int DXP = 224;                       // dX role: plane row stride (bf16 elements) of gru_bwd_dx
};

template <int I>
__device__ __forceinline__ void dw2_role(const GruArgs& a, const GruDir& D, const float* ax_, const uint32_t akey_, const int n_tiles,
                                         const int wg, const int nwg, __bf16* ring) {
  using G = BwdB3<I>;
  constexpr int NKB = I / 16;
  constexpr int SD = G::SD, SX = G::SX, DGP = G::DGP, XHP = G::XHP, BUFE = G::BUFE;
  constexpr int NXV = (16 * I / 4 + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int u0 = w * 16 + lq * 4;
  const int n_units = n_tiles * D.n_steps;
  if (wg >= n_units) return;                           // no unit: its partial row is never read (reduce_dw takes min(nwg, units) rows)
  // per-lane LDS offsets: BwdB3's (gru_bwd_b3 above)
  const int sw_li = ((li >> 2) & 1) * 8;
  const int wr_dg = li * SD + (u0 ^ sw_li);
  const int wr_h = 3 * DGP + li * SX + ((I + u0) ^ (((li >> 2) & 3) * 4));
  const int trow = 4 * (lq & 1) + (li >> 2);
  const int tr_dg = trow * SD + ((4 * (li & 3)) ^ ((lq & 1) * 8));
  const int tr_xh0 = 3 * DGP + trow * SX + ((4 * (li & 3)) ^ ((lq & 1) * 4));
  const int tr_xh1 = 3 * DGP + (trow + 8) * SX + ((4 * (li & 3)) ^ ((2 + (lq & 1)) * 4));
  const int sel = (lq >> 1) ? BUFE : 0;                // k groups 2, 3 contract the second unit of the pair
  int xrow_off[NXV]; bool xlive[NXV];
#pragma unroll
  for (int v = 0; v < NXV; ++v) {
    const int idx = tid + 256 * v, row = idx / (I / 4), c4 = idx - row * (I / 4);
    xlive[v] = idx < 16 * I / 4;
    xrow_off[v] = xlive[v] ? 3 * DGP + row * SX + ((4 * c4) ^ (((row >> 2) & 3) * 4)) : 3 * DGP;
  }
  f32x4 accH[3][4], accI[3][NKB];      // TRANSPOSED tiles: lane (li, lq), element e <-> gate row w*16 + li, column cb*16 + 4 lq + e
#pragma unroll
  for (int g = 0; g < 3; ++g) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) accH[g][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < NKB; ++cb) accI[g][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float bacc[4][4];                                    // plane column order [dr|dz|dhn|dn]
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) bacc[g][e] = 0.f;
  struct Unit { float4 g[4], hp, xv[NXV]; uint32_t xw[NXV]; float hkeep; };      // stash order: dr, dz, dn, dhn
  const int drop_thr = a.drop_thr; const float dscale = a.drop_scale;
  auto load = [&](Unit& U, int unit) {
    const int tile = unit / D.n_steps, s = unit - tile * D.n_steps;
    const int t = D.t_start + D.t_sign * s;
    const float4* sp = D.stash + ((size_t)unit * 4 + w) * 4 * 64 + lane;
    U.g[0] = sp[0]; U.g[1] = sp[64]; U.g[2] = sp[128]; U.g[3] = sp[192];
    const int b = min(tile * 16 + li, a.B - 1);        // rows >= B carry dg == 0: any finite operand will do
    U.hp = *(const float4*)(D.h + (int64_t)b * D.h_bs + (int64_t)(s > 0 ? t - D.t_sign : t) * D.h_ts + D.h_col + u0);
    U.hkeep = (s == 0) ? 0.0f : 1.0f;                  // h_{-1} = 0, applied in `planes`: a consumer next to the load would wait for it here
#pragma unroll
    for (int v = 0; v < NXV; ++v) {
      const int idx = (tid + 256 * v) % (16 * I / 4), row = idx / (I / 4), c4 = idx - row * (I / 4);
      const int bb = min(tile * 16 + row, a.B - 1);
      const int64_t e0 = (int64_t)bb * a.x_bs + (int64_t)t * a.x_ts + 4 * c4;
      U.xv[v] = *(const float4*)(ax_ + e0);
      U.xw[v] = drop_word((uint32_t)e0, akey_);
    }
  };
  auto planes = [&](const Unit& U, int boff) {
    const float dr[4] = {U.g[0].x, U.g[0].y, U.g[0].z, U.g[0].w}, dz[4] = {U.g[1].x, U.g[1].y, U.g[1].z, U.g[1].w};
    const float dn[4] = {U.g[2].x, U.g[2].y, U.g[2].z, U.g[2].w}, dhn[4] = {U.g[3].x, U.g[3].y, U.g[3].z, U.g[3].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { bacc[0][e] += dr[e]; bacc[1][e] += dz[e]; bacc[2][e] += dhn[e]; bacc[3][e] += dn[e]; }
    bf16x4 pc[4][3];
    split3_quad(dr, pc[0]); split3_quad(dz, pc[1]); split3_quad(dhn, pc[2]); split3_quad(dn, pc[3]);
    __bf16* pw = ring + boff + wr_dg;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) *(bf16x4*)&pw[pp * DGP + g * 64] = pc[g][pp];
    {
      const float hp[4] = {U.hp.x * U.hkeep, U.hp.y * U.hkeep, U.hp.z * U.hkeep, U.hp.w * U.hkeep};
      bf16x4 hpc[3];
      split3_quad(hp, hpc);
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) *(bf16x4*)&ring[boff + wr_h + pp * XHP] = hpc[pp];
    }
#pragma unroll
    for (int v = 0; v < NXV; ++v) {
      float q[4] = {U.xv[v].x, U.xv[v].y, U.xv[v].z, U.xv[v].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) q[e] *= drop_mul(U.xw[v], e, drop_thr, dscale);      // branch-free: thr 0 keeps everything with scale 1
      bf16x4 xpc[3];
      split3_quad(q, xpc);
      if (xlive[v]) {
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) *(bf16x4*)&ring[boff + xrow_off[v] + pp * XHP] = xpc[pp];
      }
    }
  };
  auto tr_frag = [&](int off0, int off1, int pstride, bf16x8 (&f)[3]) {
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) {
      const bf16x4 lo = lds_tr_read(ring + off0 + pp * pstride), hi = lds_tr_read(ring + off1 + pp * pstride);
      f[pp] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };
  auto dw_phase = [&]() {                               // dW += dg^T [x | h_prev] over the pair in the two buffers
    const int ta = sel + tr_dg + w * 16;
    bf16x8 Ar[3], Az[3], Ahn[3], An[3];
    tr_frag(ta + 0 * 64, ta + 0 * 64 + 8 * SD, DGP, Ar);
    tr_frag(ta + 1 * 64, ta + 1 * 64 + 8 * SD, DGP, Az);
    tr_frag(ta + 2 * 64, ta + 2 * 64 + 8 * SD, DGP, Ahn);
    tr_frag(ta + 3 * 64, ta + 3 * 64 + 8 * SD, DGP, An);
    bf16x8 Bf[2][3];                                    // B block bi+1 is read under the MFMAs of bi: h_prev blocks, then x blocks
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
        accH[0][bi] = mfma_bf16x3<CT_DW>(Bf[bi & 1], Ar, accH[0][bi]);
        accH[1][bi] = mfma_bf16x3<CT_DW>(Bf[bi & 1], Az, accH[1][bi]);
        accH[2][bi] = mfma_bf16x3<CT_DW>(Bf[bi & 1], Ahn, accH[2][bi]);
      } else {
        accI[0][bi - 4] = mfma_bf16x3<CT_DW>(Bf[bi & 1], Ar, accI[0][bi - 4]);
        accI[1][bi - 4] = mfma_bf16x3<CT_DW>(Bf[bi & 1], Az, accI[1][bi - 4]);
        accI[2][bi - 4] = mfma_bf16x3<CT_DW>(Bf[bi & 1], An, accI[2][bi - 4]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // units wg, wg + nwg, wg + 2 nwg, ... in pairs; an unpaired last unit contracts with an all-zero phantom
  Unit UA, UB;
  int u = wg;
  bool hasB = u + nwg < n_units;
  load(UA, u);
  if (hasB) load(UB, u + nwg);
  for (;;) {
    planes(UA, 0);
    if (hasB) planes(UB, BUFE);
    else for (int i = tid; i < BUFE / 8; i += 256) *(float4*)&ring[BUFE + 8 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
    u += 2 * nwg;
    const bool nA = u < n_units, nB = u + nwg < n_units;
    lds_barrier();
    if (nA) load(UA, u);                                // the next pair's operands arrive under this pair's MFMAs
    if (nB) load(UB, u + nwg);
    dw_phase();
    lds_barrier();                                      // every read of the planes is done
    if (!nA) break;
    hasB = nB;
  }
  // ---- partial: [dW_ih 192*I][dW_hh 192*64][db 256 = dr,dz,dn,dhn] ----
  // The MFMAs above take [x | h_prev] as the A operand and the gate gradients as B: a lane's four accumulator elements are four
  // consecutive COLUMNS (4 lq + e) of gate row w*16 + li, i.e. one 16-byte store (the D layout of dg^T x would be four rows of one
  // column: 432 scattered dword stores per lane, ~10 us of a workgroup's 40 at the reference's batch size).
  float* P = D.part + (size_t)wg * (192 * I + 192 * 64 + 256);
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int row = g * 64 + w * 16 + li;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
      *(float4*)&P[192 * I + (size_t)row * 64 + cb * 16 + lq * 4] = make_float4(accH[g][cb][0], accH[g][cb][1], accH[g][cb][2], accH[g][cb][3]);
#pragma unroll
    for (int cb = 0; cb < NKB; ++cb)
      *(float4*)&P[(size_t)row * I + cb * 16 + lq * 4] = make_float4(accI[g][cb][0], accI[g][cb][1], accI[g][cb][2], accI[g][cb][3]);
  }
  float* scratch = (float*)ring;                        // bias gradients: fold the 16 batch rows; scratch columns [dr|dz|dhn|dn]
#pragma unroll
  for (int g = 0; g < 4; ++g) *(float4*)&scratch[li * RS + g * 64 + u0] = make_float4(bacc[g][0], bacc[g][1], bacc[g][2], bacc[g][3]);
  __syncthreads();
  float bsum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) bsum += scratch[r * RS + tid];
  P[192 * I + 192 * 64 + (tid < 128 ? tid : (tid < 192 ? tid + 64 : tid - 64))] = bsum;
}

