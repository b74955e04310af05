// Shapes of gru_bwd_b4's LDS plane ring, shared between the kernel file (gru_bwd4.hip) and the launcher (gru.hip).
#pragma once
template <int I> struct BwdB4 {
  static constexpr bool L1K = (I == 128);
  static constexpr int SD = 288;                       // gate-gradient plane row stride (bf16 elements): 144 dwords = 16 * 9
  static constexpr int SX = L1K ? 224 : 96;            // [x | h_prev] plane row stride: 112 = 16 * 7 / 48 = 16 * 3 dwords
  static constexpr int DGP = 16 * SD, XHP = 16 * SX;   // elements per piece plane
  static constexpr int BUFE = 3 * DGP + 3 * XHP;       // elements per ring buffer (36 864 B / 49 152 B)
  // operand staging ring behind the planes: NST slots x 4 waves x NPC pieces of 1 KiB (one global_load_lds_dwordx4 each)
  static constexpr int NPC = 6, NST = L1K ? 2 : 3;
  static constexpr int STG0 = 2 * BUFE * 2;            // byte offset of the staging ring: 73 728 / 98 304
  static constexpr int SLOTB = 4 * NPC * 1024;         // 24 576 B per slot
  static constexpr int SMEM = STG0 + NST * SLOTB;      // 147 456 B for both layers
};
// gru_bwd_b6 (gru_bwd6.hip): gate-gradient planes (ring of two) | [x | h_prev] planes (ring of four) | hn ring | operand staging
struct BwdB6 {
  static constexpr int SD = 288, SX = 96, DGP = 16 * SD, XHP = 16 * SX;      // as BwdB4<32>
  static constexpr int DGBUF = 3 * DGP, XHBUF = 3 * XHP;                     // elements per plane buffer
  static constexpr int XH0 = 2 * DGBUF, NXH = 4;                             // element offset / depth of the [x | h_prev] ring
  static constexpr int HNH0 = (XH0 + NXH * XHBUF) * 2;                       // byte offset of the hn ring: 2 x (4 unit blocks x 1 KiB)
  static constexpr int STG0 = HNH0 + 2 * 4096;                               // byte offset of the staging ring
  static constexpr int NPC = 4, NST = 2, SLOTB = 4 * NPC * 1024;             // r, z, h_prev, upstream dh per chain wave; two steps in flight
  static constexpr int WHN0 = STG0 + NST * SLOTB;                            // byte offset of bulk A's hn weight pieces: 2 waves x 2 unit blocks x 6 KiB
  static constexpr int SMEM = WHN0 + 2 * 12288;                              // 157 696 B
};
struct GruArgs;
struct FoldCtx;
// grid (workgroups, directions, folds); folds = fc.stride != 0.  Returns a hipError_t / MSIG_E_* code.
// waves8: layer 0 as gru_bwd_b5 (512 threads: four chain waves + four bulk waves)
int launch_gru_bwd_b4(int I, bool folds, const GruArgs& a, int n_tiles, int nwg, int ndir, const FoldCtx& fc, hipStream_t st, bool waves8 = false);
// the recurrence of the latency form (replaces gru_bwd_seq): dh_mode 0 = layer 0, 1 = layer 1; grid (n_tiles, ndir, fc.n)
int launch_gru_bwd_seq4(int dh_mode, const GruArgs& a, int n_tiles, int ndir, const FoldCtx& fc, hipStream_t st);
// the same for layer 0 with layer 1's dW workgroups (gru_dw2.h) in the launch: grid (n_tiles + nwg1, 2, fc.n), roles by dispatch order
int launch_gru_bwd_seq4_dw1(const GruArgs& a, const GruArgs& a1, int n_tiles, int nwg1, const FoldCtx& fc, hipStream_t st);
int gru_bwd_b4_lds_optin();
// layer 0 as gru_bwd_b6: 512 threads, W_hn h + b_hn recomputed by the bulk waves (the forward pass stores r, z only)
int launch_gru_bwd_b6(bool folds, const GruArgs& a, int n_tiles, int nwg, int ndir, const FoldCtx& fc, hipStream_t st);
int gru_bwd_b6_lds_optin();
