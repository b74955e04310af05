// ChannelAttention + cnn_encoder (models.py:7-31,45-54) forward and backward for gfx950.
//
// Layouts: the input window is (B,C,T) as the reference hands it over; every activation
// after conv1 is time-major "NLC" (B, L, channels) so that (i) a 16x16x4 MFMA result tile
// (lane = position, 4 consecutive channels per lane) is stored as one 16-byte vector per
// lane, and (ii) the final pooled tensor IS the (B,T',32) sequence the GRU consumes
// (models.py:77's permute never materialises).
//
// Both convolutions are im2col contractions fed from an LDS-staged window chunk:
//   conv1: Y[t][o]  = sum_{c,kk} (s[b,c] w[o,c,kk]) x[c][2t+kk-3]      K = 7C   (42 @ C=6)
//   conv2: Y[t][o]  = sum_{kk,c} w[o,c,kk] p1[2t+kk-2][c]               K = 80
// The sigmoid gate of ChannelAttention is folded into conv1's weights per window
// (x * s is never written).  BatchNorm (training) needs batch statistics, so each conv
// kernel also emits per-workgroup partial sums; bn_finalize turns them into scale/shift
// and updates the running statistics exactly like nn.BatchNorm1d (biased variance for
// the output, unbiased for running_var).
#include "msig_dev.h"

// ------------------------------------------------------------------------------------
// K1/K2: per-window channel means and the gate MLP (models.py:24-29)
// ------------------------------------------------------------------------------------
// CT = the channel count at compile time (0: any): the channels' loads of an iteration are issued together and their reductions
// interleave — one channel after the other the kernel was C dependent round trips to memory (14.5 us at the reference's 64
// windows, where a workgroup's 92 KB are the whole job of its CU).  Per channel the sums run over the same samples in the same
// order either way.
template <int CT>
__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ x, const float* __restrict__ W1,
                                                   const float* __restrict__ W2, float* __restrict__ mean_out,
                                                   float* __restrict__ pre_out, float* __restrict__ s_out,
                                                   float* __restrict__ eo_out, int C, int T, int Cr, const FoldCtx fc) {
  FOLD_BEGIN; FS(x); FS(W1); FS(W2); FS(mean_out); FS(pre_out); FS(s_out); FS(eo_out);
  __shared__ float red[4][MSIG_MAX_C];
  __shared__ float red_eo[4][2 * MSIG_MAX_C];
  __shared__ float mean_s[MSIG_MAX_C];
  __shared__ float hid_s[MSIG_MAX_C / 4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* xb = x + (size_t)b * C * T;
  if (CT > 0 && (T & 3) == 0) {
    constexpr int CN = CT > 0 ? CT : 1;
    float acc[CN], ev[CN], od[CN];
#pragma unroll
    for (int c = 0; c < CN; ++c) acc[c] = ev[c] = od[c] = 0.f;
    const int T4 = T / 4;
    constexpr int U = CN <= 4 ? 4 : 2;             // iterations whose loads are issued together (U x CN float4 per thread)
    for (int i0 = tid; i0 < T4; i0 += 256 * U) {
      float4 q[U][CN];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + 256 * u < T4 ? i0 + 256 * u : i0;      // clamped: a valid address, the value is not used
#pragma unroll
        for (int c = 0; c < CN; ++c) q[u][c] = ((const float4*)xb)[(size_t)c * T4 + i];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + 256 * u < T4) {
#pragma unroll
          for (int c = 0; c < CN; ++c) {
            acc[c] += (q[u][c].x + q[u][c].y) + (q[u][c].z + q[u][c].w);
            ev[c] += q[u][c].x + q[u][c].z; od[c] += q[u][c].y + q[u][c].w;
          }
        }
    }
#pragma unroll
    for (int c = 0; c < CN; ++c) {
      const float a = wave_sum(acc[c]);
      if (lane == 0) red[w][c] = a;
      if (eo_out) {                                  // uniform
        const float e = wave_sum(ev[c]), o = wave_sum(od[c]);
        if (lane == 0) { red_eo[w][2 * c] = e; red_eo[w][2 * c + 1] = o; }
      }
    }
  } else
  for (int c = 0; c < C; ++c) {
    const float* xc = xb + (size_t)c * T;
    float acc = 0.f, ev = 0.f, od = 0.f;         // ev / od: sums of the even- / odd-indexed samples (training: conv1's backward needs sum_t x[2t+k-3])
    if ((T & 3) == 0) {
      const float4* x4 = (const float4*)xc;
      for (int i = tid; i < T / 4; i += 256) {
        const float4 q = x4[i];
        acc += (q.x + q.y) + (q.z + q.w);
        ev += q.x + q.z; od += q.y + q.w;
      }
    } else {
      for (int i = tid; i < T; i += 256) { const float v = xc[i]; acc += v; if (i & 1) od += v; else ev += v; }
    }
    acc = wave_sum(acc);
    if (lane == 0) red[w][c] = acc;
    if (eo_out) {                                  // uniform
      ev = wave_sum(ev); od = wave_sum(od);
      if (lane == 0) { red_eo[w][2 * c] = ev; red_eo[w][2 * c + 1] = od; }
    }
  }
  __syncthreads();
  if (tid < C) {
    const float m = (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) / (float)T;
    mean_s[tid] = m;
    mean_out[(size_t)b * C + tid] = m;
  }
  if (eo_out && tid < 2 * C) eo_out[(size_t)b * 2 * C + tid] = (red_eo[0][tid] + red_eo[1][tid]) + (red_eo[2][tid] + red_eo[3][tid]);
  __syncthreads();
  if (tid < Cr) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += W1[tid * C + c] * mean_s[c];
    pre_out[(size_t)b * Cr + tid] = a;
    hid_s[tid] = a > 0.f ? a : 0.f;
  }
  __syncthreads();
  if (tid < C) {
    float a = 0.f;
    for (int j = 0; j < Cr; ++j) a += W2[tid * Cr + j] * hid_s[j];
    s_out[(size_t)b * C + tid] = sigmoidf_fast(a);   // Cr == 0 -> sigmoid(0) = 0.5
  }
}

// ChannelAttention.forward on its own (models.py:24-31): out = x * s[:, :, None].  Inside the model the product is never written
// (the gate is folded into conv1's taps); this is the module's stand-alone forward for callers that use it directly.
__global__ __launch_bounds__(256) void gate_scale_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ out,
                                                         int T, int64_t n_rows) {
  for (int64_t row = blockIdx.x; row < n_rows; row += gridDim.x) {           // row = (window, channel)
    const float sc = s[row];
    const float* xr = x + row * T;
    float* o = out + row * T;
    if ((T & 3) == 0 && (((uintptr_t)xr | (uintptr_t)o) & 15) == 0) {
      for (int i = threadIdx.x; i < T / 4; i += 256) {
        const float4 q = ((const float4*)xr)[i];
        ((float4*)o)[i] = make_float4(q.x * sc, q.y * sc, q.z * sc, q.w * sc);
      }
    } else {
      for (int i = threadIdx.x; i < T; i += 256) o[i] = xr[i] * sc;
    }
  }
}
int launch_channel_attention(const float* x, const float* W1, const float* W2, int B, int C, int T, float* out, float* s, float* scratch,
                             hipStream_t st) {
  const int Cr = C / 4;
  const FoldCtx fc = single_fold(nullptr);
  gate_kernel<0><<<dim3(B, 1, 1), 256, 0, st>>>(x, W1, W2, scratch, scratch + (size_t)B * C, s, nullptr, C, T, Cr, fc);
  MSIG_LAUNCH_CHECK();
  const int64_t rows = (int64_t)B * C;
  gate_scale_kernel<<<dim3((unsigned)(rows < 4096 ? rows : 4096)), 256, 0, st>>>(x, s, out, T, rows);
  MSIG_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------
// conv1 forward: Conv1d(C,16,k7,s2,p3) on gate-scaled input, NLC output + BN partials
// ------------------------------------------------------------------------------------
#define C1_CHUNK 256            // output positions per work item (4 waves x 4 blocks of 16)
#define C1_XW (2 * C1_CHUNK + 8)        // samples staged per channel: x[2*t0 - 4 .. 2*t0 + 515]
#define C1_XP (C1_XW / 2 + 4)           // floats per parity plane (even / odd samples)
#define C1_MAXKM 28                     // ceil(16 * 7 / 4)

// Stages one chunk of the input window.  DEINT: samples are split into an even and an odd plane
// per channel (xs[(2c + parity) * C1_XP + i] = x[c][2*t0 - 4 + 2i + parity]) so that the stride-2
// im2col reads of the forward are consecutive dwords; otherwise the natural order (row stride C1_XW).
template <bool DEINT>
__device__ __forceinline__ void stage_x_chunk(float* xs, const float* __restrict__ xb, int C, int T, int t0, int tid, int xstride = C1_XW) {
  const int g_base = 2 * t0 - 4;
  if ((T & 3) == 0) {
    for (int i = tid; i < C * (C1_XW / 4); i += 256) {
      const int c = i / (C1_XW / 4), i4 = i - c * (C1_XW / 4), g0 = g_base + 4 * i4;
      const int gc = g0 < 0 ? 0 : (g0 > T - 4 ? T - 4 : g0);          // unconditional, clamped load
      float4 q = *(const float4*)(xb + (size_t)c * T + gc);
      if (g0 < 0 || g0 > T - 4) q = make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding (whole vectors: T % 4 == 0)
      if (DEINT) {
        *(float2*)&xs[(2 * c) * C1_XP + 2 * i4] = make_float2(q.x, q.z);
        *(float2*)&xs[(2 * c + 1) * C1_XP + 2 * i4] = make_float2(q.y, q.w);
      } else {
        *(float4*)&xs[c * xstride + 4 * i4] = q;
      }
    }
  } else {
    for (int i = tid; i < C * C1_XW; i += 256) {
      const int c = i / C1_XW, j = i - c * C1_XW, g = g_base + j;
      const int gc = g < 0 ? 0 : (g > T - 1 ? T - 1 : g);
      float v = xb[(size_t)c * T + gc];
      if (g < 0 || g > T - 1) v = 0.f;
      if (DEINT) xs[(2 * c + (j & 1)) * C1_XP + (j >> 1)] = v; else xs[c * xstride + j] = v;
    }
  }
}

// CT > 0: channel count known at compile time (straight-line MFMA stream, no per-MFMA guards);
// CT == 0: generic fallback for C > 8 with wave-uniform guards.
template <int CT>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                        const float* __restrict__ gate_s, float* __restrict__ y1,
                                                        float* __restrict__ part, int B, int Crt, int T, int L1,
                                                        int want_stats, const FoldCtx fc) {
  FOLD_BEGIN; FS(x); FS(w1); FS(gate_s); FS(y1); FS(part);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = CT > 0 ? CT : Crt;
  const int K = C * 7, KM = (K + 3) / 4;
  constexpr int KMC = CT > 0 ? (CT * 7 + 3) / 4 : C1_MAXKM;
  float* xs = smem;                       // [2C][C1_XP] parity planes
  float* ws = smem + 2 * C * C1_XP;       // [4*KM][16]  gate-scaled weights, k = c*7 + kk
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  // per-lane LDS offset of the B operand of k-step m (k = 4m + lq -> channel c, tap kk):
  // sample j = 2*pl + kk + 1 of the staged chunk  ->  plane (kk even ? odd : even), index pl + (kk+1)/2
  int xo[KMC];
#pragma unroll
  for (int m = 0; m < KMC; ++m) {
    const int k = 4 * m + lq, kc = k < K ? k : 0, c = kc / 7, kk = kc - 7 * c;
    xo[m] = (2 * c + ((kk + 1) & 1)) * C1_XP + ((kk + 1) >> 1);
  }
  const int nchunk = (L1 + C1_CHUNK - 1) / C1_CHUNK;
  const int nitems = B * nchunk;
  f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
  // Software pipeline (compile-time channel count, T % 4 == 0): the next item's x chunk and gate values are
  // loaded into registers while the current item's MFMAs run, so an item is  sync, regs -> LDS, sync, issue the
  // next loads, compute  and no workgroup ever sits on a global-load latency between two barriers.
  constexpr bool PIPE = CT > 0;
  constexpr int NX4 = PIPE ? (CT * (C1_XW / 4) + 255) / 256 : 1;      // float4 pieces of the x chunk per thread
  constexpr int NWS = PIPE ? (4 * KMC * 16 + 255) / 256 : 1;          // gate-scaled weight entries per thread
  const bool pipe = PIPE && (T & 3) == 0;
  float4 xr[NX4];
  float wfix[NWS], gr[NWS];
  if (pipe) {
#pragma unroll
    for (int j = 0; j < NWS; ++j) {      // the weight itself never changes: only the window's gate does
      const int i = tid + 256 * j, k = i >> 4, o = i & 15;
      wfix[j] = (i < 4 * KM * 16 && k < K) ? w1[o * K + k] : 0.f;
    }
  }
  auto prefetch = [&](int item) {
    const int b = item / nchunk, t0 = (item - b * nchunk) * C1_CHUNK, g_base = 2 * t0 - 4;
    const float* xb = x + (size_t)b * C * T;
#pragma unroll
    for (int j = 0; j < NX4; ++j) {
      const int i = tid + 256 * j, ic = i < C * (C1_XW / 4) ? i : 0;
      const int c = ic / (C1_XW / 4), i4 = ic - c * (C1_XW / 4), g0 = g_base + 4 * i4;
      const int gc = g0 < 0 ? 0 : (g0 > T - 4 ? T - 4 : g0);              // unconditional, clamped load
      xr[j] = *(const float4*)(xb + (size_t)c * T + gc);
    }
#pragma unroll
    for (int j = 0; j < NWS; ++j) {
      const int i = tid + 256 * j, k = i >> 4, kc = (i < 4 * KM * 16 && k < K) ? k : 0;
      gr[j] = gate_s[(size_t)b * C + kc / 7];
    }
  };
  if (pipe && (int)blockIdx.x < nitems) prefetch(blockIdx.x);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / nchunk, t0 = (item - b * nchunk) * C1_CHUNK;
    __syncthreads();
    if (pipe) {
      const int g_base = 2 * t0 - 4;
#pragma unroll
      for (int j = 0; j < NX4; ++j) {
        const int i = tid + 256 * j;
        if (i < C * (C1_XW / 4)) {
          const int c = i / (C1_XW / 4), i4 = i - c * (C1_XW / 4), g0 = g_base + 4 * i4;
          float4 q = xr[j];
          if (g0 < 0 || g0 > T - 4) q = make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding (whole vectors: T % 4 == 0)
          *(float2*)&xs[(2 * c) * C1_XP + 2 * i4] = make_float2(q.x, q.z);
          *(float2*)&xs[(2 * c + 1) * C1_XP + 2 * i4] = make_float2(q.y, q.w);
        }
      }
#pragma unroll
      for (int j = 0; j < NWS; ++j) {
        const int i = tid + 256 * j;
        if (i < 4 * KM * 16) ws[i] = wfix[j] * gr[j];
      }
    } else {
      stage_x_chunk<true>(xs, x + (size_t)b * C * T, C, T, t0, tid);
      for (int i = tid; i < 4 * KM * 16; i += 256) {
        const int k = i >> 4, o = i & 15, kc = k < K ? k : 0;
        const float v = w1[o * K + kc] * gate_s[(size_t)b * C + kc / 7];
        ws[i] = (k < K) ? v : 0.f;
      }
    }
    __syncthreads();
    if (pipe && item + (int)gridDim.x < nitems) prefetch(item + gridDim.x);
    // the A operands (this window's gate-scaled weights) are the same for the wave's four position blocks: read
    // them once per item (a DS instruction costs a wave ~12 cycles; this removes 33 of the 88 per item)
    float wa[KMC];
#pragma unroll
    for (int m = 0; m < KMC; ++m) wa[m] = (CT > 0 || m < KM) ? ws[(4 * m + lq) * 16 + li] : 0.f;
#pragma unroll
    for (int pbi = 0; pbi < 4; ++pbi) {
      const int pl = (w * 4 + pbi) * 16 + li;      // position within the chunk
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < KMC; m += 2) {           // fully unrolled
        if (CT > 0 || m < KM) acc0 = mfma16(wa[m], xs[xo[m] + pl], acc0);
        if (m + 1 < KMC && (CT > 0 || m + 1 < KM)) acc1 = mfma16(wa[m + 1], xs[xo[m + 1] + pl], acc1);
      }
      const f32x4 acc = acc0 + acc1;
      const int t = t0 + pl;
      if (t < L1) {
        *(float4*)(y1 + ((size_t)b * L1 + t) * 16 + lq * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { ssum[e] += acc[e]; ssq[e] += acc[e] * acc[e]; }
      }
    }
  }
  if (want_stats) {
    // reduce over the 16 positions (li) of each lane group, then over waves
    __syncthreads();
    float* red = smem;   // [4 waves][32]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s1 = ssum[e], s2 = ssq[e];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
      if (li == 0) { red[w * 32 + lq * 4 + e] = s1; red[w * 32 + 16 + lq * 4 + e] = s2; }
    }
    __syncthreads();
    if (tid < 32) part[(size_t)blockIdx.x * 32 + tid] = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid];
  }
}

// ------------------------------------------------------------------------------------
// conv2 forward: Conv1d(16,32,k5,s2,p2), NLC in / NLC out + BN partials — with stage 1's BatchNorm + ReLU + MaxPool fused into
// its staging (round 2).  As two kernels, p1 was written by bn_relu_pool<16> and read back by conv2_fwd (0.31 + 0.23 ms,
// 2.56 GB); here an item stages the y1 rows of its chunk
// (4 t0 - 5 .. 4 t0 + 513) normalised into LDS, pools them into the p1 rows the convolution needs (2 t0 - 2 .. 2 t0 + 256), writes the
// rows it OWNS (2 t0 .. 2 t0 + 255: every p1 row has exactly one owner) with their pooling codes for the backward pass, and runs
// the convolution from the pooled tile: 2.03 GB, 0.47 ms.  Same expressions as bn_relu_pool_kernel -> the same p1 and codes bit for bit.
// ------------------------------------------------------------------------------------
#define C2_CHUNK 128            // output positions per work item (4 waves x 2 blocks of 16)
#define C2_ROWS (2 * C2_CHUNK + 3)
#define C2_PS 20                // LDS row stride (floats) of a 16-channel row
#define PC_YROWS (2 * C2_ROWS + 1)
__device__ __forceinline__ int first_argmax3(float l, float c, float r);
__global__ __launch_bounds__(256, 2) void pool1_conv2_fwd_kernel(const float* __restrict__ y1, const float* __restrict__ stat1,
                                                                 float* __restrict__ p1, uint8_t* __restrict__ code1,
                                                                 const float* __restrict__ w2, float* __restrict__ y2,
                                                                 float* __restrict__ part, int B, int L1, int P1, int L2,
                                                                 int want_stats, const FoldCtx fc) {
  FOLD_BEGIN; FS(y1); FS(stat1); FS(p1); FS(code1); FS(w2); FS(y2); FS(part);
  __shared__ __attribute__((aligned(16))) float zs[PC_YROWS * C2_PS];     // bn1(y1) rows, row r <-> position 4 t0 - 5 + r
  __shared__ __attribute__((aligned(16))) float ps[C2_ROWS * C2_PS];      // p1 rows, row i <-> position 2 t0 - 2 + i
  __shared__ float red[4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  float A[2][20];
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int m = 0; m < 20; ++m) A[ob][m] = w2[((ob * 16 + li) * 16 + lq * 4 + (m & 3)) * 5 + (m >> 2)];
  const int nchunk = (L2 + C2_CHUNK - 1) / C2_CHUNK;
  const int nitems = B * nchunk;
  f32x4 ssum[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ssq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const float NINF = -__builtin_huge_valf();
  // every piece a thread stages has channel quad c4 = tid & 3
  const int c4s = tid & 3;
  const float4 sc4 = *(const float4*)(stat1 + 2 * 16 + c4s * 4), sh4 = *(const float4*)(stat1 + 3 * 16 + c4s * 4);
  const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
  constexpr int NY4 = (PC_YROWS * 4 + 255) / 256, NP4 = (C2_ROWS * 4 + 255) / 256;
  float4 yr[NY4];
  auto prefetch = [&](int item) {
    const int b = item / nchunk, t0 = (item - b * nchunk) * C2_CHUNK, ty0 = 4 * t0 - 5;
#pragma unroll
    for (int j = 0; j < NY4; ++j) {
      const int i = tid + 256 * j, row = i >> 2, ty = ty0 + row;
      const int tc = ty < 0 ? 0 : (ty > L1 - 1 ? L1 - 1 : ty);             // unconditional, clamped load
      yr[j] = *(const float4*)(y1 + ((size_t)b * L1 + tc) * 16 + c4s * 4);
    }
  };
  if ((int)blockIdx.x < nitems) prefetch(blockIdx.x);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / nchunk, t0 = (item - b * nchunk) * C2_CHUNK;
    const int base = 2 * t0 - 2, ty0 = 4 * t0 - 5;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NY4; ++j) {
      const int i = tid + 256 * j, row = i >> 2, ty = ty0 + row;
      if (i < PC_YROWS * 4) {
        const bool ok = ty >= 0 && ty < L1;
        const float qv[4] = {yr[j].x, yr[j].y, yr[j].z, yr[j].w};
        float z[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = ok ? qv[e] * scv[e] + shv[e] : NINF;
        *(float4*)&zs[row * C2_PS + c4s * 4] = make_float4(z[0], z[1], z[2], z[3]);
      }
    }
    __syncthreads();
    if (item + (int)gridDim.x < nitems) prefetch(item + gridDim.x);
#pragma unroll
    for (int j = 0; j < NP4; ++j) {
      const int i = tid + 256 * j, row = i >> 2, pp = base + row;
      if (i < C2_ROWS * 4) {
        float best[4] = {0.f, 0.f, 0.f, 0.f};                               // rows outside [0, P1): the convolution's zero padding
        if (pp >= 0 && pp < P1) {
          const float4 l4 = *(const float4*)&zs[(2 * row) * C2_PS + c4s * 4], m4 = *(const float4*)&zs[(2 * row + 1) * C2_PS + c4s * 4],
                       r4 = *(const float4*)&zs[(2 * row + 2) * C2_PS + c4s * 4];
          const float zl[4] = {l4.x, l4.y, l4.z, l4.w}, zc[4] = {m4.x, m4.y, m4.z, m4.w}, zr[4] = {r4.x, r4.y, r4.z, r4.w};
          unsigned cd = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int win = first_argmax3(zl[e], zc[e], zr[e]);
            const float m = win == 0 ? zl[e] : (win == 1 ? zc[e] : zr[e]);
            best[e] = fmaxf(m, 0.f);
            cd |= (unsigned)(m > 0.f ? win : 3) << (2 * e);
          }
          if (row >= 2 && row < 2 + 2 * C2_CHUNK) {                         // the rows this item owns
            *(float4*)(p1 + ((size_t)b * P1 + pp) * 16 + c4s * 4) = make_float4(best[0], best[1], best[2], best[3]);
            if (code1) code1[((size_t)b * P1 + pp) * 4 + c4s] = (uint8_t)cd;
          }
        }
        *(float4*)&ps[row * C2_PS + c4s * 4] = make_float4(best[0], best[1], best[2], best[3]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int pbi = 0; pbi < 2; ++pbi) {
      const int pl = (w * 2 + pbi) * 16 + li;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 5; ++kk) {
        const float4 q = *(const float4*)&ps[(2 * pl + kk) * C2_PS + lq * 4];
        acc0 = mfma16(A[0][4 * kk + 0], q.x, acc0); acc1 = mfma16(A[1][4 * kk + 0], q.x, acc1);
        acc0 = mfma16(A[0][4 * kk + 1], q.y, acc0); acc1 = mfma16(A[1][4 * kk + 1], q.y, acc1);
        acc0 = mfma16(A[0][4 * kk + 2], q.z, acc0); acc1 = mfma16(A[1][4 * kk + 2], q.z, acc1);
        acc0 = mfma16(A[0][4 * kk + 3], q.w, acc0); acc1 = mfma16(A[1][4 * kk + 3], q.w, acc1);
      }
      const int t = t0 + pl;
      if (t < L2) {
        float* dst = y2 + ((size_t)b * L2 + t) * 32 + lq * 4;
        *(float4*)dst = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
        *(float4*)(dst + 16) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ssum[0][e] += acc0[e]; ssq[0][e] += acc0[e] * acc0[e];
          ssum[1][e] += acc1[e]; ssq[1][e] += acc1[e] * acc1[e];
        }
      }
    }
  }
  if (want_stats) {
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s1 = ssum[ob][e], s2 = ssq[ob][e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (li == 0) { red[w * 64 + ob * 16 + lq * 4 + e] = s1; red[w * 64 + 32 + ob * 16 + lq * 4 + e] = s2; }
      }
    __syncthreads();
    if (tid < 64) part[(size_t)blockIdx.x * 64 + tid] = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
  }
}

// Column sums of part[nrows][ncol] (ncol <= 64) by one 1024-thread workgroup into red[0..ncol): 1024/ncol
// row-lanes, 8 loads in flight per lane, fixed combination order (deterministic for a given nrows).
#define FIN_THREADS 1024
__device__ __forceinline__ void fin_colsums(const float* __restrict__ part, int nrows, int ncol, double* red) {
  const int tid = threadIdx.x;
  const int col = tid % ncol, grp = tid / ncol, ngrp = FIN_THREADS / ncol;
  double a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.0;
  if (grp < ngrp) {
    int r = grp;
    for (; r + 7 * ngrp < nrows; r += 8 * ngrp) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(r + j * ngrp) * ncol + col];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += (double)v[j];
    }
    for (; r < nrows; r += ngrp) a[0] += (double)part[(size_t)r * ncol + col];
  }
  red[tid] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  double s = 0.0;
  if (tid < ncol)
    for (int g = 0; g < ngrp; ++g) s += red[g * ncol + tid];
  __syncthreads();
  if (tid < ncol) red[tid] = s;
  __syncthreads();
}

// ------------------------------------------------------------------------------------
// BatchNorm statistics -> (mean, invstd, scale, shift); running-stat update
// part: [nrows][2*CH] = per-workgroup (sum[CH], sumsq[CH]).  stat: 4*CH floats.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(FIN_THREADS) void bn_finalize_kernel(const float* __restrict__ part, int nrows, int CH, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var,
                                                          int64_t* __restrict__ nbt, float momentum, float eps,
                                                          int training, float* __restrict__ stat, const FoldCtx fc) {
  FOLD_BEGIN; FS(part); FS(gamma); FS(beta); FS(run_mean); FS(run_var); FS(nbt); FS(stat);
  __shared__ double red[FIN_THREADS];
  const int tid = threadIdx.x;
  if (training) {
    fin_colsums(part, nrows, 2 * CH, red);          // 2*CH <= 64 columns
    if (tid < CH) {
      const double mean = red[tid] / count;
      double var = red[CH + tid] / count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      const float sc = gamma[tid] * invstd;
      stat[tid] = (float)mean;
      stat[CH + tid] = invstd;
      stat[2 * CH + tid] = sc;
      stat[3 * CH + tid] = beta[tid] - (float)mean * sc;
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      run_mean[tid] = (1.0f - momentum) * run_mean[tid] + momentum * (float)mean;
      run_var[tid] = (1.0f - momentum) * run_var[tid] + momentum * (float)unbiased;
    }
    if (tid == 0) nbt[0] += 1;
  } else if (tid < CH) {
    const float invstd = 1.0f / sqrtf(run_var[tid] + eps);
    const float sc = gamma[tid] * invstd;
    stat[tid] = run_mean[tid];
    stat[CH + tid] = invstd;
    stat[2 * CH + tid] = sc;
    stat[3 * CH + tid] = beta[tid] - run_mean[tid] * sc;
  }
}

// ------------------------------------------------------------------------------------
// BN apply + ReLU + MaxPool1d(3,2,1) on NLC data (models.py:47-49, 51-53)
// one thread = one pooled position x 4 channels
// ------------------------------------------------------------------------------------
// Pooling decision of a window, 2 bits: 0 / 1 / 2 = its left / centre / right candidate is the FIRST maximum (ATen's
// max_pool1d tie rule) and is positive, 3 = the maximum is <= 0 (ReLU passes no gradient).  Written by the forward pass in
// training (one byte per (window, 4 channels)), it lets every backward consumer route dP to dz = dL/d(bn output) on the fly:
// the (B, L, CH) dz tensors are never written or read (stage 1: 1.0 GB written + 1.0 GB read per step at B = 8192).
__device__ __forceinline__ int first_argmax3(float l, float c, float r) {
  // returns 0 (left) / 1 (centre) / 2 (right); invalid candidates are passed as -inf
  int a = 0; float m = l;
  if (c > m) { a = 1; m = c; }
  if (r > m) { a = 2; }
  return a;
}

template <int CH>
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const float* __restrict__ y, const float* __restrict__ stat,
                                                           float* __restrict__ p, uint8_t* __restrict__ code, int B, int L, int P,
                                                           const FoldCtx fc) {
  FOLD_BEGIN; FS(y); FS(stat); FS(p); FS(code);
  constexpr int C4 = CH / 4;
  const int64_t total = (int64_t)B * P * C4;
  const float NINF = -__builtin_huge_valf();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int64_t bp = i / C4;
    const int pp = (int)(bp % P), b = (int)(bp / P);
    const float4 sc = *(const float4*)(stat + 2 * CH + c4 * 4), sh = *(const float4*)(stat + 3 * CH + c4 * 4);
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    float z[3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int t = 2 * pp - 1 + j;
      const bool ok = t >= 0 && t < L;
      const float4 q = *(const float4*)(y + ((size_t)b * L + (ok ? t : 2 * pp)) * CH + c4 * 4);      // clamped, unconditional
      const float qv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) z[j][e] = ok ? qv[e] * scv[e] + shv[e] : NINF;
    }
    float best[4];
    unsigned cd = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int win = first_argmax3(z[0][e], z[1][e], z[2][e]);
      const float m = win == 0 ? z[0][e] : (win == 1 ? z[1][e] : z[2][e]);
      best[e] = fmaxf(m, 0.f);
      cd |= (unsigned)(m > 0.f ? win : 3) << (2 * e);
    }
    *(float4*)(p + ((size_t)b * P + pp) * CH + c4 * 4) = make_float4(best[0], best[1], best[2], best[3]);
    if (code) code[((size_t)b * P + pp) * C4 + c4] = (uint8_t)cd;
  }
}

// dz = dL/d(bn output) at position t, channels 4 c4 .. +3, routed from dP through the recorded pooling decisions: position
// t = 2 ph is the centre of window ph; t = 2 ph + 1 is the right candidate of window ph and the left one of window ph + 1.
// Split in two so that a software pipeline can issue the loads (RoutedRaw) an item ahead of their use.
struct RoutedRaw { float4 g0, g1; unsigned c0, c1; };
template <int CH>
__device__ __forceinline__ RoutedRaw routed_load(const float* __restrict__ dp, const uint8_t* __restrict__ code, int b, int t, int P, int c4) {
  constexpr int C4 = CH / 4;
  const int ph = t >> 1, ph1 = (ph + 1 < P) ? ph + 1 : ph;       // clamped: the second window only counts for odd t with ph + 1 < P
  RoutedRaw r;
  r.g0 = *(const float4*)(dp + ((size_t)b * P + ph) * CH + c4 * 4);
  r.g1 = *(const float4*)(dp + ((size_t)b * P + ph1) * CH + c4 * 4);
  r.c0 = code[((size_t)b * P + ph) * C4 + c4];
  r.c1 = code[((size_t)b * P + ph1) * C4 + c4];
  return r;
}
__device__ __forceinline__ float4 routed_dz(const RoutedRaw& r, int t, int P) {
  const bool odd = t & 1, has1 = odd && ((t >> 1) + 1 < P);
  const unsigned want0 = odd ? 2u : 1u;
  const float g0[4] = {r.g0.x, r.g0.y, r.g0.z, r.g0.w}, g1[4] = {r.g1.x, r.g1.y, r.g1.z, r.g1.w};
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    o[e] = (((r.c0 >> (2 * e)) & 3u) == want0) ? g0[e] : 0.f;
    if (has1 && ((r.c1 >> (2 * e)) & 3u) == 0u) o[e] += g1[e];
  }
  return make_float4(o[0], o[1], o[2], o[3]);
}

// ====================================================================================
// Backward
// ====================================================================================
// MaxPool + ReLU + BatchNorm backward of stage 2, pass 1: the per-channel sums of dz and dz * xhat that BatchNorm's backward needs, dz
// routed from dP by the forward pass's pooling decisions.  One thread per POSITION x 4 channels; dP2 arrives as two tensors (the
// directions of GRU layer 0) and dz2 is stored for conv2_bwd (measured when conv2's dX and dW were two kernels: routing it twice
// more cost them +0.08 ms, more than the 0.25 GB saved).  Stage 1 has no such pass any more: conv1_bwd sums while it contracts.
template <int CH>
__global__ __launch_bounds__(256) void pool_bn_bwd_pass1(const float* __restrict__ dp_a, const float* __restrict__ dp_b,
                                                         const uint8_t* __restrict__ code, const float* __restrict__ y,
                                                         const float* __restrict__ stat, float* __restrict__ dz,
                                                         float* __restrict__ part, int B, int L, int P, const FoldCtx fc) {
  FOLD_BEGIN; FS(dp_a); FS(dp_b); FS(code); FS(y); FS(stat); FS(dz); FS(part);
  constexpr int C4 = CH / 4;
  __shared__ float red[256 * 8];
  const int c4 = threadIdx.x % C4;             // grid stride is a multiple of C4: a thread keeps its channels
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float mean[4], invstd[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { mean[e] = stat[c4 * 4 + e]; invstd[e] = stat[CH + c4 * 4 + e]; }
  const int64_t total = (int64_t)B * L * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t bt = i / C4;
    const int t = (int)(bt % L), b = (int)(bt / L);
    RoutedRaw r = routed_load<CH>(dp_a, code, b, t, P, c4);
    if (dp_b) {
      const RoutedRaw r2 = routed_load<CH>(dp_b, code, b, t, P, c4);
      r.g0.x += r2.g0.x; r.g0.y += r2.g0.y; r.g0.z += r2.g0.z; r.g0.w += r2.g0.w;
      r.g1.x += r2.g1.x; r.g1.y += r2.g1.y; r.g1.z += r2.g1.z; r.g1.w += r2.g1.w;
    }
    const float4 d = routed_dz(r, t, P);
    const float4 yq = *(const float4*)(y + ((size_t)b * L + t) * CH + c4 * 4);
    *(float4*)(dz + ((size_t)b * L + t) * CH + c4 * 4) = d;
    const float dv[4] = {d.x, d.y, d.z, d.w}, yv[4] = {yq.x, yq.y, yq.z, yq.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[e] += dv[e]; s2[e] += dv[e] * (yv[e] - mean[e]) * invstd[e]; }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[threadIdx.x * 8 + e] = s1[e]; red[threadIdx.x * 8 + 4 + e] = s2[e]; }
  __syncthreads();
  if (threadIdx.x < 2 * CH) {
    // column layout of the partial: [sum dz (CH)][sum dz*xhat (CH)]
    const int which = threadIdx.x / CH, ch = threadIdx.x % CH, cc4 = ch / 4, e = ch % 4;
    float acc = 0.f;
    for (int t = cc4; t < 256; t += C4) acc += red[t * 8 + which * 4 + e];
    part[(size_t)blockIdx.x * 2 * CH + threadIdx.x] = acc;
  }
}

// sums -> c1 = mean(dz), c2 = mean(dz*xhat); also d(gamma), d(beta)
__global__ __launch_bounds__(FIN_THREADS) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nrows, int CH, double count,
                                                              float* __restrict__ cstat, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, const FoldCtx fc) {
  FOLD_BEGIN; FS(part); FS(cstat); FS(dgamma); FS(dbeta);
  __shared__ double red[FIN_THREADS];
  const int tid = threadIdx.x, ncol = 2 * CH;
  fin_colsums(part, nrows, ncol, red);
  if (tid < ncol) {
    const double s = red[tid];
    if (tid < CH) { dbeta[tid] = (float)s; cstat[tid] = (float)(s / count); }
    else { dgamma[tid - CH] = (float)s; cstat[tid] = (float)(s / count); }
  }
}

// ------------------------------------------------------------------------------------
// conv2 backward, both gradients in one kernel (one staging of dy2 serves dX and dW):
//   wrt input:   dp1[pos][c] = sum_{o,kk} dy2[(pos+2-kk)/2][o] w2[o][c][kk]
//                even pos = 2u uses kk in {0,2,4} (t = u+1, u, u-1); odd pos = 2u+1 uses kk in {1,3} (t = u+1, u)
//   wrt weights: dW2[o][c][kk] = sum_{b,t} dy2[b][t][o] p1[b][2t+kk-2][c]
// ------------------------------------------------------------------------------------
#define D2_UCH 128              // t (= u) values per item (4 waves x 2 blocks of 16)
#define D2_ROWS (D2_UCH + 2)
#define D2_PS 36
#define W2_PROWS (2 * D2_UCH + 3)

// dy2 is not read from memory: the kernel stages dz2 (the gradient w.r.t. the BatchNorm-2 OUTPUT) together with y2 and
// applies BatchNorm's second backward pass, dy = scale * (dz - c1 - xhat * c2), on the way into LDS — as conv1_bwd does
// for stage 1 — so the separate elementwise pass over the (B, L2, 32) tensor (1.5 GB of traffic, one launch) is gone.
// Round 2: dX and dW were two kernels that each staged the same dz2 / y2 rows (0.32 + 0.32 ms, 1.0 GB read twice); an item is
// now one 128-step chunk of one window for both — dy2 rows t0-1 .. t0+128 (dX needs the halo) and p1 rows 2 t0 - 2 .. 2 t0 + 256.
// The item -> workgroup map and the accumulation order of dW are the former conv2_bwd_dw's, so dW2 keeps its bits.
__global__ __launch_bounds__(256) void conv2_bwd_kernel(const float* __restrict__ dz2, const float* __restrict__ y2,
                                                        const float* __restrict__ stat, const float* __restrict__ cstat,
                                                        const float* __restrict__ w2, const float* __restrict__ p1,
                                                        float* __restrict__ dp1, float* __restrict__ part, int B, int P1, int L2,
                                                        const FoldCtx fc) {
  FOLD_BEGIN; FS(dz2); FS(y2); FS(stat); FS(cstat); FS(w2); FS(p1); FS(dp1); FS(part);
  __shared__ __attribute__((aligned(16))) float ds_[D2_ROWS * D2_PS];     // dy2: row i <-> t = t0 - 1 + i
  __shared__ __attribute__((aligned(16))) float ps[W2_PROWS * C2_PS];     // p1:  row i <-> pos = 2 t0 - 2 + i
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  // dX A operands (rows = input channel c = li):  k-step m: o = lq*8 + (m&7), tap index m>>3
  float Ae[24], Ao[16];
#pragma unroll
  for (int m = 0; m < 24; ++m) Ae[m] = w2[((lq * 8 + (m & 7)) * 16 + li) * 5 + 2 * (m >> 3)];
#pragma unroll
  for (int m = 0; m < 16; ++m) Ao[m] = w2[((lq * 8 + (m & 7)) * 16 + li) * 5 + 1 + 2 * (m >> 3)];
  f32x4 acc[2][5];
#pragma unroll
  for (int ob = 0; ob < 2; ++ob)
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) acc[ob][kk] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nchunk = (L2 + D2_UCH - 1) / D2_UCH;
  const int nitems = B * nchunk;
  // BatchNorm constants of the four channels this thread stages (c4 = tid & 7 for every piece it handles)
  float bn_mean[4], bn_inv[4], bn_sc[4], bn_c1[4], bn_c2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ch = (tid & 7) * 4 + e;
    bn_mean[e] = stat[ch]; bn_inv[e] = stat[32 + ch]; bn_sc[e] = stat[64 + ch]; bn_c1[e] = cstat[ch]; bn_c2[e] = cstat[32 + ch];
  }
  // software pipeline: the next item's rows are loaded into registers while this item's MFMAs run
  constexpr int ND4 = (D2_ROWS * 8 + 255) / 256, NP4 = (W2_PROWS * 4 + 255) / 256;
  float4 dr[ND4], yr[ND4], qr[NP4];
  auto prefetch = [&](int item) {
    const int b = item / nchunk, t0 = (item - b * nchunk) * D2_UCH, base = 2 * t0 - 2;
#pragma unroll
    for (int j = 0; j < ND4; ++j) {
      const int i = tid + 256 * j, row = i >> 3, c4 = i & 7, t = t0 - 1 + row;
      const int tc = t < 0 ? 0 : (t > L2 - 1 ? L2 - 1 : t);               // unconditional, clamped loads
      dr[j] = *(const float4*)(dz2 + ((size_t)b * L2 + tc) * 32 + c4 * 4);
      yr[j] = *(const float4*)(y2 + ((size_t)b * L2 + tc) * 32 + c4 * 4);
    }
#pragma unroll
    for (int j = 0; j < NP4; ++j) {
      const int i = tid + 256 * j, row = i >> 2, c4 = i & 3, src = base + row;
      const int sc = src < 0 ? 0 : (src > P1 - 1 ? P1 - 1 : src);
      qr[j] = *(const float4*)(p1 + ((size_t)b * P1 + sc) * 16 + c4 * 4);
    }
  };
  if ((int)blockIdx.x < nitems) prefetch(blockIdx.x);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / nchunk, t0 = (item - b * nchunk) * D2_UCH;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ND4; ++j) {
      const int i = tid + 256 * j, row = i >> 3, c4 = i & 7, t = t0 - 1 + row;
      if (i < D2_ROWS * 8) {
        const float4 dzv = dr[j];
        float4 q;
        q.x = bn_sc[0] * (dzv.x - bn_c1[0] - (yr[j].x - bn_mean[0]) * bn_inv[0] * bn_c2[0]);
        q.y = bn_sc[1] * (dzv.y - bn_c1[1] - (yr[j].y - bn_mean[1]) * bn_inv[1] * bn_c2[1]);
        q.z = bn_sc[2] * (dzv.z - bn_c1[2] - (yr[j].z - bn_mean[2]) * bn_inv[2] * bn_c2[2]);
        q.w = bn_sc[3] * (dzv.w - bn_c1[3] - (yr[j].w - bn_mean[3]) * bn_inv[3] * bn_c2[3]);
        if (t < 0 || t >= L2) q = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)&ds_[row * D2_PS + c4 * 4] = q;
      }
    }
    const int base = 2 * t0 - 2;
#pragma unroll
    for (int j = 0; j < NP4; ++j) {
      const int i = tid + 256 * j, row = i >> 2, c4 = i & 3, src = base + row;
      if (i < W2_PROWS * 4) {
        float4 q = qr[j];
        if (src < 0 || src >= P1) q = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)&ps[row * C2_PS + c4 * 4] = q;
      }
    }
    __syncthreads();
    if (item + (int)gridDim.x < nitems) prefetch(item + gridDim.x);
    // ---- dX: two blocks of 16 u per wave
#pragma unroll
    for (int ubi = 0; ubi < 2; ++ubi) {
      const int ul = (w * 2 + ubi) * 16 + li;       // local u; LDS row of t=u is ul+1
      f32x4 ae0 = {0.f, 0.f, 0.f, 0.f}, ae1 = {0.f, 0.f, 0.f, 0.f}, ao0 = {0.f, 0.f, 0.f, 0.f}, ao1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ti = 0; ti < 3; ++ti) {              // tap index ti <-> t = u + 1 - ti
        const float* rowp = &ds_[(ul + 2 - ti) * D2_PS + lq * 8];
        const float4 q0 = *(const float4*)rowp, q1 = *(const float4*)(rowp + 4);
        const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int mm = 0; mm < 8; ++mm) {
          if (mm & 1) ae1 = mfma16(Ae[ti * 8 + mm], qv[mm], ae1); else ae0 = mfma16(Ae[ti * 8 + mm], qv[mm], ae0);
          if (ti < 2) { if (mm & 1) ao1 = mfma16(Ao[ti * 8 + mm], qv[mm], ao1); else ao0 = mfma16(Ao[ti * 8 + mm], qv[mm], ao0); }
        }
      }
      const int u = t0 + ul;
      if (2 * u < P1)
        *(float4*)(dp1 + ((size_t)b * P1 + 2 * u) * 16 + lq * 4) = make_float4(ae0[0] + ae1[0], ae0[1] + ae1[1], ae0[2] + ae1[2], ae0[3] + ae1[3]);
      if (2 * u + 1 < P1)
        *(float4*)(dp1 + ((size_t)b * P1 + 2 * u + 1) * 16 + lq * 4) = make_float4(ao0[0] + ao1[0], ao0[1] + ao1[1], ao0[2] + ao1[2], ao0[3] + ao1[3]);
    }
    // ---- dW: each wave contracts its 32 t's (8 k-steps of 4) of the chunk
#pragma unroll 2
    for (int m = 0; m < D2_UCH / 16; ++m) {
      const int tl = w * 32 + 4 * m + lq;
      const float a0 = ds_[(tl + 1) * D2_PS + li], a1 = ds_[(tl + 1) * D2_PS + 16 + li];
#pragma unroll
      for (int kk = 0; kk < 5; ++kk) {
        const float bv = ps[(2 * tl + kk) * C2_PS + li];
        acc[0][kk] = mfma16(a0, bv, acc[0][kk]);
        acc[1][kk] = mfma16(a1, bv, acc[1][kk]);
      }
    }
  }
  // dW: cross-wave reduction through LDS, wave by wave, then one partial row per workgroup in w2's own layout
  __syncthreads();
  float* red = ds_;    // 2560 floats
  for (int ww = 0; ww < 4; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int kk = 0; kk < 5; ++kk)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int o = ob * 16 + lq * 4 + e, c = li;
            const int idx = (o * 16 + c) * 5 + kk;
            red[idx] = (ww == 0 ? 0.f : red[idx]) + acc[ob][kk][e];
          }
    }
    __syncthreads();
  }
  for (int i = tid; i < 2560; i += 256) part[(size_t)blockIdx.x * 2560 + i] = red[i];
}

// ------------------------------------------------------------------------------------
// conv1 backward: per window G[o][c,kk] = sum_t dy1[t][o] x[c][2t+kk-3]; then
//   dW1[o][c][kk] += s[b,c] * G      and      ds[b,c] = sum_{o,kk} w1[o][c][kk] * G
// ------------------------------------------------------------------------------------
#define G1_TCH 256
#ifndef CONV1_BWD_PIPE
#define CONV1_BWD_PIPE 1     // software pipeline: the next chunk's x / dP1 / pooling codes / y1 are loaded into registers under this chunk's MFMAs
#endif
#define CONV1_BWD_WGS (CONV1_BWD_PIPE ? 2 : 4)
#define G1_MAXNB 7              // ceil(16*7/16)
// LDS row stride of a channel's samples: 12 mod 32 dwords, so the (up to three) channels a 16-column block of the B operand touches
// fall into disjoint bank windows — columns kk = 0..6 and the two positions 2 lq of a 32-lane group span 9 banks per channel
// (C1_XW = 520 = 8 mod 32 made neighbouring channels overlap: 33 % of the LDS cycles were bank conflicts, r02_pmc_lds_B8192.csv).
#define G1_XS (C1_XW + 4)
// dz / xhat rows live in quads of 4 positions x 16 channels = 64 dwords, 80 dwords apart: an MFMA A-operand read (4 positions x 16
// channels per wave-instruction) stays inside one quad, and the 16-byte stores of two neighbouring quads (one 8-lane store group)
// fall into different halves of the 32 store banks.
#define G1_QS 80
__device__ __forceinline__ int g1_row(int r) { return (r >> 2) * G1_QS + (r & 3) * 16; }

// Round 3: the BatchNorm-backward sums of stage 1 are no longer a pass of their own (pool_bn_bwd_pass1<16>: 1.56 GB of the step's
// traffic, all of it re-read here).  dy1 = scale * (dz - c1 - xhat * c2) is LINEAR in the two sums c1 = mean(dz), c2 = mean(dz * xhat),
// so the correlation splits into pieces that do not need them:
//   G[o][j] = scale[o] * ( Gdz[o][j] - c1[o] * Sx[j] - c2[o] * Gxh[o][j] ),
//   Gdz = sum_t dz[t][o] xp[t][j],   Gxh = sum_t xhat[t][o] xp[t][j],   Sx = sum_{t < L1} xp[t][j],   xp[t][j = (c,kk)] = x[c][2t+kk-3].
// This kernel walks whole windows (persistent over b), stages dz (dP1 routed by the forward pass's pooling decisions) and
// xhat = (y1 - mean) * invstd per 256-position chunk, contracts BOTH against the same B operand (two MFMAs per LDS read of x),
// sums dz and dz * xhat per channel on the side (-> the partials bn_bwd_finalize turns into c1, c2, d gamma, d beta) and stores the
// window's Gdz and Gxh (6 KB per window at C = 6).  conv1_bwd_fin then combines them with c1, c2 and Sx (which is a sum of samples of
// one parity less a few at the window's edges: the gate kernel leaves the two parity sums) and forms dW1 and ds.
// Neither dy1 nor dz1 ever exists in HBM.
// fp32 MFMA shares its pipe with the VALU, so every VALU instruction here costs matrix time (the first version of this kernel
// issued 450 per chunk and wave for staging — 24 separately addressed loads, 2-bit decisions decoded per position — and 14 per MFMA
// k-step).  Now a thread stages a QUAD of positions (4 t x 4 channels): three dP1 vectors, three decision bytes and four y1 vectors
// from two base addresses with immediate offsets, each decision decoded once; the k loop is fully unrolled (LDS addresses are
// immediates) and carries only the two channel sums.
// A small batch (the reference's B = 64) has too few windows to fill the chip with one workgroup per window: a window is then cut into
// SEG segments of CPS chunks, each with its own record, and conv1_bwd_fin adds a window's records in segment order.  SEG depends on
// the shape only (never on the fold count of a launch), so a fold's bits do not depend on its companions.
__host__ __device__ inline int conv1_bwd_cps(int B, int nchunk) { return B >= 256 ? nchunk : (nchunk + 7) / 8; }
__host__ __device__ inline int conv1_bwd_segs(int B, int nchunk) { const int cps = conv1_bwd_cps(B, nchunk); return (nchunk + cps - 1) / cps; }
template <int CT>
__global__ __launch_bounds__(256, CONV1_BWD_WGS) void conv1_bwd_kernel(const float* __restrict__ dp1, const uint8_t* __restrict__ code1,
                                                        const float* __restrict__ y1, const float* __restrict__ stat,
                                                        const float* __restrict__ x, float* __restrict__ g1w,
                                                        float* __restrict__ bpart, int B, int Crt,
                                                        int T, int L1, int P1, const FoldCtx fc) {
  FOLD_BEGIN; FS(dp1); FS(code1); FS(y1); FS(stat); FS(x); FS(g1w); FS(bpart);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = CT > 0 ? CT : Crt;
  const int K = C * 7, NB = (K + 15) / 16, NB16 = NB * 16;
  constexpr int NBC = CT > 0 ? (CT * 7 + 15) / 16 : G1_MAXNB;
  float* xs = smem;                          // [C][G1_XS] natural order, sample j <-> x[2*t0 - 4 + j]
  float* dzs = xs + C * G1_XS;               // [G1_TCH / 4 quads][G1_QS] dz
  float* xhs = dzs + (G1_TCH / 4) * G1_QS;   // same layout, xhat
  float* red = smem;                         // window end: [4 waves][2][16][NB16]; aliases the staging area (the host sizes smem for both)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lq = lane >> 4;
  int coff[NBC];
  // per-thread BN constants for the 4 channels it stages (c4 = tid & 3)
  float bn_mean[4], bn_inv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ch = (tid & 3) * 4 + e;
    bn_mean[e] = stat[ch]; bn_inv[e] = stat[16 + ch];
  }
#pragma unroll
  for (int nb = 0; nb < NBC; ++nb) {
    const int col = nb * 16 + li, cc = col < K ? col : K - 1, c = cc / 7, kk = cc - 7 * c;   // padding columns alias the last real one (LDS broadcast, result unused)
    coff[nb] = c * G1_XS + kk + 1 + 2 * (w * 64 + lq);       // B operand of k-step m: xs[coff + 8 m]
  }
  const int aoff = (w * 16) * G1_QS + lq * 16 + li;          // A operand of k-step m: dzs / xhs[aoff + G1_QS m]
  float s1 = 0.f, s2 = 0.f;                  // this lane's share of sum dz, sum dz*xhat for channel li
  const int nchunk = (L1 + G1_TCH - 1) / G1_TCH;
  // Register prefetch of the NEXT chunk, staged by quads (compile-time channel count, T % 8 == 0, i.e. L1 % 4 == 0 and P1 = L1 / 2):
  // without it a chunk is load -> wait -> LDS -> sync -> MFMA with nothing but the other resident workgroups to hide the wait.
  constexpr bool PIPE_OK = CONV1_BWD_PIPE && CT > 0;
  const bool pipe = PIPE_OK && (T & 7) == 0 && 2 * P1 == L1 && L1 >= 8;
  constexpr int NX4 = PIPE_OK ? (CT * (C1_XW / 4) + 255) / 256 : 1;
  const int sq = tid >> 2, sc4 = tid & 3;    // staging role: quad sq of the chunk, channels 4 sc4 .. +3
  float4 xr[NX4], yq[4], gq[3];
  unsigned cq[3];
  auto prefetch = [&](int b, int ch) {
    const int t0 = ch * G1_TCH, g_base = 2 * t0 - 4;
    const float* xb = x + (size_t)b * C * T;
#pragma unroll
    for (int j = 0; j < NX4; ++j) {
      const int i = tid + 256 * j, ic = i < C * (C1_XW / 4) ? i : 0;
      const int c = ic / (C1_XW / 4), i4 = ic - c * (C1_XW / 4), g0 = g_base + 4 * i4;
      const int gc = g0 < 0 ? 0 : (g0 > T - 4 ? T - 4 : g0);          // unconditional, clamped load
      xr[j] = *(const float4*)(xb + (unsigned)(c * T + gc));
    }
    const int tq = t0 + 4 * sq, tqc = tq < L1 - 4 ? tq : L1 - 4;       // a quad is wholly inside or wholly outside the window
    const int ph = tqc >> 1, ph2 = ph + 2 < P1 ? ph + 2 : P1 - 1;
    const float* dpb = dp1 + (size_t)b * P1 * 16;
    const uint8_t* cb = code1 + (size_t)b * P1 * 4;
    const float* yb = y1 + (size_t)b * L1 * 16;
    const unsigned o = (unsigned)(ph * 16 + sc4 * 4), oc = (unsigned)(ph * 4 + sc4), oy = (unsigned)(tqc * 16 + sc4 * 4);
    gq[0] = *(const float4*)(dpb + o); gq[1] = *(const float4*)(dpb + o + 16); gq[2] = *(const float4*)(dpb + (unsigned)(ph2 * 16 + sc4 * 4));
    cq[0] = cb[oc]; cq[1] = cb[oc + 4]; cq[2] = cb[(unsigned)(ph2 * 4 + sc4)];
#pragma unroll
    for (int j = 0; j < 4; ++j) yq[j] = *(const float4*)(yb + oy + 16 * j);
  };
  const int CPS = conv1_bwd_cps(B, nchunk), SEG = conv1_bwd_segs(B, nchunk), nitems = B * SEG;
  if (pipe && (int)blockIdx.x < nitems) prefetch(blockIdx.x / SEG, (blockIdx.x % SEG) * CPS);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / SEG, ch_lo = (item - b * SEG) * CPS, ch_hi = ch_lo + CPS < nchunk ? ch_lo + CPS : nchunk;
    f32x4 acc1[NBC], acc2[NBC];
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb) { acc1[nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const float* xb = x + (size_t)b * C * T;
    for (int ch = ch_lo; ch < ch_hi; ++ch) {
      const int t0 = ch * G1_TCH;
      __syncthreads();
      if (pipe) {
        const int g_base = 2 * t0 - 4;
#pragma unroll
        for (int j = 0; j < NX4; ++j) {
          const int i = tid + 256 * j;
          if (i < C * (C1_XW / 4)) {
            const int c = i / (C1_XW / 4), i4 = i - c * (C1_XW / 4), g0 = g_base + 4 * i4;
            float4 q = xr[j];
            if (g0 < 0 || g0 > T - 4) q = make_float4(0.f, 0.f, 0.f, 0.f);      // zero padding (whole vectors: T % 4 == 0)
            *(float4*)&xs[c * G1_XS + 4 * i4] = q;
          }
        }
        // the quad's four positions t = tq .. tq + 3 (tq even): window ph = tq / 2 is centred on tq, ph + 1 on tq + 2;
        //   dz[tq]     = [c(ph) == centre] g(ph)
        //   dz[tq + 1] = [c(ph) == right] g(ph)     + [c(ph+1) == left] g(ph+1)
        //   dz[tq + 2] = [c(ph+1) == centre] g(ph+1)
        //   dz[tq + 3] = [c(ph+1) == right] g(ph+1) + [c(ph+2) == left] g(ph+2)        (no window ph + 2 at the end of the sequence)
        const int tq = t0 + 4 * sq;
        const bool inside = tq < L1, has2 = (tq >> 1) + 2 < P1;
        const float g0[4] = {gq[0].x, gq[0].y, gq[0].z, gq[0].w}, g1[4] = {gq[1].x, gq[1].y, gq[1].z, gq[1].w}, g2[4] = {gq[2].x, gq[2].y, gq[2].z, gq[2].w};
        float d[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned c0 = (cq[0] >> (2 * e)) & 3u, c1 = (cq[1] >> (2 * e)) & 3u, c2 = has2 ? (cq[2] >> (2 * e)) & 3u : 3u;
          d[0][e] = c0 == 1u ? g0[e] : 0.f;
          d[1][e] = (c0 == 2u ? g0[e] : 0.f) + (c1 == 0u ? g1[e] : 0.f);
          d[2][e] = c1 == 1u ? g1[e] : 0.f;
          d[3][e] = (c1 == 2u ? g1[e] : 0.f) + (c2 == 0u ? g2[e] : 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 dq = make_float4(d[j][0], d[j][1], d[j][2], d[j][3]), h;
          h.x = (yq[j].x - bn_mean[0]) * bn_inv[0];
          h.y = (yq[j].y - bn_mean[1]) * bn_inv[1];
          h.z = (yq[j].z - bn_mean[2]) * bn_inv[2];
          h.w = (yq[j].w - bn_mean[3]) * bn_inv[3];
          if (!inside) { dq = make_float4(0.f, 0.f, 0.f, 0.f); h = dq; }
          *(float4*)&dzs[sq * G1_QS + j * 16 + sc4 * 4] = dq;
          *(float4*)&xhs[sq * G1_QS + j * 16 + sc4 * 4] = h;
        }
      } else {
        stage_x_chunk<false>(xs, xb, C, T, t0, tid, G1_XS);
        for (int i = tid; i < G1_TCH * 4; i += 256) {
          const int row = i >> 2, c4 = i & 3, t = t0 + row;
          const int tc = t < L1 ? t : L1 - 1;
          const RoutedRaw r1 = routed_load<16>(dp1, code1, b, tc, P1, c4);
          const float4 yv = *(const float4*)(y1 + ((size_t)b * L1 + tc) * 16 + c4 * 4);
          float4 dq = routed_dz(r1, tc, P1), h;
          h.x = (yv.x - bn_mean[0]) * bn_inv[0];
          h.y = (yv.y - bn_mean[1]) * bn_inv[1];
          h.z = (yv.z - bn_mean[2]) * bn_inv[2];
          h.w = (yv.w - bn_mean[3]) * bn_inv[3];
          if (t >= L1) { dq = make_float4(0.f, 0.f, 0.f, 0.f); h = dq; }
          *(float4*)&dzs[g1_row(row) + c4 * 4] = dq;
          *(float4*)&xhs[g1_row(row) + c4 * 4] = h;
        }
      }
      __syncthreads();
      if (pipe) {
        if (ch + 1 < ch_hi) prefetch(b, ch + 1);
        else if (item + (int)gridDim.x < nitems) { const int ni = item + gridDim.x; prefetch(ni / SEG, (ni % SEG) * CPS); }
      }
#pragma unroll
      for (int m = 0; m < G1_TCH / 16; ++m) {       // each wave: 64 t's = 16 k-steps, position w*64 + 4m + lq
        const float a1 = dzs[aoff + G1_QS * m], a2 = xhs[aoff + G1_QS * m];
        s1 += a1; s2 = fmaf(a1, a2, s2);
#pragma unroll
        for (int nb = 0; nb < NBC; ++nb)
          if (CT > 0 || nb < NB) {
            const float bv = xs[coff[nb] + 8 * m];
            acc1[nb] = mfma16(a1, bv, acc1[nb]);
            acc2[nb] = mfma16(a2, bv, acc2[nb]);
          }
      }
    }
    // window done: this wave's shares of Gdz / Gxh -> LDS, summed over the waves in a fixed order -> the window's record
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NBC; ++nb)
      if (CT > 0 || nb < NB) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[((w * 2 + 0) * 16 + lq * 4 + e) * NB16 + nb * 16 + li] = acc1[nb][e];
          red[((w * 2 + 1) * 16 + lq * 4 + e) * NB16 + nb * 16 + li] = acc2[nb][e];
        }
      }
    __syncthreads();
    {
      float* gw = g1w + (size_t)item * 32 * NB16;
      for (int i = tid; i < 32 * NB16; i += 256)      // [which][o][col] <- sum over waves
        gw[i] = (red[i] + red[32 * NB16 + i]) + (red[64 * NB16 + i] + red[96 * NB16 + i]);
    }
  }
  // per-channel sums of this workgroup: lanes (lq, wave) hold shares of channel li
  __syncthreads();
  red[(w * 4 + lq) * 32 + li] = s1;
  red[(w * 4 + lq) * 32 + 16 + li] = s2;
  __syncthreads();
  if (tid < 32) {
    float a = 0.f;
    for (int q = 0; q < 16; ++q) a += red[q * 32 + tid];
    bpart[(size_t)blockIdx.x * 32 + tid] = a;          // column layout of the partial: [sum dz (16)][sum dz*xhat (16)]
  }
}

// Second half of conv1's backward (see conv1_bwd_kernel): with c1, c2 known, per window
//   G = scale * (Gdz - c1 * Sx - c2 * Gxh);   dW1 += s[b,c] * G (one partial row per workgroup);   ds[b,c] = sum_{o,kk} w1[o][c][kk] * G.
// Sx[c][kk] = sum_{0 <= t < L1} x[c][2t + kk - 3] (zero padding) = the sum of all samples of the parity of kk - 3 (eo, from the gate
// kernel) less the few at the window's edges that no t reaches.
__global__ __launch_bounds__(256) void conv1_bwd_fin_kernel(const float* __restrict__ g1w, const float* __restrict__ stat,
                                                            const float* __restrict__ cstat, const float* __restrict__ w1,
                                                            const float* __restrict__ gate_s, const float* __restrict__ eo,
                                                            const float* __restrict__ x, float* __restrict__ part,
                                                            float* __restrict__ ds_out, int B, int C, int T, int L1, int SEG, const FoldCtx fc) {
  FOLD_BEGIN; FS(g1w); FS(stat); FS(cstat); FS(w1); FS(gate_s); FS(eo); FS(x); FS(part); FS(ds_out);
  __shared__ float prod[16 * G1_MAXNB * 16];
  __shared__ float pc[16 * MSIG_MAX_C];
  __shared__ float ss[MSIG_MAX_C];
  __shared__ float sxs[G1_MAXNB * 16];
  const int K = C * 7, NB = (K + 15) / 16, NB16 = NB * 16, tid = threadIdx.x;
  float dwacc[G1_MAXNB], wv[G1_MAXNB], sc[G1_MAXNB], k1[G1_MAXNB], k2[G1_MAXNB];
  int col_[G1_MAXNB];
#pragma unroll
  for (int j = 0; j < G1_MAXNB; ++j) {
    const int idx = tid + 256 * j, o = (idx / NB16) & 15, col = idx % NB16;
    const bool ok = j < NB && col < K;
    dwacc[j] = 0.f;
    col_[j] = ok ? col : -1;
    wv[j] = ok ? w1[o * K + col] : 0.f;
    sc[j] = stat[32 + o]; k1[j] = cstat[o]; k2[j] = cstat[16 + o];
  }
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float* gw = g1w + (size_t)b * SEG * 32 * NB16;
    if (tid < C) ss[tid] = gate_s[(size_t)b * C + tid];
    if (tid >= 64 && tid < 64 + K) {
      const int col = tid - 64, c = col / 7, d = col - 7 * c - 3, par = d & 1;
      const float* xc = x + ((size_t)b * C + c) * T;
      float a = eo[((size_t)b * C + c) * 2 + par];
      for (int i = par; i < d; i += 2) a -= xc[i];                                   // samples before the first position's tap
      for (int i = 2 * (L1 - 1) + d + 2; i < T; i += 2) if (i >= 0) a -= xc[i];      // samples after the last position's tap
      sxs[col] = a;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < G1_MAXNB; ++j)
      if (j < NB) {
        const int idx = tid + 256 * j;
        float g = 0.f;
        if (col_[j] >= 0) {
          // the segments' partial sums, eight loads of each in flight, added in segment order (a plain loop is SEG dependent round trips)
          float gdz = gw[idx], gxh = gw[16 * NB16 + idx];
          for (int s0 = 1; s0 < SEG; s0 += 8) {
            float a8[8], b8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int sg = s0 + u < SEG ? s0 + u : SEG - 1;
              a8[u] = gw[(size_t)sg * 32 * NB16 + idx]; b8[u] = gw[(size_t)sg * 32 * NB16 + 16 * NB16 + idx];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (s0 + u < SEG) { gdz += a8[u]; gxh += b8[u]; }
          }
          g = sc[j] * (gdz - k1[j] * sxs[col_[j]] - gxh * k2[j]);
          dwacc[j] += ss[col_[j] / 7] * g;
        }
        prod[idx] = wv[j] * g;
      }
    __syncthreads();
    if (tid < 16 * C) {
      const int o = tid / C, c = tid - o * C;
      float a = 0.f;
      for (int kk = 0; kk < 7; ++kk) a += prod[o * NB16 + c * 7 + kk];
      pc[o * C + c] = a;
    }
    __syncthreads();
    if (tid < C) {
      float a = 0.f;
      for (int o = 0; o < 16; ++o) a += pc[o * C + tid];
      ds_out[(size_t)b * C + tid] = a;
    }
  }
#pragma unroll
  for (int j = 0; j < G1_MAXNB; ++j)
    if (j < NB && col_[j] >= 0) {
      const int idx = tid + 256 * j, o = idx / NB16;
      part[(size_t)blockIdx.x * 16 * K + o * K + col_[j]] = dwacc[j];
    }
}

// ------------------------------------------------------------------------------------
// Gate MLP backward (tiny): one workgroup per output weight, block-reduced over the batch
//   dz2[b,c] = ds[b,c] s(1-s);  dW2[c][j] = sum_b dz2[b,c] relu(a1[b,j])
//   da1[b,j] = (a1>0) sum_c W2[c][j] dz2[b,c];  dW1[j][c] = sum_b da1[b,j] mean[b,c]
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                       const float* __restrict__ pre, const float* __restrict__ mean,
                                                       const float* __restrict__ W2, float* __restrict__ dW1,
                                                       float* __restrict__ dW2, int B, int C, int Cr, const FoldCtx fc) {
  FOLD_BEGIN; FS(ds); FS(s); FS(pre); FS(mean); FS(W2); FS(dW1); FS(dW2);
  __shared__ double red[4];
  const int v = blockIdx.x, tid = threadIdx.x;
  const int which = v / (C * Cr), rem = v % (C * Cr);
  double acc = 0.0;
  for (int b = tid; b < B; b += 256) {
    if (which == 0) {                // dW2[c][j], rem = c*Cr + j
      const int c = rem / Cr, j = rem % Cr;
      const float sv = s[(size_t)b * C + c], a = pre[(size_t)b * Cr + j];
      acc += (double)(ds[(size_t)b * C + c] * sv * (1.f - sv) * (a > 0.f ? a : 0.f));
    } else {                         // dW1[j][c], rem = j*C + c
      const int j = rem / C, c = rem % C;
      if (pre[(size_t)b * Cr + j] > 0.f) {
        float da = 0.f;
        for (int cc = 0; cc < C; ++cc) {
          const float sv = s[(size_t)b * C + cc];
          da += W2[cc * Cr + j] * ds[(size_t)b * C + cc] * sv * (1.f - sv);
        }
        acc += (double)(da * mean[(size_t)b * C + c]);
      }
    }
  }
  acc = wave_sum_d(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) {
    const float r = (float)(red[0] + red[1] + red[2] + red[3]);
    if (which == 0) dW2[rem] = r; else dW1[rem] = r;
  }
}

// ------------------------------------------------------------------------------------
// Host launchers
// ------------------------------------------------------------------------------------
static inline int clampi(int64_t v, int hi) { return (int)(v < hi ? (v < 1 ? 1 : v) : hi); }

int launch_frontend_fwd(const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po, const FoldCtx& fc, hipStream_t st) {
  const float* P = b->params;
  float* mean = w.p<float>(MSIG_WS_GATE_MEAN);
  float* pre = w.p<float>(MSIG_WS_GATE_PRE);
  float* gs = w.p<float>(MSIG_WS_GATE_S);
  {
    MSIG_K("gate", st);
#define GATE(CT) gate_kernel<CT><<<dim3(d.B, 1, fc.n), 256, 0, st>>>(b->x, P + po[MSIG_P_GATE_W1], P + po[MSIG_P_GATE_W2], mean, pre, gs, \
                                                           b->training ? w.p<float>(MSIG_WS_GATE_EO) : nullptr, d.C, d.T, d.Cr, fc)
    switch (d.C) {
      case 1: GATE(1); break; case 2: GATE(2); break; case 3: GATE(3); break; case 4: GATE(4); break;
      case 5: GATE(5); break; case 6: GATE(6); break; case 7: GATE(7); break; case 8: GATE(8); break;
      default: GATE(0); break;
    }
#undef GATE
  }
  MSIG_LAUNCH_CHECK();
  const int tr = b->training;
  // ---- stage 1
  {
    const int nchunk = (d.L1 + C1_CHUNK - 1) / C1_CHUNK;
    const int grid = clampi((int64_t)d.B * nchunk, MSIG_PERSIST_WG);
    const int KM = (d.C * 7 + 3) / 4;
    size_t smem = (size_t)(2 * d.C * C1_XP + 4 * KM * 16) * sizeof(float);
    if (smem < 4 * 32 * sizeof(float)) smem = 4 * 32 * sizeof(float);
    {
      MSIG_K("conv1_fwd", st);
#define C1F(CT) conv1_fwd_kernel<CT><<<dim3(grid, 1, fc.n), 256, smem, st>>>(b->x, P + po[MSIG_P_CONV1_W], gs, w.p<float>(MSIG_WS_Y1), \
                                                             w.p<float>(MSIG_WS_BN1_PART), d.B, d.C, d.T, d.L1, tr, fc)
      switch (d.C) {
        case 1: C1F(1); break; case 2: C1F(2); break; case 3: C1F(3); break; case 4: C1F(4); break;
        case 5: C1F(5); break; case 6: C1F(6); break; case 7: C1F(7); break; case 8: C1F(8); break;
        default: C1F(0); break;
      }
#undef C1F
    }
    MSIG_LAUNCH_CHECK();
    { MSIG_K("bn_finalize", st); bn_finalize_kernel<<<dim3(1, 1, fc.n), FIN_THREADS, 0, st>>>(w.p<float>(MSIG_WS_BN1_PART), grid, 16, (double)d.B * d.L1, P + po[MSIG_P_BN1_G],
                                          P + po[MSIG_P_BN1_B], b->bn_state, b->bn_state + 16, b->bn_count, b->bn_momentum,
                                          b->bn_eps, tr, w.p<float>(MSIG_WS_BN1_STAT), fc); }
    MSIG_LAUNCH_CHECK();
  }
  // ---- stage 1 pooling + stage 2 convolution (one kernel: p1 is written for the backward pass, never read back here)
  {
    const int nchunk = (d.L2 + C2_CHUNK - 1) / C2_CHUNK;
    const int grid = clampi((int64_t)d.B * nchunk, MSIG_PERSIST_WG);
    { MSIG_K("pool1_conv2_fwd", st); pool1_conv2_fwd_kernel<<<dim3(grid, 1, fc.n), 256, 0, st>>>(w.p<float>(MSIG_WS_Y1), w.p<float>(MSIG_WS_BN1_STAT), w.p<float>(MSIG_WS_P1),
                                            tr ? w.p<uint8_t>(MSIG_WS_POOLC1) : nullptr, P + po[MSIG_P_CONV2_W], w.p<float>(MSIG_WS_Y2),
                                            w.p<float>(MSIG_WS_BN2_PART), d.B, d.L1, d.P1, d.L2, tr, fc); }
    MSIG_LAUNCH_CHECK();
    { MSIG_K("bn_finalize", st); bn_finalize_kernel<<<dim3(1, 1, fc.n), FIN_THREADS, 0, st>>>(w.p<float>(MSIG_WS_BN2_PART), grid, 32, (double)d.B * d.L2, P + po[MSIG_P_BN2_G],
                                          P + po[MSIG_P_BN2_B], b->bn_state + 32, b->bn_state + 64, b->bn_count + 1,
                                          b->bn_momentum, b->bn_eps, tr, w.p<float>(MSIG_WS_BN2_STAT), fc); }
    MSIG_LAUNCH_CHECK();
    const int64_t n = (int64_t)d.B * d.TP * 8;
    { MSIG_K("bn_relu_pool_32", st); bn_relu_pool_kernel<32><<<dim3(clampi((n + 255) / 256, 8192), 1, fc.n), 256, 0, st>>>(w.p<float>(MSIG_WS_Y2), w.p<float>(MSIG_WS_BN2_STAT),
                                                                          w.p<float>(MSIG_WS_P2), tr ? w.p<uint8_t>(MSIG_WS_POOLC2) : nullptr, d.B, d.L2, d.TP, fc); }
    MSIG_LAUNCH_CHECK();
  }
  return 0;
}

int launch_frontend_bwd(const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po, ColsumPlan& plan, const FoldCtx& fc, hipStream_t st) {
  const float* P = b->params;
  float* G = b->grads;
  const PartOffsets pof = part_offsets(d);
  float* part2 = w.p<float>(MSIG_WS_GRAD_PART) + pof.conv2;
  float* part1 = w.p<float>(MSIG_WS_GRAD_PART) + pof.conv1;
  float* bpart = w.p<float>(MSIG_WS_BNB_PART);
  float* cstat = w.p<float>(MSIG_WS_BNB_STAT);
  // ---- stage 2: pool2/relu/bn2 backward.  dP2 = DX0[fwd] + DX0[rev]
  {
    const float* dxa = w.p<float>(MSIG_WS_DX0);
    const float* dxb = dxa + (size_t)d.B * d.TP * 32;
    const int grid = clampi(((int64_t)d.B * d.L2 * 8 + 255) / 256, MSIG_PERSIST_WG);
    { MSIG_K("pool_bn_bwd_pass1_32", st); pool_bn_bwd_pass1<32><<<dim3(grid, 1, fc.n), 256, 0, st>>>(dxa, dxb, w.p<uint8_t>(MSIG_WS_POOLC2), w.p<float>(MSIG_WS_Y2),
                                                w.p<float>(MSIG_WS_BN2_STAT), w.p<float>(MSIG_WS_DY2), bpart, d.B, d.L2, d.TP, fc); }
    MSIG_LAUNCH_CHECK();
    { MSIG_K("bn_bwd_finalize", st); bn_bwd_finalize_kernel<<<dim3(1, 1, fc.n), FIN_THREADS, 0, st>>>(bpart, grid, 32, (double)d.B * d.L2, cstat, G + po[MSIG_P_BN2_G], G + po[MSIG_P_BN2_B], fc); }
    MSIG_LAUNCH_CHECK();
    // (pass 2 of this stage is fused into the staging of conv2_bwd: WS_DY2 keeps dL/d(bn2 output))
  }
  // ---- conv2 backward
  {
    const int gdw = clampi((int64_t)d.B * ((d.L2 + D2_UCH - 1) / D2_UCH), MSIG_CONV_DW_WG);
    { MSIG_K("conv2_bwd", st); conv2_bwd_kernel<<<dim3(gdw, 1, fc.n), 256, 0, st>>>(w.p<float>(MSIG_WS_DY2), w.p<float>(MSIG_WS_Y2), w.p<float>(MSIG_WS_BN2_STAT), cstat,
                                                                          P + po[MSIG_P_CONV2_W], w.p<float>(MSIG_WS_P1), w.p<float>(MSIG_WS_DP1), part2,
                                                                          d.B, d.P1, d.L2, fc); }
    MSIG_LAUNCH_CHECK();
    if (!plan.add(part2, gdw, 2560, 0, 2560, G + po[MSIG_P_CONV2_W])) return MSIG_E_SHAPE;
  }
  // ---- stage 1 (pool1 / relu / bn1 backward) + conv1 backward: one pass over dP1 / y1 / x (conv1_bwd), the BatchNorm sums
  //      finalised from its partials, then the per-window combination (conv1_bwd_fin)
  {
    const int K = d.C * 7, NB = (K + 15) / 16;
    const int nchunk1 = (d.L1 + G1_TCH - 1) / G1_TCH, seg = conv1_bwd_segs(d.B, nchunk1);
    const int grid = clampi((int64_t)d.B * seg, MSIG_CONV_DW_WG), grid_fin = clampi(d.B, MSIG_CONV_DW_WG);
    size_t smem = (size_t)(d.C * G1_XS + 2 * (G1_TCH / 4) * G1_QS) * sizeof(float);
    if (smem < (size_t)128 * NB * 16 * sizeof(float)) smem = (size_t)128 * NB * 16 * sizeof(float);     // the window-end reduction buffer aliases the staging area
    float* g1w = w.p<float>(MSIG_WS_G1W);
    {
      MSIG_K("conv1_bwd", st);
#define C1B(CT) conv1_bwd_kernel<CT><<<dim3(grid, 1, fc.n), 256, smem, st>>>(w.p<float>(MSIG_WS_DP1), w.p<uint8_t>(MSIG_WS_POOLC1), w.p<float>(MSIG_WS_Y1), w.p<float>(MSIG_WS_BN1_STAT), b->x, \
                                                             g1w, bpart, d.B, d.C, d.T, d.L1, d.P1, fc)
      switch (d.C) {
        case 1: C1B(1); break; case 2: C1B(2); break; case 3: C1B(3); break; case 4: C1B(4); break;
        case 5: C1B(5); break; case 6: C1B(6); break; case 7: C1B(7); break; case 8: C1B(8); break;
        default: C1B(0); break;
      }
#undef C1B
    }
    MSIG_LAUNCH_CHECK();
    { MSIG_K("bn_bwd_finalize", st); bn_bwd_finalize_kernel<<<dim3(1, 1, fc.n), FIN_THREADS, 0, st>>>(bpart, grid, 16, (double)d.B * d.L1, cstat, G + po[MSIG_P_BN1_G], G + po[MSIG_P_BN1_B], fc); }
    MSIG_LAUNCH_CHECK();
    { MSIG_K("conv1_bwd_fin", st); conv1_bwd_fin_kernel<<<dim3(grid_fin, 1, fc.n), 256, 0, st>>>(g1w, w.p<float>(MSIG_WS_BN1_STAT), cstat, P + po[MSIG_P_CONV1_W], w.p<float>(MSIG_WS_GATE_S),
                                                                                         w.p<float>(MSIG_WS_GATE_EO), b->x, part1, w.p<float>(MSIG_WS_DS), d.B, d.C, d.T, d.L1, seg, fc); }
    MSIG_LAUNCH_CHECK();
    if (!plan.add(part1, grid_fin, 16 * K, 0, 16 * K, G + po[MSIG_P_CONV1_W])) return MSIG_E_SHAPE;
    if (d.Cr > 0) {
      { MSIG_K("gate_bwd", st); gate_bwd_kernel<<<dim3(2 * d.C * d.Cr, 1, fc.n), 256, 0, st>>>(w.p<float>(MSIG_WS_DS), w.p<float>(MSIG_WS_GATE_S), w.p<float>(MSIG_WS_GATE_PRE),
                                                        w.p<float>(MSIG_WS_GATE_MEAN), P + po[MSIG_P_GATE_W2], G + po[MSIG_P_GATE_W1],
                                                        G + po[MSIG_P_GATE_W2], d.B, d.C, d.Cr, fc); }
      MSIG_LAUNCH_CHECK();
    }
  }
  return 0;
}
