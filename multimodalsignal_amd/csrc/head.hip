// Classifier head (models.py:66-71,80), CrossEntropyLoss (trainer.py:69,147), the
// eval-time softmax/argmax (trainer.py:224-225), Adam (trainer.py:68,149) and the small
// utility kernels (partial-sum reduction, window gather).  These stages are <0.1 % of the
// FLOPs, so they are plain VALU kernels; what matters is that they stay on the stream and
// never force a host sync.
#include "msig_dev.h"

#define HEAD_ROWS 16
#define W0T_S 65     // padded row stride of the transposed Linear(128,64) weight in LDS
#define HEAD_WG 128

// ------------------------------------------------------------------------------------
// feat (B,128) -> hid = dropout(relu(W0 feat + b0)) (B,64) -> logits = W3 hid + b3 (B,K)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ W0,
                                                       const float* __restrict__ b0, const float* __restrict__ W3,
                                                       const float* __restrict__ b3, float* __restrict__ hid,
                                                       float* __restrict__ logits, int B, int K, int drop_thr,
                                                       uint32_t drop_key, float dscale, const FoldCtx fc) {
  FOLD_BEGIN; FS(feat); FS(W0); FS(b0); FS(W3); FS(b3); FS(hid); FS(logits); drop_key = fc.key_head[blockIdx.z];
  __shared__ float W0t[128 * W0T_S];
  __shared__ float W3s[MSIG_MAX_K * 64];
  __shared__ float fs[HEAD_ROWS * 128];
  __shared__ float hs[HEAD_ROWS * 64];
  const int tid = threadIdx.x;
  for (int j0 = 0; j0 < 32; j0 += 8) {            // 32 KB of W0, transposed into LDS: eight loads in flight per thread (a plain loop is 32 round trips)
    float q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = W0[tid + 256 * (j0 + u)];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = tid + 256 * (j0 + u), v = i >> 7, k = i & 127; W0t[k * W0T_S + v] = q[u]; }
  }
  for (int i = tid; i < K * 64; i += 256) W3s[i] = W3[i];
  const int ngroups = (B + HEAD_ROWS - 1) / HEAD_ROWS;
  const int v = tid & 63, rg = tid >> 6;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int r0 = grp * HEAD_ROWS;
    __syncthreads();
    for (int i = tid; i < HEAD_ROWS * 128; i += 256) {
      const int row = r0 + (i >> 7);
      fs[i] = row < B ? feat[(size_t)row * 128 + (i & 127)] : 0.f;
    }
    __syncthreads();
    float acc[4] = {b0[v], b0[v], b0[v], b0[v]};
    for (int k = 0; k < 128; ++k) {
      const float wv = W0t[k * W0T_S + v];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += wv * fs[(rg * 4 + r) * 128 + k];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + rg * 4 + r;
      float hv = acc[r] > 0.f ? acc[r] : 0.f;
      if (drop_thr > 0) {
        const uint32_t e = (uint32_t)row * 64u + (uint32_t)v;
        hv *= drop_mul(drop_word(e, drop_key), e & 3, drop_thr, dscale);
      }
      hs[(rg * 4 + r) * 64 + v] = hv;
      if (row < B) hid[(size_t)row * 64 + v] = hv;
    }
    __syncthreads();
    for (int i = tid; i < HEAD_ROWS * K; i += 256) {
      const int rl = i / K, c = i - rl * K, row = r0 + rl;
      float a = b3[c];
      for (int vv = 0; vv < 64; ++vv) a += W3s[c * 64 + vv] * hs[rl * 64 + vv];
      if (row < B) logits[(size_t)row * K + c] = a;
    }
  }
}

// One row of CrossEntropy: argmax and log-sum-exp of its K logits — shared by ce_kernel, head_step_kernel and the loss finalizer of
// the fused step (colsum_adam_kernel), which must agree in every bit.
__device__ __forceinline__ float ce_row_lse(const float* lg, int K, int& am) {
  float mx = lg[0]; am = 0;
  for (int c = 1; c < K; ++c) if (lg[c] > mx) { mx = lg[c]; am = c; }
  float se = 0.f;
  for (int c = 0; c < K; ++c) se += expf(lg[c] - mx);
  return mx + logf(se);
}

// ------------------------------------------------------------------------------------
// CrossEntropy (mean), dlogits, softmax probabilities, argmax, accuracy counter.
// Single workgroup (one deterministic sum) of CE_THREADS threads — with 256 the 32 rows per thread of a B = 8192 batch took 62 us,
// all of it exp / log latency.  lossbuf[0] = mean loss of this batch;
// lossbuf[1] = summed loss of this batch (loss.item() * B, trainer.py:152,221); lossbuf[2] = #correct of this batch — plain
// stores, no running sums: layouts of different batch sizes alias one pooled workspace, and the epoch sums live with the caller.
// ------------------------------------------------------------------------------------
#define CE_THREADS 1024
__global__ __launch_bounds__(CE_THREADS) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                 float* __restrict__ probs, int* __restrict__ pred,
                                                 float* __restrict__ dlogits, float* __restrict__ lossbuf, double* __restrict__ lacc, int B, int K, const FoldCtx fc) {
  FOLD_BEGIN; FS(logits); FS(labels); FS(probs); FS(pred); FS(dlogits); FS(lossbuf); FS(lacc);
  __shared__ double red[2][CE_THREADS / 64];
  const int tid = threadIdx.x;
  double lsum = 0.0, correct = 0.0;
  const float invB = 1.0f / (float)B;
  for (int row = tid; row < B; row += CE_THREADS) {
    const float* lg = logits + (size_t)row * K;
    int am;
    const float lse = ce_row_lse(lg, K, am);
    const int y = (int)labels[row];
    lsum += (double)(lse - lg[y]);
    correct += (am == y) ? 1.0 : 0.0;
    pred[row] = am;
    for (int c = 0; c < K; ++c) {
      const float p = expf(lg[c] - lse);
      probs[(size_t)row * K + c] = p;
      if (dlogits) dlogits[(size_t)row * K + c] = (p - (c == y ? 1.f : 0.f)) * invB;
    }
  }
  lsum = wave_sum_d(lsum); correct = wave_sum_d(correct);
  if ((tid & 63) == 0) { red[0][tid >> 6] = lsum; red[1][tid >> 6] = correct; }
  __syncthreads();
  if (tid == 0) {
    double ls = 0.0, cs = 0.0;
    for (int i = 0; i < CE_THREADS / 64; ++i) { ls += red[0][i]; cs += red[1][i]; }
    lossbuf[0] = (float)(ls / (double)B);
    lossbuf[1] = (float)ls;
    lossbuf[2] = (float)cs;
    if (lacc) { lacc[0] += ls; lacc[1] += cs; }      // msig_batch.loss_acc: the caller's running sums of a pass (one thread, stream order)
  }
}

// The reduction half of ce_kernel for the fused head (head_step_kernel wrote pred / probs / dlogits; the loss and the accuracy
// counter are sums over the whole batch): one workgroup of 256 threads plays ce_kernel's 1024 — virtual thread q * 256 + tid sums
// the rows ce_kernel's thread of that index sums, each real wave reduces four virtual waves, thread 0 adds the sixteen wave sums in
// ce_kernel's order.  Same values, same order: the loss is ce_kernel's, bit for bit.
__device__ __forceinline__ void loss_finalize(const LossFin& lf, double (*red)[CE_THREADS / 64]) {
  const int tid = threadIdx.x;
#pragma unroll 1
  for (int q = 0; q < CE_THREADS / 256; ++q) {
    double lsum = 0.0, correct = 0.0;
    for (int row = q * 256 + tid; row < lf.B; row += CE_THREADS) {
      const float* lg = lf.logits + (size_t)row * lf.K;
      int am;
      const float lse = ce_row_lse(lg, lf.K, am);
      const int y = (int)lf.labels[row];
      lsum += (double)(lse - lg[y]);
      correct += (am == y) ? 1.0 : 0.0;
    }
    lsum = wave_sum_d(lsum); correct = wave_sum_d(correct);
    if ((tid & 63) == 0) { red[0][q * 4 + (tid >> 6)] = lsum; red[1][q * 4 + (tid >> 6)] = correct; }
  }
  __syncthreads();
  if (tid == 0) {
    double ls = 0.0, cs = 0.0;
    for (int i = 0; i < CE_THREADS / 64; ++i) { ls += red[0][i]; cs += red[1][i]; }
    lf.lossbuf[0] = (float)(ls / (double)lf.B);
    lf.lossbuf[1] = (float)ls;
    lf.lossbuf[2] = (float)cs;
    if (lf.lacc) { lf.lacc[0] += ls; lf.lacc[1] += cs; }
  }
}

// softmax + argmax only (no labels)
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ logits, float* __restrict__ probs,
                                                      int* __restrict__ pred, int B, int K, const FoldCtx fc) {
  FOLD_BEGIN; FS(logits); FS(probs); FS(pred);
  for (int row = blockIdx.x * 256 + threadIdx.x; row < B; row += gridDim.x * 256) {
    const float* lg = logits + (size_t)row * K;
    float mx = lg[0]; int am = 0;
    for (int c = 1; c < K; ++c) if (lg[c] > mx) { mx = lg[c]; am = c; }
    float se = 0.f;
    for (int c = 0; c < K; ++c) se += expf(lg[c] - mx);
    for (int c = 0; c < K; ++c) probs[(size_t)row * K + c] = expf(lg[c] - mx) / se;
    pred[row] = am;
  }
}

// ------------------------------------------------------------------------------------
// head backward: dlogits -> dW3, db3, dW0, db0 (per-workgroup partials) and dfeat
// partial row layout: [dW0 64*128][db0 64][dW3 K*64][db3 K]
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ feat,
                                                       const float* __restrict__ hid, const float* __restrict__ W0,
                                                       const float* __restrict__ W3, float* __restrict__ dfeat,
                                                       float* __restrict__ part, int B, int K, float dscale, const FoldCtx fc) {
  FOLD_BEGIN; FS(dlogits); FS(feat); FS(hid); FS(W0); FS(W3); FS(dfeat); FS(part);
  __shared__ float W0t[128 * W0T_S];
  __shared__ float W3s[MSIG_MAX_K * 64];
  __shared__ float fs[HEAD_ROWS * 128];
  __shared__ float hs[HEAD_ROWS * 64];
  __shared__ float dps[HEAD_ROWS * 64];
  __shared__ float dls[HEAD_ROWS * MSIG_MAX_K];
  const int tid = threadIdx.x;
  for (int j0 = 0; j0 < 32; j0 += 8) {            // 32 KB of W0, transposed into LDS: eight loads in flight per thread (a plain loop is 32 round trips)
    float q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = W0[tid + 256 * (j0 + u)];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = tid + 256 * (j0 + u), v = i >> 7, k = i & 127; W0t[k * W0T_S + v] = q[u]; }
  }
  for (int i = tid; i < K * 64; i += 256) W3s[i] = W3[i];
  float dW0acc[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) dW0acc[j] = 0.f;
  float dW3acc[4] = {0.f, 0.f, 0.f, 0.f};
  float db0acc = 0.f, db3acc = 0.f;
  const int ngroups = (B + HEAD_ROWS - 1) / HEAD_ROWS;
  const int v = tid & 63, rg = tid >> 6, kcol = tid & 127, half = tid >> 7;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int r0 = grp * HEAD_ROWS;
    __syncthreads();
    for (int i = tid; i < HEAD_ROWS * 128; i += 256) { const int row = r0 + (i >> 7); fs[i] = row < B ? feat[(size_t)row * 128 + (i & 127)] : 0.f; }
    for (int i = tid; i < HEAD_ROWS * 64; i += 256) { const int row = r0 + (i >> 6); hs[i] = row < B ? hid[(size_t)row * 64 + (i & 63)] : 0.f; }
    for (int i = tid; i < HEAD_ROWS * K; i += 256) { const int row = r0 + i / K; dls[(i / K) * MSIG_MAX_K + (i % K)] = row < B ? dlogits[(size_t)row * K + (i % K)] : 0.f; }
    __syncthreads();
    // d(pre-activation of Linear(128,64))
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rl = rg * 4 + r;
      float a = 0.f;
      for (int c = 0; c < K; ++c) a += W3s[c * 64 + v] * dls[rl * MSIG_MAX_K + c];
      dps[rl * 64 + v] = hs[rl * 64 + v] > 0.f ? a * dscale : 0.f;
    }
    __syncthreads();
    // weight-gradient accumulation
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + 256 * j;
      if (idx < K * 64) {
        const int c = idx >> 6, vv = idx & 63;
        float a = 0.f;
#pragma unroll 4
        for (int rl = 0; rl < HEAD_ROWS; ++rl) a += dls[rl * MSIG_MAX_K + c] * hs[rl * 64 + vv];
        dW3acc[j] += a;
      }
    }
    if (tid < K) { float a = 0.f; for (int rl = 0; rl < HEAD_ROWS; ++rl) a += dls[rl * MSIG_MAX_K + tid]; db3acc += a; }
    if (tid < 64) { float a = 0.f; for (int rl = 0; rl < HEAD_ROWS; ++rl) a += dps[rl * 64 + tid]; db0acc += a; }
#pragma unroll 1
    for (int rl = 0; rl < HEAD_ROWS; ++rl) {
      const float fv = fs[rl * 128 + kcol];
#pragma unroll
      for (int j = 0; j < 32; ++j) dW0acc[j] += dps[rl * 64 + half * 32 + j] * fv;
    }
    // dfeat[row][k] = sum_v W0[v][k] dpre[row][v]; thread: column kcol, rows half*8..+8
#pragma unroll 1
    for (int r = 0; r < 8; ++r) {
      const int rl = half * 8 + r, row = r0 + rl;
      float a = 0.f;
#pragma unroll 8
      for (int vv = 0; vv < 64; ++vv) a += W0t[kcol * W0T_S + vv] * dps[rl * 64 + vv];
      if (row < B) dfeat[(size_t)row * 128 + kcol] = a;
    }
  }
  float* P = part + (size_t)blockIdx.x * (64 * 128 + 64 + K * 64 + K);
#pragma unroll
  for (int j = 0; j < 32; ++j) P[(half * 32 + j) * 128 + kcol] = dW0acc[j];
  if (tid < 64) P[64 * 128 + tid] = db0acc;
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int idx = tid + 256 * j; if (idx < K * 64) P[64 * 128 + 64 + idx] = dW3acc[j]; }
  if (tid < K) P[64 * 128 + 64 + K * 64 + tid] = db3acc;
}

// ------------------------------------------------------------------------------------
// The head of a fused train step at few windows, one launch instead of three (head_fwd, ce, head_bwd): a workgroup takes ONE
// group of 16 rows through the classifier, its rows' CrossEntropy terms (dlogits need nothing but their own row) and the
// backward pass, with the weights, the features, the hidden layer and dlogits staying in LDS.  What needs the whole batch — the
// loss and the accuracy counter — is summed by loss_finalize in the step's last launch.  Every statement is the corresponding one
// of head_fwd_kernel / ce_kernel / head_bwd_kernel: the step's bits are those of the three launches
// (tests/test_parity_gpu.py::test_fused_step_equals_separate_calls_bit_for_bit).  grid = (B + 15) / 16 <= HEAD_WG workgroups, each
// writing one partial row, as head_bwd_kernel does at that size.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_step_kernel(const float* __restrict__ feat, const float* __restrict__ W0, const float* __restrict__ b0,
                                                        const float* __restrict__ W3, const float* __restrict__ b3, const int64_t* __restrict__ labels,
                                                        float* __restrict__ hid, float* __restrict__ logits, float* __restrict__ probs, int* __restrict__ pred,
                                                        float* __restrict__ dlogits, float* __restrict__ dfeat, float* __restrict__ part,
                                                        int B, int K, int drop_thr, uint32_t drop_key, float dscale, float dscale_bwd, const FoldCtx fc) {
  FOLD_BEGIN; FS(feat); FS(W0); FS(b0); FS(W3); FS(b3); FS(labels); FS(hid); FS(logits); FS(probs); FS(pred); FS(dlogits); FS(dfeat); FS(part);
  drop_key = fc.key_head[blockIdx.z];
  __shared__ float W0t[128 * W0T_S];
  __shared__ float W3s[MSIG_MAX_K * 64];
  __shared__ float fs[HEAD_ROWS * 128];
  __shared__ float hs[HEAD_ROWS * 64];
  __shared__ float dps[HEAD_ROWS * 64];
  __shared__ float dls[HEAD_ROWS * MSIG_MAX_K];
  __shared__ float lgs[HEAD_ROWS * MSIG_MAX_K];
  const int tid = threadIdx.x;
  for (int j0 = 0; j0 < 32; j0 += 8) {
    float q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = W0[tid + 256 * (j0 + u)];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = tid + 256 * (j0 + u), v = i >> 7, k = i & 127; W0t[k * W0T_S + v] = q[u]; }
  }
  for (int i = tid; i < K * 64; i += 256) W3s[i] = W3[i];
  const int v = tid & 63, rg = tid >> 6, kcol = tid & 127, half = tid >> 7;
  const int r0 = blockIdx.x * HEAD_ROWS;
  for (int i = tid; i < HEAD_ROWS * 128; i += 256) {
    const int row = r0 + (i >> 7);
    fs[i] = row < B ? feat[(size_t)row * 128 + (i & 127)] : 0.f;
  }
  __syncthreads();
  // ---- head_fwd_kernel ----
  {
    float acc[4] = {b0[v], b0[v], b0[v], b0[v]};
    for (int k = 0; k < 128; ++k) {
      const float wv = W0t[k * W0T_S + v];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += wv * fs[(rg * 4 + r) * 128 + k];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + rg * 4 + r;
      float hv = acc[r] > 0.f ? acc[r] : 0.f;
      if (drop_thr > 0) {
        const uint32_t e = (uint32_t)row * 64u + (uint32_t)v;
        hv *= drop_mul(drop_word(e, drop_key), e & 3, drop_thr, dscale);
      }
      hs[(rg * 4 + r) * 64 + v] = hv;
      if (row < B) hid[(size_t)row * 64 + v] = hv;
    }
  }
  __syncthreads();
  for (int i = tid; i < HEAD_ROWS * K; i += 256) {
    const int rl = i / K, c = i - rl * K, row = r0 + rl;
    float a = b3[c];
    for (int vv = 0; vv < 64; ++vv) a += W3s[c * 64 + vv] * hs[rl * 64 + vv];
    lgs[rl * MSIG_MAX_K + c] = a;
    if (row < B) logits[(size_t)row * K + c] = a;
  }
  __syncthreads();
  // ---- ce_kernel, the rows of this group (its sums: loss_finalize) ----
  if (tid < HEAD_ROWS) {
    const int row = r0 + tid;
    if (row < B) {
      const float* lg = &lgs[tid * MSIG_MAX_K];
      const float invB = 1.0f / (float)B;
      int am;
      const float lse = ce_row_lse(lg, K, am);
      const int y = (int)labels[row];
      pred[row] = am;
      for (int c = 0; c < K; ++c) {
        const float p = expf(lg[c] - lse);
        probs[(size_t)row * K + c] = p;
        const float dl = (p - (c == y ? 1.f : 0.f)) * invB;
        dlogits[(size_t)row * K + c] = dl;
        dls[tid * MSIG_MAX_K + c] = dl;
      }
    } else {
      for (int c = 0; c < K; ++c) dls[tid * MSIG_MAX_K + c] = 0.f;
    }
  }
  // head_bwd_kernel reads hid as 0 for the rows past the batch
  for (int i = tid; i < HEAD_ROWS * 64; i += 256) if (r0 + (i >> 6) >= B) hs[i] = 0.f;
  __syncthreads();
  // ---- head_bwd_kernel ----
  float dW0acc[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) dW0acc[j] = 0.f;
  float dW3acc[4] = {0.f, 0.f, 0.f, 0.f};
  float db0acc = 0.f, db3acc = 0.f;
  {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rl = rg * 4 + r;
      float a = 0.f;
      for (int c = 0; c < K; ++c) a += W3s[c * 64 + v] * dls[rl * MSIG_MAX_K + c];
      dps[rl * 64 + v] = hs[rl * 64 + v] > 0.f ? a * dscale_bwd : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + 256 * j;
      if (idx < K * 64) {
        const int c = idx >> 6, vv = idx & 63;
        float a = 0.f;
#pragma unroll 4
        for (int rl = 0; rl < HEAD_ROWS; ++rl) a += dls[rl * MSIG_MAX_K + c] * hs[rl * 64 + vv];
        dW3acc[j] += a;
      }
    }
    if (tid < K) { float a = 0.f; for (int rl = 0; rl < HEAD_ROWS; ++rl) a += dls[rl * MSIG_MAX_K + tid]; db3acc += a; }
    if (tid < 64) { float a = 0.f; for (int rl = 0; rl < HEAD_ROWS; ++rl) a += dps[rl * 64 + tid]; db0acc += a; }
#pragma unroll 1
    for (int rl = 0; rl < HEAD_ROWS; ++rl) {
      const float fv = fs[rl * 128 + kcol];
#pragma unroll
      for (int j = 0; j < 32; ++j) dW0acc[j] += dps[rl * 64 + half * 32 + j] * fv;
    }
#pragma unroll 1
    for (int r = 0; r < 8; ++r) {
      const int rl = half * 8 + r, row = r0 + rl;
      float a = 0.f;
#pragma unroll 8
      for (int vv = 0; vv < 64; ++vv) a += W0t[kcol * W0T_S + vv] * dps[rl * 64 + vv];
      if (row < B) dfeat[(size_t)row * 128 + kcol] = a;
    }
  }
  float* P = part + (size_t)blockIdx.x * (64 * 128 + 64 + K * 64 + K);
#pragma unroll
  for (int j = 0; j < 32; ++j) P[(half * 32 + j) * 128 + kcol] = dW0acc[j];
  if (tid < 64) P[64 * 128 + tid] = db0acc;
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int idx = tid + 256 * j; if (idx < K * 64) P[64 * 128 + 64 + idx] = dW3acc[j]; }
  if (tid < K) P[64 * 128 + 64 + K * 64 + tid] = db3acc;
}

// ------------------------------------------------------------------------------------
// out[c] = sum_r part[r*row_stride + c], fp64 accumulation in a fixed order
// ------------------------------------------------------------------------------------
// 32 columns x 8 row-lanes per workgroup; a lane sums every 8th row with 8 loads in flight into 8
// accumulators that are combined in a fixed order (deterministic for a given nrows)
#define CS_COLS 32
#define CS_LANES 8
__device__ __forceinline__ double colsum_lane(const float* __restrict__ p, int nrows, size_t stride, int ry) {
  double a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.0;
  int r = ry;
  for (; r + 7 * CS_LANES < nrows; r += 8 * CS_LANES) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(r + j * CS_LANES) * stride];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += (double)v[j];
  }
  for (; r < nrows; r += CS_LANES) a[0] += (double)p[(size_t)r * stride];
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
__device__ __forceinline__ double colsum_fold(double (*red)[CS_COLS], int cx) {
  return ((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx])) + ((red[4][cx] + red[5][cx]) + (red[6][cx] + red[7][cx]));
}

// One launch reduces every job: blockIdx.x walks the 32-column blocks of ALL jobs back to back (blk0[j] = first block of job j),
// blockIdx.y = fold.  Round 4's grid was (blocks of the WIDEST job, jobs, folds): 768 x 32 x folds workgroups of which nine in ten
// found no columns and left — at 15 folds 370 000 workgroups, and the launch took what dispatching them takes (0.16 ms) whatever
// the bytes; now 4 000 per fold, every one with work.
struct ColsumJobs { ColsumJob j[MSIG_MAX_JOBS]; int blk0[MSIG_MAX_JOBS + 1]; int n; };
__device__ __forceinline__ int colsum_find_job(const ColsumJobs& jobs, int blk) {      // uniform: scalar loop over <= 40 entries
  int j = 0;
  while (j + 1 < jobs.n && blk >= jobs.blk0[j + 1]) ++j;
  return j;
}
static int colsum_fill(const ColsumPlan& plan, ColsumJobs& a) {
  int at = 0;
  for (int i = 0; i < plan.n; ++i) { a.j[i] = plan.job[i]; a.blk0[i] = at; at += (plan.job[i].ncols + CS_COLS - 1) / CS_COLS; }
  for (int i = plan.n; i < MSIG_MAX_JOBS; ++i) { a.j[i] = ColsumJob{nullptr, 0, 0, 0, 0, nullptr}; a.blk0[i] = at; }
  a.blk0[MSIG_MAX_JOBS] = at;
  a.n = plan.n;
  return at;
}

__global__ __launch_bounds__(256) void colsum_plan_kernel(const ColsumJobs jobs, const FoldCtx fc) {
  __shared__ double red[CS_LANES][CS_COLS];
  const int ji = colsum_find_job(jobs, blockIdx.x);
  ColsumJob jb = jobs.j[ji];
  const int64_t foff_ = (int64_t)fc.slot[blockIdx.y] * fc.stride;
  FS(jb.part); FS(jb.out);
  const int cx = threadIdx.x & (CS_COLS - 1), ry = threadIdx.x / CS_COLS;
  const int c = ((int)blockIdx.x - jobs.blk0[ji]) * CS_COLS + cx;
  red[ry][cx] = c < jb.ncols ? colsum_lane(jb.part + jb.col0 + c, jb.nrows, (size_t)jb.row_stride, ry) : 0.0;
  __syncthreads();
  if (ry == 0 && c < jb.ncols) jb.out[c] = (float)colsum_fold(red, cx);
}

__global__ __launch_bounds__(256) void colsum_adam_kernel(const ColsumJobs jobs, const AdamArgs ad_in, const LossFin loss_in, const FoldCtx fc) {
  __shared__ double red[CS_LANES][CS_COLS];
  if ((int)blockIdx.x == jobs.blk0[MSIG_MAX_JOBS]) {          // one workgroup past the column blocks: the fused head's loss (launched only then)
    __shared__ double lred[2][CE_THREADS / 64];
    LossFin lf = loss_in;
    const int64_t foff_ = (int64_t)fc.slot[blockIdx.y] * fc.stride;
    FS(lf.logits); FS(lf.labels); FS(lf.lossbuf); FS(lf.lacc);
    loss_finalize(lf, lred);
    return;
  }
  const int ji = colsum_find_job(jobs, blockIdx.x);
  ColsumJob jb = jobs.j[ji];
  AdamArgs ad = ad_in;
  const int64_t foff_ = (int64_t)fc.slot[blockIdx.y] * fc.stride;
  FS(jb.part); FS(jb.out); FS(ad.p); FS(ad.g); FS(ad.m); FS(ad.v);
  ad.lr_over_bc1 = fc.lr_over_bc1[blockIdx.y];
  ad.inv_sqrt_bc2 = fc.inv_sqrt_bc2[blockIdx.y];
  const int cx = threadIdx.x & (CS_COLS - 1), ry = threadIdx.x / CS_COLS;
  const int c = ((int)blockIdx.x - jobs.blk0[ji]) * CS_COLS + cx;
  if (jb.nrows > 0) {
    red[ry][cx] = c < jb.ncols ? colsum_lane(jb.part + jb.col0 + c, jb.nrows, (size_t)jb.row_stride, ry) : 0.0;
    __syncthreads();
  }
  if (ry == 0 && c < jb.ncols) {
    float gsum;
    if (jb.nrows > 0) { gsum = (float)colsum_fold(red, cx); jb.out[c] = gsum; }
    else gsum = jb.out[c];
    // torch.optim.Adam with L2-in-gradient weight decay — the arithmetic of adam_kernel, element by element
    const int64_t i = (jb.out - ad.g) + c;
    const float gr = gsum + ad.wd * ad.p[i];
    const float mm = ad.b1 * ad.m[i] + (1.f - ad.b1) * gr;
    const float vv = ad.b2 * ad.v[i] + (1.f - ad.b2) * gr * gr;
    const float denom = sqrtf(vv) * ad.inv_sqrt_bc2 + ad.eps;
    ad.p[i] -= ad.lr_over_bc1 * (mm / denom);
    ad.m[i] = mm;
    ad.v[i] = vv;
  }
}

int launch_colsum_adam_plan(const ColsumPlan& plan, const AdamArgs& ad, const FoldCtx& fc, hipStream_t st) {
  if (plan.n <= 0) return 0;
  ColsumJobs a;
  const int nblk = colsum_fill(plan, a);
  if (nblk <= 0) return 0;
  { MSIG_K("colsum_adam", st); colsum_adam_kernel<<<dim3(nblk + (plan.loss.logits ? 1 : 0), fc.n), 256, 0, st>>>(a, ad, plan.loss, fc); }
  MSIG_LAUNCH_CHECK();
  return 0;
}

int launch_colsum_plan(const ColsumPlan& plan, const FoldCtx& fc, hipStream_t st) {
  if (plan.n <= 0) return 0;
  ColsumJobs a;
  const int nblk = colsum_fill(plan, a);
  if (nblk <= 0) return 0;
  { MSIG_K("colsum", st); colsum_plan_kernel<<<dim3(nblk, fc.n), 256, 0, st>>>(a, fc); }
  MSIG_LAUNCH_CHECK();
  return 0;
}

PartOffsets part_offsets(const StageDims& d) {
  PartOffsets o;
  const int64_t units = (int64_t)d.NT * d.TP;
  o.gru_rows = (int)(units < MSIG_DW_WG ? units : MSIG_DW_WG);
  int64_t at = 0;
  auto take = [&](int64_t n) { const int64_t r = at; at += (n + 63) / 64 * 64; return r; };
  o.head = take((int64_t)HEAD_WG * (64 * 128 + 64 + d.K * 64 + d.K));
  o.l1 = take(2 * (int64_t)o.gru_rows * (192 * 128 + 192 * 64 + 256));
  o.l0 = take(2 * (int64_t)o.gru_rows * (192 * 32 + 192 * 64 + 256));
  o.conv2 = take((int64_t)MSIG_CONV_DW_WG * 2560);
  o.conv1 = take((int64_t)MSIG_CONV_DW_WG * 16 * d.C * 7);
  o.total = at;
  return o;
}

// ------------------------------------------------------------------------------------
// Adam with L2 weight decay folded into the gradient (torch.optim.Adam semantics)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n4, float lr_over_bc1, float inv_sqrt_bc2,
                                                   float b1, float b2, float eps, float wd) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pp = ((float4*)p)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
    const float4 gg = ((const float4*)g)[i];
    float pa[4] = {pp.x, pp.y, pp.z, pp.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
    const float ga[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = ga[e] + wd * pa[e];
      ma[e] = b1 * ma[e] + (1.f - b1) * gr;
      va[e] = b2 * va[e] + (1.f - b2) * gr * gr;
      const float denom = sqrtf(va[e]) * inv_sqrt_bc2 + eps;
      pa[e] -= lr_over_bc1 * (ma[e] / denom);
    }
    ((float4*)p)[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
    ((float4*)m)[i] = make_float4(ma[0], ma[1], ma[2], ma[3]);
    ((float4*)v)[i] = make_float4(va[0], va[1], va[2], va[3]);
  }
}

// blockIdx.z = fold (msig_gather_windows_multi): the store is shared, idx is (n, B) contiguous, the outputs are per-arena
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ store, const int64_t* __restrict__ store_y,
                                                     const int64_t* __restrict__ idx, int64_t idx_row_stride, int64_t w4,
                                                     float* __restrict__ ox, int64_t* __restrict__ oy, const FoldCtx fc) {
  FOLD_BEGIN; FS(ox); FS(oy);
  const int i = blockIdx.y;
  const int64_t src = idx[(int64_t)blockIdx.z * idx_row_stride + i];
  const float4* s4 = (const float4*)(store) + src * w4;
  float4* d4 = (float4*)(ox) + (int64_t)i * w4;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < w4; j += (int64_t)gridDim.x * 256) d4[j] = s4[j];
  if (oy && store_y && blockIdx.x == 0 && threadIdx.x == 0) oy[i] = store_y[src];
}

// ------------------------------------------------------------------------------------
// Per-subject normalisation of raw windows (dataset.py:36-48) + the (N,T,C)->(N,C,T) fp32 layout
// ------------------------------------------------------------------------------------
#define NORM_WG 512
struct NormCols { int col[MSIG_MAX_C]; };

__global__ __launch_bounds__(256) void norm_stats_kernel(const double* __restrict__ raw, int64_t rows, int C_all, NormCols cols,
                                                         int C, uint32_t log1p_mask, double* __restrict__ part) {
  __shared__ double red[4][2 * MSIG_MAX_C];
  double s1[MSIG_MAX_C], s2[MSIG_MAX_C];
#pragma unroll
  for (int c = 0; c < MSIG_MAX_C; ++c) s1[c] = s2[c] = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    const double* row = raw + r * C_all;
#pragma unroll
    for (int c = 0; c < MSIG_MAX_C; ++c)
      if (c < C) {
        double v = row[cols.col[c]];
        if ((log1p_mask >> c) & 1u) v = log1p(v);
        s1[c] += v; s2[c] += v * v;
      }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < MSIG_MAX_C; ++c)
    if (c < C) {
      const double a = wave_sum_d(s1[c]), b = wave_sum_d(s2[c]);
      if (lane == 0) { red[w][c] = a; red[w][MSIG_MAX_C + c] = b; }
    }
  __syncthreads();
  if (threadIdx.x < 2 * MSIG_MAX_C)
    part[(size_t)blockIdx.x * 2 * MSIG_MAX_C + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void norm_finalize_kernel(const double* __restrict__ part, int nparts, int C, double count,
                                                           double* __restrict__ stats) {
  const int c = threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int i = 0; i < nparts; ++i) { a += part[(size_t)i * 2 * MSIG_MAX_C + c]; b += part[(size_t)i * 2 * MSIG_MAX_C + MSIG_MAX_C + c]; }
  const double mean = a / count;
  double var = b / count - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[c] = mean;
  stats[MSIG_MAX_C + c] = 1.0 / (sqrt(var) + 1e-8);
}

__global__ __launch_bounds__(256) void norm_apply_kernel(const double* __restrict__ raw, int64_t N, int T, int C_all, NormCols cols, int C,
                                                         uint32_t log1p_mask, const double* __restrict__ stats, float* __restrict__ out) {
  const int64_t rows = N * T;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    const int64_t n = r / T;
    const int t = (int)(r - n * T);
    const double* row = raw + r * C_all;
#pragma unroll
    for (int c = 0; c < MSIG_MAX_C; ++c)
      if (c < C) {
        double v = row[cols.col[c]];
        if ((log1p_mask >> c) & 1u) v = log1p(v);
        out[((size_t)n * C + c) * T + t] = (float)((v - stats[c]) * stats[MSIG_MAX_C + c]);
      }
  }
}

int launch_normalise(const double* raw, int64_t N, int T, int C_all, const int* cols, int C, uint32_t mask, float* out, void* scratch,
                     hipStream_t st) {
  NormCols nc;
  for (int c = 0; c < MSIG_MAX_C; ++c) nc.col[c] = c < C ? cols[c] : 0;
  double* part = (double*)scratch;
  double* stats = part + (size_t)NORM_WG * 2 * MSIG_MAX_C;
  const int64_t rows = N * T;
  int grid = (int)((rows + 255) / 256);
  if (grid > NORM_WG) grid = NORM_WG;
  if (grid < 1) grid = 1;
  { MSIG_K("norm_stats", st); norm_stats_kernel<<<grid, 256, 0, st>>>(raw, rows, C_all, nc, C, mask, part); }
  MSIG_LAUNCH_CHECK();
  { MSIG_K("norm_finalize", st); norm_finalize_kernel<<<1, 64, 0, st>>>(part, grid, C, (double)rows, stats); }
  MSIG_LAUNCH_CHECK();
  int g2 = (int)((rows + 255) / 256);
  if (g2 > 8192) g2 = 8192;
  { MSIG_K("norm_apply", st); norm_apply_kernel<<<g2, 256, 0, st>>>(raw, N, T, C_all, nc, C, mask, stats, out); }
  MSIG_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------
// Host launchers
// ------------------------------------------------------------------------------------
int launch_head_fwd(const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po, const FoldCtx& fc, hipStream_t st) {
  const float* P = b->params;
  const int thr = b->training ? b->dropout_thr : 0;
  const int ngroups = (d.B + HEAD_ROWS - 1) / HEAD_ROWS;
  const int grid = ngroups < 1024 ? ngroups : 1024;
  { MSIG_K("head_fwd", st); head_fwd_kernel<<<dim3(grid, 1, fc.n), 256, 0, st>>>(w.p<float>(MSIG_WS_FEAT), P + po[MSIG_P_CLS0_W], P + po[MSIG_P_CLS0_B], P + po[MSIG_P_CLS3_W],
                                         P + po[MSIG_P_CLS3_B], w.p<float>(MSIG_WS_HID), w.p<float>(MSIG_WS_LOGITS), d.B, d.K, thr,
                                         b->key_head, drop_scale(thr), fc); }
  MSIG_LAUNCH_CHECK();
  if (b->labels) {
    MSIG_K("ce", st);
    ce_kernel<<<dim3(1, 1, fc.n), CE_THREADS, 0, st>>>(w.p<float>(MSIG_WS_LOGITS), b->labels, w.p<float>(MSIG_WS_PROBS), w.p<int>(MSIG_WS_PRED),
                                 b->training ? w.p<float>(MSIG_WS_DLOGITS) : nullptr, w.p<float>(MSIG_WS_LOSS), b->loss_acc, d.B, d.K, fc);
  } else {
    MSIG_K("softmax", st);
    softmax_kernel<<<dim3((d.B + 255) / 256, 1, fc.n), 256, 0, st>>>(w.p<float>(MSIG_WS_LOGITS), w.p<float>(MSIG_WS_PROBS), w.p<int>(MSIG_WS_PRED), d.B, d.K, fc);
  }
  MSIG_LAUNCH_CHECK();
  return 0;
}

int launch_head_bwd(const msig_batch* b, const float* dlogits, const StageDims& d, const WsPtrs& w, const int64_t* po, ColsumPlan& plan,
                    const FoldCtx& fc, hipStream_t st) {
  const float* P = b->params;
  float* G = b->grads;
  const int thr = b->training ? b->dropout_thr : 0;
  const int ngroups = (d.B + HEAD_ROWS - 1) / HEAD_ROWS;
  const int grid = ngroups < HEAD_WG ? ngroups : HEAD_WG;
  float* part = w.p<float>(MSIG_WS_GRAD_PART) + part_offsets(d).head;
  const int PS = 64 * 128 + 64 + d.K * 64 + d.K;
  { MSIG_K("head_bwd", st); head_bwd_kernel<<<dim3(grid, 1, fc.n), 256, 0, st>>>(dlogits ? dlogits : w.p<float>(MSIG_WS_DLOGITS), w.p<float>(MSIG_WS_FEAT), w.p<float>(MSIG_WS_HID),
                                         P + po[MSIG_P_CLS0_W], P + po[MSIG_P_CLS3_W], w.p<float>(MSIG_WS_DFEAT), part, d.B, d.K,
                                         thr > 0 ? drop_scale(thr) : 1.0f, fc); }
  MSIG_LAUNCH_CHECK();
  const bool ok = plan.add(part, grid, PS, 0, 64 * 128, G + po[MSIG_P_CLS0_W]) && plan.add(part, grid, PS, 64 * 128, 64, G + po[MSIG_P_CLS0_B]) &&
                  plan.add(part, grid, PS, 64 * 128 + 64, d.K * 64, G + po[MSIG_P_CLS3_W]) &&
                  plan.add(part, grid, PS, 64 * 128 + 64 + d.K * 64, d.K, G + po[MSIG_P_CLS3_B]);
  return ok ? 0 : MSIG_E_SHAPE;
}

// The fused head applies where a workgroup per group of 16 rows is also head_bwd_kernel's grid (one partial row per group).
bool head_step_applies(const msig_batch* b, const StageDims& d) {
  return b->training && b->labels && (d.B + HEAD_ROWS - 1) / HEAD_ROWS <= HEAD_WG;
}
int launch_head_step(const msig_batch* b, const StageDims& d, const WsPtrs& w, const int64_t* po, ColsumPlan& plan, const FoldCtx& fc, hipStream_t st) {
  const float* P = b->params;
  float* G = b->grads;
  const int thr = b->dropout_thr;
  const int grid = (d.B + HEAD_ROWS - 1) / HEAD_ROWS;
  float* part = w.p<float>(MSIG_WS_GRAD_PART) + part_offsets(d).head;
  const int PS = 64 * 128 + 64 + d.K * 64 + d.K;
  { MSIG_K("head_step", st); head_step_kernel<<<dim3(grid, 1, fc.n), 256, 0, st>>>(w.p<float>(MSIG_WS_FEAT), P + po[MSIG_P_CLS0_W], P + po[MSIG_P_CLS0_B], P + po[MSIG_P_CLS3_W],
                                         P + po[MSIG_P_CLS3_B], b->labels, w.p<float>(MSIG_WS_HID), w.p<float>(MSIG_WS_LOGITS), w.p<float>(MSIG_WS_PROBS), w.p<int>(MSIG_WS_PRED),
                                         w.p<float>(MSIG_WS_DLOGITS), w.p<float>(MSIG_WS_DFEAT), part, d.B, d.K, thr, b->key_head, drop_scale(thr),
                                         thr > 0 ? drop_scale(thr) : 1.0f, fc); }
  MSIG_LAUNCH_CHECK();
  plan.loss = LossFin{w.p<float>(MSIG_WS_LOGITS), b->labels, w.p<float>(MSIG_WS_LOSS), b->loss_acc, d.B, d.K};
  const bool ok = plan.add(part, grid, PS, 0, 64 * 128, G + po[MSIG_P_CLS0_W]) && plan.add(part, grid, PS, 64 * 128, 64, G + po[MSIG_P_CLS0_B]) &&
                  plan.add(part, grid, PS, 64 * 128 + 64, d.K * 64, G + po[MSIG_P_CLS3_W]) &&
                  plan.add(part, grid, PS, 64 * 128 + 64 + d.K * 64, d.K, G + po[MSIG_P_CLS3_B]);
  return ok ? 0 : MSIG_E_SHAPE;
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                int64_t step, hipStream_t st) {
  const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  { MSIG_K("adam", st); adam_kernel<<<(int)blocks, 256, 0, st>>>(p, g, m, v, n4, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), b1, b2, eps, wd); }
  MSIG_LAUNCH_CHECK();
  return 0;
}

int launch_gather(const float* store, const int64_t* sy, const int64_t* idx, int64_t idx_row_stride, int B, int64_t wfloats, float* ox, int64_t* oy,
                  const FoldCtx& fc, hipStream_t st) {
  const int64_t w4 = wfloats / 4;
  int gx = (int)((w4 + 255) / 256);
  if (gx > 64) gx = 64;
  { MSIG_K("gather", st); gather_kernel<<<dim3(gx, B, fc.n), 256, 0, st>>>(store, sy, idx, idx_row_stride, w4, ox, oy, fc); }
  MSIG_LAUNCH_CHECK();
  return 0;
}
