/* msig.h — C ABI of libmsig_hip.so: the MI355X (gfx950) training/eval path of
 * 17LiQi/MultimodalSignal's CnnGruAttentionModel.
 *
 * The reference has no FFI: its boundary for this path is the Python class API
 * (models.py:39-81, trainer.py:130-153,193-247).  Every entry point below states
 * which reference call it stands in for (paths are into the reference tree).  A
 * maintainer binds these with ctypes; multimodalsignal_amd/
 * is exactly such a binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless marked "host"; the caller owns all
 *    memory (the library allocates nothing and keeps no state);
 *  - every launcher is asynchronous on `stream` (a hipStream_t passed as void*),
 *    never synchronises, and is safe to capture into a hipGraph;
 *  - return value: 0 ok; >0 a hipError_t from a launch; <0 an MSIG_E_* argument
 *    error (nothing was launched);
 *  - all arithmetic is fp32 (labels int64, counters int64), like the reference.  The GRU contractions run as SPLIT products on the
 *    16-bit matrix pipes with fp32 accumulation — three bf16 pieces per operand (backward, layer-0 projection) or two fp16 pieces
 *    (forward recurrences, layer-1 projection) — at an error no larger than the fp32 matrix instruction's own.
 */
#ifndef MSIG_H
#define MSIG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSIG_ABI_VERSION 4      /* 2: msig_multi.form_folds, msig_struct_bytes, kernel forms; 3: msig_multi.step (per-fold Adam step);
                                   4: kernel forms are per call (msig_batch.fwd_form / bwd_form; msig_set_kernel_form is gone),
                                      MSIG_E_FORM, the two-vector stash only inside the fused train-step calls, msig_batch.loss_acc */

#define MSIG_E_NULL      (-1)  /* a required pointer is NULL                         */
#define MSIG_E_SHAPE     (-2)  /* B/C/T/K outside the supported range                */
#define MSIG_E_ALIGN     (-3)  /* a buffer is not 16-byte aligned                    */
#define MSIG_E_WORKSPACE (-4)  /* ws_bytes smaller than msig_workspace_bytes()       */
#define MSIG_E_FORM      (-5)  /* msig_batch.fwd_form / bwd_form is not a kernel form, or not one this call can run (fold batches) */

/* Fixed architecture of the hot path (models.py:39-40 defaults; main.py:48-55). */
#define MSIG_CNN1   16   /* Conv1d(C,16,k7,s2,p3)   models.py:46 */
#define MSIG_CNN2   32   /* Conv1d(16,32,k5,s2,p2)  models.py:50 */
#define MSIG_HID    64   /* GRU hidden size         models.py:58 */
#define MSIG_GATES  192  /* 3*MSIG_HID, rows [r;z;n] */
#define MSIG_HEAD   64   /* Linear(128,64)          models.py:67 */
#define MSIG_MAX_C  16
#define MSIG_MAX_K  16

/* ---- parameter tensors, in nn.Module.parameters() order (models.py:43-71) ---- */
enum msig_param {
  MSIG_P_GATE_W1 = 0,  /* channel_attention.fc.0.weight (C/4, C)   models.py:18 */
  MSIG_P_GATE_W2,      /* channel_attention.fc.2.weight (C, C/4)   models.py:20 */
  MSIG_P_CONV1_W,      /* cnn_encoder.0.weight (16, C, 7)                       */
  MSIG_P_BN1_G,        /* cnn_encoder.1.weight (16)                             */
  MSIG_P_BN1_B,        /* cnn_encoder.1.bias   (16)                             */
  MSIG_P_CONV2_W,      /* cnn_encoder.4.weight (32, 16, 5)                      */
  MSIG_P_BN2_G,        /* cnn_encoder.5.weight (32)                             */
  MSIG_P_BN2_B,        /* cnn_encoder.5.bias   (32)                             */
  MSIG_P_GRU,          /* 16 tensors: for l in {0,1}, dir in {fwd,reverse}:     */
                       /*   weight_ih (192,I) weight_hh (192,64) bias_ih bias_hh */
  MSIG_P_CLS0_W = MSIG_P_GRU + 16, /* classifier.0.weight (64,128) models.py:67 */
  MSIG_P_CLS0_B,       /* classifier.0.bias (64)                                */
  MSIG_P_CLS3_W,       /* classifier.3.weight (K,64)               models.py:70 */
  MSIG_P_CLS3_B,       /* classifier.3.bias (K)                                 */
  MSIG_NPARAM
};
#define MSIG_P_GRU_T(layer, dir, which) (MSIG_P_GRU + ((layer) * 2 + (dir)) * 4 + (which))

/* Offsets (in floats, each a multiple of 4) of every tensor inside the flat
 * parameter buffer; offsets[MSIG_NPARAM] is the padded total.  The same layout is
 * used for gradients and both Adam moments.  Replaces nothing in the reference —
 * it is what lets `model.parameters()` live in one buffer for msig_adam_step. */
int msig_param_layout(int C, int K, int64_t* offsets /* host, [MSIG_NPARAM+1] */);

/* BatchNorm buffers (cnn_encoder.{1,5}.running_mean/var, models.py:47,51):
 * bn_state = [rm1(16) rv1(16) rm2(32) rv2(32)] floats; bn_count = 2 x int64
 * num_batches_tracked. */
#define MSIG_BN_STATE_FLOATS 96

/* ---- workspace regions (activations, stashes, gradient scratch) ------------- */
enum msig_ws {
  MSIG_WS_GATE_MEAN = 0, /* (B,C)      mean over T                models.py:28 */
  MSIG_WS_GATE_PRE,      /* (B,max(C/4,1)) fc.0 output before ReLU             */
  MSIG_WS_GATE_S,        /* (B,C)      sigmoid gate               models.py:29 */
  MSIG_WS_Y1,            /* (B,L1,16)  conv1 output, time-major (NLC)          */
  MSIG_WS_BN1_PART,      /* partial sums for BatchNorm-1 statistics            */
  MSIG_WS_BN1_STAT,      /* mean,invstd,scale,shift (4 x 16)                   */
  MSIG_WS_P1,            /* (B,P1,16)  after BN+ReLU+MaxPool (NLC)             */
  MSIG_WS_Y2,            /* (B,L2,32)  conv2 output (NLC)                      */
  MSIG_WS_BN2_PART,
  MSIG_WS_BN2_STAT,      /* 4 x 32                                             */
  MSIG_WS_P2,            /* (B,TP,32)  == x.permute(0,2,1)        models.py:77 */
  MSIG_WS_H0,            /* (B,TP,128) GRU layer-0 outputs [fwd|rev]           */
  MSIG_WS_H1,            /* (B,TP,64)  GRU layer-1 forward-direction outputs   */
  MSIG_WS_STASH0,        /* layer-0 gate stash, 2 directions: four float4 slots per lane and step — the forward pass fills
                            r, z, - , W_hn h + b_hn (n is recovered from h by the backward pass); the latency-form backward
                            overwrites all four with dr, dz, dn, dhn for its bulk dX / dW kernels */
  MSIG_WS_STASH1,        /* layer-1 forward direction stash                    */
  MSIG_WS_STASH1R,       /* layer-1 reverse direction, single step             */
  MSIG_WS_FEAT,          /* (B,128)    outputs[:, -1, :]          models.py:79 */
  MSIG_WS_HID,           /* (B,64)     classifier hidden after ReLU+Dropout    */
  MSIG_WS_LOGITS,        /* (B,K)                                              */
  MSIG_WS_PROBS,         /* (B,K)      softmax                  trainer.py:224 */
  MSIG_WS_PRED,          /* (B) int32  argmax                   trainer.py:225 */
  MSIG_WS_LOSS,          /* of this batch: [0] = mean CE, [1] = summed CE (mean * B), [2] = #correct (plain stores; epoch sums are the caller's) */
  MSIG_WS_DLOGITS,       /* (B,K)                                              */
  MSIG_WS_DFEAT,         /* (B,128)                                            */
  MSIG_WS_DH0,           /* (B,TP,128) grad wrt (dropped) layer-0 outputs      */
  MSIG_WS_DX0,           /* (2,B,TP,32) grad wrt P2 from each layer-0 direction */
  MSIG_WS_DY2,           /* (B,L2,32)  dL/d(bn2 output); BN-backward pass 2 is fused into the conv2 backward kernels */
  MSIG_WS_DP1,           /* (B,P1,16)                                          */
  MSIG_WS_DS,            /* (B,C)      grad wrt gate                           */
  MSIG_WS_BNB_PART,      /* partial sums for BatchNorm backward                */
  MSIG_WS_BNB_STAT,      /* c1,c2 per channel (2 x 32)                         */
  MSIG_WS_GRAD_PART,     /* per-workgroup partial weight gradients             */
  MSIG_WS_GI,            /* layer-1 input projections (small batches only: < 192 batch tiles) */
  MSIG_WS_POOLC1,        /* (B,P1,4) bytes: MaxPool-1 decisions, 2 bits per channel (training only)      */
  MSIG_WS_POOLC2,        /* (B,TP,8) bytes: MaxPool-2 decisions                                          */
  MSIG_WS_G1W,           /* (B, S, 2, 16, 16*ceil(7C/16)) per-window pieces of conv1's weight-gradient correlation, [sum dz x][sum xhat x]: the
                            BatchNorm-1 backward is applied to them by linearity (training only).  S = 1 from B = 256 on; smaller
                            batches cut a window into up to 8 segments, one record each (room for 8 is reserved)        */
  MSIG_WS_GATE_EO,       /* (B, C, 2)  sums of the even- / odd-indexed samples of a channel (training only)              */
  MSIG_NWS
};

typedef struct msig_shape {
  int32_t B;   /* windows in the batch (>=1)                                   */
  int32_t C;   /* input channels, 1..MSIG_MAX_C      (main.py:47)              */
  int32_t T;   /* samples per window (3840 = 60 s @ 64 Hz; any T >= 16)        */
  int32_t K;   /* classes, 2..MSIG_MAX_K             (main.py:45)              */
} msig_shape;

/* Temporal sizes after conv1 / pool1 / conv2 / pool2 (models.py:46-53). */
int msig_stage_lengths(int T, int32_t* out /* host, [4] = L1,P1,L2,TP */);

/* Byte offset of every workspace region (256-byte aligned); offsets[MSIG_NWS] is
 * the total.  `training`=0 omits stashes and gradient scratch. */
int msig_workspace_layout(const msig_shape* s, int training, int64_t* offsets /* host, [MSIG_NWS+1] */);
int64_t msig_workspace_bytes(const msig_shape* s, int training);

/* One mini-batch of work: everything trainer.py:140-149 touches. */
typedef struct msig_batch {
  msig_shape shape;
  int32_t  training;      /* 1 = model.train(): batch-stat BN, dropout, stashes; 0 = model.eval() */
  float    bn_momentum;   /* 0.1   (nn.BatchNorm1d default)                    */
  float    bn_eps;        /* 1e-5                                              */
  int32_t  dropout_thr;   /* round(p*256): element kept iff hash byte >= thr; 0 disables */
  uint32_t key_gru;       /* dropout key, inter-layer GRU dropout (models.py:62) */
  uint32_t key_head;      /* dropout key, classifier dropout      (models.py:69) */
  const float*   x;       /* (B,C,T) contiguous                                */
  const int64_t* labels;  /* (B) or NULL when only logits are wanted           */
  const float*   params;  /* flat, msig_param_layout                           */
  float*         grads;   /* flat, same layout; written (not accumulated) by the *_bwd calls */
  float*         bn_state;/* MSIG_BN_STATE_FLOATS                              */
  int64_t*       bn_count;/* [2]                                               */
  void*          ws;      /* workspace                                         */
  int64_t        ws_bytes;
  int32_t  gru_layers;    /* 0 or 2: the reference's 2-layer bidirectional GRU (models.py:56-63); 1: ONE layer — outputs[:, -1, :]
                             is then layer 0's state at the last position (forward direction's final state, reverse direction's first
                             step) and there is no inter-layer dropout: the hierarchical experiment's second model (main.py:35-40,
                             gru_hidden_size = 32, gru_num_layers = 1) runs on this path with its 32 units embedded in the 64-unit
                             layout (padded units stay exactly zero and receive exactly zero gradient; models.py of this repo) */
  int16_t  fwd_form;      /* GRU kernel forms of THIS call: 0 = the default (by batch size; a process may preset another default ONCE */
  int16_t  bwd_form;      /* through the environment, see "Kernel forms" below), MSIG_FWD_x + 1 / MSIG_BWD_x + 1 = that form.  Per call, */
                          /* not per process: two host threads may drive different models with different forms.                          */
  double*  loss_acc;      /* optional (NULL = unused): two running sums kept OUTSIDE the workspace, updated by the loss kernel of every call
                             with labels: loss_acc[0] += summed CrossEntropy of this batch (loss.item() * B, trainer.py:152,221),
                             loss_acc[1] += correctly classified windows.  One thread adds, in stream order (deterministic); the caller
                             zeroes them when a pass starts and reads them when it ends — the per-step accumulation launch and host op of
                             trainer.py:152-153 disappear.  In a fold batch fold s's pair sits s * stride_bytes further on, like every buffer */
} msig_batch;

/* ChannelAttention + cnn_encoder forward (models.py:75-76): x -> WS_P2. */
int msig_frontend_fwd(const msig_batch* b, void* stream);
/* 2-layer bidirectional GRU + outputs[:, -1, :] (models.py:77-79): WS_P2 -> WS_FEAT. */
int msig_gru_fwd(const msig_batch* b, void* stream);
/* classifier (models.py:80) -> WS_LOGITS; when labels != NULL also CrossEntropyLoss
 * (trainer.py:147), softmax/argmax (trainer.py:224-225) and WS_DLOGITS. */
int msig_head_ce_fwd(const msig_batch* b, void* stream);
/* autograd backward of the three stages above (trainer.py:148), in reverse order.
 * `dlogits` NULL means: use WS_DLOGITS produced by msig_head_ce_fwd. */
int msig_head_ce_bwd(const msig_batch* b, const float* dlogits, void* stream);
int msig_gru_bwd(const msig_batch* b, void* stream);
int msig_frontend_bwd(const msig_batch* b, void* stream);

/* model(inputs) (trainer.py:146,217): frontend + GRU + head in one call. */
int msig_forward(const msig_batch* b, void* stream);
/* loss.backward() (trainer.py:148): all three *_bwd in order. */
int msig_backward(const msig_batch* b, const float* dlogits, void* stream);

/* torch.optim.Adam(lr, weight_decay as L2-in-gradient).step() over n floats
 * (trainer.py:68,149).  `step` is the 1-based count of this update. */
int msig_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                   void* stream);

/* optimizer.zero_grad(); forward; CE; backward; Adam — trainer.py:144-149 as one call. */
int msig_train_step(const msig_batch* b, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int64_t step, void* stream);

/* ---- fold batching: several independent models per launch (main.py:98-125's fold loop) -------------
 * At the reference's batch size (64 windows = 4 batch tiles) one model's step is ~30 launches that each keep a few CUs busy,
 * and fifteen folds on fifteen streams are bound by the command processor's dispatch rate.  The *_multi entry points run
 * the SAME step for n models in ONE set of launches (blockIdx.z = fold): `b` describes fold slot 0 — parameters, gradients,
 * BN state, workspace, input batch and labels — and the corresponding buffers of slot s live exactly s * stride_bytes further
 * on, for every pointer in `b` and for exp_avg / exp_avg_sq alike (one arena per fold, identical offsets inside each).
 * Shape, flags and dropout threshold are shared; dropout keys and learning rates are per fold.  Folds never interact: every
 * reduction (BatchNorm statistics, weight gradients, loss) stays inside its arena, so each fold's results are bit-identical
 * to the single-model call with the same backward GRU kernel form.  Batches below 192 tiles of 16 windows only.  Forward: the
 * latency form and gru_fwd_ws agree in every bit, and the library picks per layer and per launch (layer 0 gru_fwd_ws, layer 1 from
 * 32 tiles over the folds of the launch on).  Backward: the latency form, or — from 12 tiles over the folds of the launch on — the
 * fused kernels (gru_bwd_b6 / gru_bwd_b3 with per-fold pointers).  The two BACKWARD forms round differently (both within the parity
 * tolerance of the oracle): pin one with msig_batch.bwd_form, or the fold count the choice is made for with
 * msig_multi.form_folds, where a fold's numbers must not depend on its companions. */
#define MSIG_MAX_FOLDS 16
typedef struct msig_multi {
  int32_t  n;                        /* folds in this launch, 1..MSIG_MAX_FOLDS                          */
  int32_t  slot[MSIG_MAX_FOLDS];     /* arena index of each (distinct)                                    */
  int64_t  stride_bytes;             /* bytes between consecutive arenas; multiple of 256                 */
  uint32_t key_gru[MSIG_MAX_FOLDS];  /* per-fold dropout keys (msig_batch.key_gru / key_head are ignored) */
  uint32_t key_head[MSIG_MAX_FOLDS];
  float    lr[MSIG_MAX_FOLDS];       /* per-fold learning rate (msig_train_step_multi)                    */
  int32_t  form_folds;               /* fold count the BACKWARD GRU kernel form is chosen for; 0 = n, the folds in this launch.  A caller
                                        that wants every fold's rounding independent of how many folds are still active
                                        pins it (or the form itself, msig_batch.fwd_form / bwd_form)                 */
  int64_t  step[MSIG_MAX_FOLDS];     /* per-fold optimiser step count (Adam bias correction, msig_train_step_multi); 0 = the call's
                                        `step`.  Folds whose training sets differ in size take different numbers of steps per
                                        epoch (main.py:98-125 on real WESAD) and still share launches                          */
} msig_multi;
int msig_forward_multi(const msig_batch* b, const msig_multi* m, void* stream);
int msig_train_step_multi(const msig_batch* b, const msig_multi* m, float* exp_avg, float* exp_avg_sq, float beta1, float beta2,
                          float eps, float weight_decay, int64_t step, void* stream);

/* Host-side dropout key (same mixing as oracle/cnn_gru_oracle.py dropout_key). */
uint32_t msig_dropout_key(uint64_t seed, uint64_t step, uint32_t stream_id);

/* dataset.py:62-65 batched: gathers windows idx[0..B) from a device-resident
 * (N,C,T) fp32 store into a contiguous (B,C,T) batch plus labels. */
int msig_gather_windows(const float* store, const int64_t* store_labels, const int64_t* idx, int32_t B,
                        int64_t window_floats, float* out_x, int64_t* out_y, void* stream);

/* The same for a fold batch: row z of idx (rows idx_row_stride elements apart, >= B) holds fold z's window indices into
 * the SHARED store, and the outputs of fold z land in arena m->slot[z] (out_x / out_y are arena 0's buffers). */
int msig_gather_windows_multi(const float* store, const int64_t* store_labels, const int64_t* idx, int64_t idx_row_stride, int32_t B,
                              int64_t window_floats, float* out_x, int64_t* out_y, const msig_multi* m, void* stream);

/* dataset.py:36-48 and :62-65 for ONE subject, on the device: raw (N,T,C_all) float64 as written by
 * preprocess.py:217-222 -> out (N,C,T) fp32.  Per selected channel c (column cols[c] of raw): v = raw value,
 * or log1p(v) when bit c of log1p_mask is set ('chest_EDA', dataset.py:42); out = (v - mean) / (std + 1e-8)
 * with mean / population std over all N*T samples of the channel, accumulated in float64.
 * scratch: >= msig_normalise_scratch_bytes() bytes of device memory.  cols is a HOST array (C <= MSIG_MAX_C). */
int64_t msig_normalise_scratch_bytes(void);
int msig_normalise_subject(const double* raw, int64_t N, int32_t T, int32_t C_all, const int32_t* cols /* host */, int32_t C,
                           uint32_t log1p_mask, float* out, void* scratch, void* stream);

/* ChannelAttention.forward on its own (models.py:24-31): s = sigmoid(W2 relu(W1 mean_T(x))) (s == 0.5 for C < 4, where the
 * hidden layer is empty), out = x * s[:, :, None].  x, out: (B,C,T) fp32; s: (B,C); w1: (C/4,C), w2: (C,C/4) (may be NULL for
 * C < 4); scratch: B * (C + C/4) floats.  Inside msig_forward the product is never written (folded into conv1's taps). */
int msig_channel_attention(const float* x, const float* w1, const float* w2, int32_t B, int32_t C, int32_t T, float* out, float* s,
                           float* scratch, void* stream);

int msig_abi_version(void);
/* sizeof(msig_batch) (which = 0) / sizeof(msig_multi) (which = 1) as this library was compiled: lets a binding that mirrors the
 * structs (ctypes, cgo, JNA ...) check its layout at load time instead of corrupting a launch.  Other values: -1. */
int64_t msig_struct_bytes(int32_t which);

/* Kernel forms of the GRU launches (diagnostics / tests).  msig_batch.fwd_form / bwd_form select them PER CALL (0 = default,
 * MSIG_FWD_x + 1 / MSIG_BWD_x + 1 = pinned); the default picks by batch size — throughput forms at >= 192 batch tiles of 16
 * windows; below, the latency backward and, forward, gru_fwd_ws for layer 0 and (from 32 tiles per launch on) layer 1, projection +
 * recurrence for layer 1 otherwise: MSIG_FWD_LATENCY and MSIG_FWD_WS give the same bits — unless the environment variables MSIG_GRU_FWD (ws|fused|split|fp32) / MSIG_GRU_BWD
 * (b6|b5|b4|b3|split) name another default: they are read ONCE, at the first launch of the process, and immutable afterwards (the
 * library has no mutable process-global state besides the profiling aid below).
 *   forward : MSIG_FWD_LATENCY  gru_fwd_proj + gru_fwd_rec (bulk projection + lean recurrence, two-piece fp16 / split-bf16 as gru_fwd_ws;
 *                                             needs < 192 tiles, else MSIG_FWD_WS runs)
 *             MSIG_FWD_B3       gru_fwd_b3   (projection fused, split-bf16 MFMA)
 *             MSIG_FWD_FP32     gru_fwd_seq  (projection fused, fp32 MFMA)
 *             MSIG_FWD_WS       gru_fwd_ws   (wave-specialised: recurrence waves + projection waves; recurrence and layer-1 projection
 *                                             on two-piece fp16 MFMA, layer-0 projection on split-bf16; the default throughput form)
 *   backward: MSIG_BWD_SPLIT    gru_bwd_seq4 (layer 1), gru_bwd_dx<128> (dX incl. the reverse step), gru_bwd_seq4_dw1 (layer 0's recurrence with
 *                                             layer 1's dW workgroups beside the chains), then gru_bwd_dxdw<32> (one model: dX and dW of layer 0 in
 *                                             one launch) or gru_bwd_dx + gru_bwd_dw2 (fold batches); every contraction on split-bf16 MFMA
 *             MSIG_BWD_FUSED    alias of MSIG_BWD_B3 (round 1's gru_bwd_fused, whose dW ran on fp32 MFMA, is gone)
 *             MSIG_BWD_B3       gru_bwd_b3    (one kernel; every contraction on split-bf16 MFMA; round 2's throughput form)
 *             MSIG_BWD_B4       gru_bwd_b4    (layer 0; the same contractions as ONE software-pipelined stream per wave: dW on
 *                                             32x32x16 tiles, the gate math in the gaps of the MFMA streams)
 *             MSIG_BWD_B5       gru_bwd_b5    (layer 0 with two waves per SIMD: four chain waves — recurrence + gate math — and
 *                                             four bulk waves — dX / dW / x staging)
 *             MSIG_BWD_B6       gru_bwd_b6    (gru_bwd_b5's division of labour; the bulk waves also stage h_prev and recompute
 *                                             W_hn h + b_hn; the default throughput form)
 *             Layer 1 runs the throughput form of its own (gru_bwd_b3<128>) under every fused form.
 * The stash contract (round 4): the layer-0 stash holds THREE vectors per step (r, z, W_hn h + b_hn) whenever forward and backward
 * are separate calls (msig_forward / msig_gru_fwd, then msig_backward / msig_gru_bwd): every backward form can consume it, whatever
 * forms the two calls name.  Only the fused calls (msig_train_step, msig_train_step_multi), which resolve both forms from ONE
 * descriptor, let gru_fwd_ws store r, z alone when the backward form is gru_bwd_b6 (2 GB less HBM traffic per B = 8192 step).
 * One process drives one GPU: the library keeps no per-device state besides the per-device opt-in to
 * > 64 KiB of dynamic LDS, which it sets for whichever device is current at the first fused-backward launch on it. */
enum { MSIG_FWD_LATENCY = 0, MSIG_FWD_B3 = 1, MSIG_FWD_FP32 = 2, MSIG_FWD_WS = 3 };
enum { MSIG_BWD_SPLIT = 0, MSIG_BWD_FUSED = 1, MSIG_BWD_B3 = 2, MSIG_BWD_B4 = 3, MSIG_BWD_B5 = 4, MSIG_BWD_B6 = 5 };

/* Profiling aid (process-global, not thread-safe, off by default): when enabled, every
 * kernel launched by this library is bracketed by hipEventRecord on ITS stream.
 * msig_profile_report synchronises those events and writes one line per kernel name:
 * "<name> <launches> <total_ms>\n" into buf (host).  Returns bytes written or <0. */
int msig_profile_enable(int on);
int64_t msig_profile_report(char* buf /* host */, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* MSIG_H */
