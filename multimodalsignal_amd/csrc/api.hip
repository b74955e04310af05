// extern "C" surface of libmsig_hip.so (declared in include/msig.h): argument checks,
// parameter / workspace layout, and the stage launch order.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "msig_dev.h"

// ---- profiling aid --------------------------------------------------------------
struct ProfRec { const char* name; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::mutex g_prof_mu;          // several host threads may launch concurrently (one stream per fold)
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static hipEvent_t prof_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
MsigProfScope::MsigProfScope(const char* n, hipStream_t s) : name(n), st(s), rec(nullptr) {
  if (!g_prof_on) return;
  hipEvent_t ea;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_recs.push_back(ProfRec{n, prof_event(), prof_event()});
    rec = (void*)(uintptr_t)g_recs.size();
    ea = g_recs.back().a;
  }
  (void)hipEventRecord(ea, st);
}
MsigProfScope::~MsigProfScope() {
  if (!rec) return;
  hipEvent_t eb;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    eb = g_recs[(size_t)(uintptr_t)rec - 1].b;
  }
  (void)hipEventRecord(eb, st);
}
extern "C" int msig_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (ProfRec& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
  g_recs.clear();
  g_prof_on = on != 0;
  return 0;
}
extern "C" int64_t msig_profile_report(char* buf, int64_t cap) {
  if (!buf || cap < 1) return MSIG_E_NULL;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  std::map<std::string, std::pair<int64_t, double>> agg;
  std::vector<std::string> order;
  for (ProfRec& r : g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) return -5;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return -5;
    if (!agg.count(r.name)) order.push_back(r.name);
    agg[r.name].first += 1; agg[r.name].second += (double)ms;
  }
  int64_t n = 0;
  for (const std::string& k : order) {
    char line[256];
    const int len = snprintf(line, sizeof line, "%s %lld %.6f\n", k.c_str(), (long long)agg[k].first, agg[k].second);
    if (n + len >= cap) break;
    memcpy(buf + n, line, (size_t)len); n += len;
  }
  buf[n] = 0;
  return n;
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                int64_t step, hipStream_t st);
int launch_normalise(const double* raw, int64_t N, int T, int C_all, const int* cols, int C, uint32_t mask, float* out, void* scratch,
                     hipStream_t st);
int launch_gather(const float* store, const int64_t* sy, const int64_t* idx, int64_t idx_row_stride, int B, int64_t wfloats, float* ox, int64_t* oy,
                  const FoldCtx& fc, hipStream_t st);

int msig_check_forms(const msig_batch* b);      // gru.hip
int msig_check_call_forms(const msig_batch* b, int n_tiles, const FoldCtx& fc);      // gru.hip: before the first launch of a call

static int check_shape(const msig_shape* s) {
  if (!s) return MSIG_E_NULL;
  if (s->B < 1 || s->C < 1 || s->C > MSIG_MAX_C || s->K < 2 || s->K > MSIG_MAX_K || s->T < 16) return MSIG_E_SHAPE;
  const StageDims d = make_dims(*s);
  if (d.TP < 1) return MSIG_E_SHAPE;
  // dropout masks and stash indices are 32-bit
  if ((int64_t)d.B * d.TP * 128 >= (int64_t)1 << 32) return MSIG_E_SHAPE;
  return 0;
}

extern "C" int msig_abi_version(void) { return MSIG_ABI_VERSION; }
extern "C" int64_t msig_struct_bytes(int32_t which) {
  return which == 0 ? (int64_t)sizeof(msig_batch) : which == 1 ? (int64_t)sizeof(msig_multi) : -1;
}

extern "C" int msig_stage_lengths(int T, int32_t* out) {
  if (!out) return MSIG_E_NULL;
  msig_shape s{1, 1, T, 2};
  const StageDims d = make_dims(s);
  out[0] = d.L1; out[1] = d.P1; out[2] = d.L2; out[3] = d.TP;
  return 0;
}

extern "C" int msig_param_layout(int C, int K, int64_t* off) {
  if (!off) return MSIG_E_NULL;
  if (C < 1 || C > MSIG_MAX_C || K < 2 || K > MSIG_MAX_K) return MSIG_E_SHAPE;
  int64_t n[MSIG_NPARAM];
  const int Cr = C / 4;
  n[MSIG_P_GATE_W1] = (int64_t)Cr * C;
  n[MSIG_P_GATE_W2] = (int64_t)C * Cr;
  n[MSIG_P_CONV1_W] = 16 * C * 7;
  n[MSIG_P_BN1_G] = 16; n[MSIG_P_BN1_B] = 16;
  n[MSIG_P_CONV2_W] = 32 * 16 * 5;
  n[MSIG_P_BN2_G] = 32; n[MSIG_P_BN2_B] = 32;
  for (int layer = 0; layer < 2; ++layer)
    for (int dir = 0; dir < 2; ++dir) {
      n[MSIG_P_GRU_T(layer, dir, 0)] = 192 * (layer ? 128 : 32);
      n[MSIG_P_GRU_T(layer, dir, 1)] = 192 * 64;
      n[MSIG_P_GRU_T(layer, dir, 2)] = 192;
      n[MSIG_P_GRU_T(layer, dir, 3)] = 192;
    }
  n[MSIG_P_CLS0_W] = 64 * 128; n[MSIG_P_CLS0_B] = 64;
  n[MSIG_P_CLS3_W] = (int64_t)K * 64; n[MSIG_P_CLS3_B] = K;
  int64_t o = 0;
  for (int i = 0; i < MSIG_NPARAM; ++i) { off[i] = o; o += (n[i] + 3) / 4 * 4; }
  off[MSIG_NPARAM] = o;
  return 0;
}

static inline int64_t imin(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t imax(int64_t a, int64_t b) { return a > b ? a : b; }

extern "C" int msig_workspace_layout(const msig_shape* s, int training, int64_t* off) {
  int rc = check_shape(s);
  if (rc) return rc;
  if (!off) return MSIG_E_NULL;
  const StageDims d = make_dims(*s);
  const int64_t B = d.B, F = sizeof(float);
  int64_t sz[MSIG_NWS];
  for (int i = 0; i < MSIG_NWS; ++i) sz[i] = 0;
  sz[MSIG_WS_GATE_MEAN] = B * d.C * F;
  sz[MSIG_WS_GATE_PRE] = B * imax(d.Cr, 1) * F;
  sz[MSIG_WS_GATE_S] = B * d.C * F;
  sz[MSIG_WS_Y1] = B * d.L1 * 16 * F;
  sz[MSIG_WS_BN1_PART] = (int64_t)MSIG_PERSIST_WG * 32 * F;
  sz[MSIG_WS_BN1_STAT] = 64 * F;
  sz[MSIG_WS_P1] = B * d.P1 * 16 * F;
  sz[MSIG_WS_Y2] = B * d.L2 * 32 * F;
  sz[MSIG_WS_BN2_PART] = (int64_t)MSIG_PERSIST_WG * 64 * F;
  sz[MSIG_WS_BN2_STAT] = 128 * F;
  sz[MSIG_WS_P2] = B * d.TP * 32 * F;
  sz[MSIG_WS_H0] = B * d.TP * 128 * F;
  sz[MSIG_WS_H1] = B * d.TP * 64 * F;
  sz[MSIG_WS_FEAT] = B * 128 * F;
  sz[MSIG_WS_HID] = B * 64 * F;
  sz[MSIG_WS_LOGITS] = B * d.K * F;
  sz[MSIG_WS_PROBS] = B * d.K * F;
  sz[MSIG_WS_PRED] = B * (int64_t)sizeof(int32_t);
  sz[MSIG_WS_LOSS] = 4 * F;
  if (d.NT < 192) sz[MSIG_WS_GI] = 2 * (int64_t)d.NT * d.TP * 4 * 3 * 64 * 4 * F;   // latency form of the GRU forward (2 directions of layer 0)
  if (training) {
    const int64_t unit = 4096 * F;     // one (tile, step): 4 waves x 4 gates x 64 lanes x float4
    sz[MSIG_WS_STASH0] = 2 * (int64_t)d.NT * d.TP * unit;
    sz[MSIG_WS_STASH1] = (int64_t)d.NT * d.TP * unit;
    sz[MSIG_WS_STASH1R] = (int64_t)d.NT * unit;
    sz[MSIG_WS_DLOGITS] = B * d.K * F;
    sz[MSIG_WS_DFEAT] = B * 128 * F;
    sz[MSIG_WS_DH0] = B * d.TP * 128 * F;
    sz[MSIG_WS_DX0] = 2 * B * d.TP * 32 * F;
    sz[MSIG_WS_DY2] = B * d.L2 * 32 * F;
    sz[MSIG_WS_POOLC1] = B * d.P1 * 4;              // bytes
    sz[MSIG_WS_POOLC2] = B * d.TP * 8;
    sz[MSIG_WS_G1W] = B * (B >= 256 ? 1 : 8) * 32 * (int64_t)((d.C * 7 + 15) / 16 * 16) * F;      // small batches: up to 8 segment records per window (conv1_bwd_segs)
    sz[MSIG_WS_GATE_EO] = B * d.C * 2 * F;
    sz[MSIG_WS_DP1] = B * d.P1 * 16 * F;
    sz[MSIG_WS_DS] = B * d.C * F;
    sz[MSIG_WS_BNB_PART] = (int64_t)MSIG_PERSIST_WG * 64 * F;
    sz[MSIG_WS_BNB_STAT] = 64 * F;
    sz[MSIG_WS_GRAD_PART] = part_offsets(d).total * F;      // one sub-region per producer, see ColsumPlan
  }
  int64_t o = 0;
  for (int i = 0; i < MSIG_NWS; ++i) { off[i] = o; o += (sz[i] + 255) / 256 * 256; }
  off[MSIG_NWS] = o;
  return 0;
}

extern "C" int64_t msig_workspace_bytes(const msig_shape* s, int training) {
  int64_t off[MSIG_NWS + 1];
  const int rc = msig_workspace_layout(s, training, off);
  return rc ? (int64_t)rc : off[MSIG_NWS];
}

struct Ctx {
  StageDims d;
  WsPtrs w;
  int64_t po[MSIG_NPARAM + 1];
};

static int make_ctx(const msig_batch* b, Ctx& c, bool need_grads) {
  if (!b) return MSIG_E_NULL;
  int rc = check_shape(&b->shape);
  if (rc) return rc;
  if (!b->x || !b->params || !b->bn_state || !b->bn_count || !b->ws) return MSIG_E_NULL;
  if (need_grads && !b->grads) return MSIG_E_NULL;
  if (((uintptr_t)b->x | (uintptr_t)b->params | (uintptr_t)b->ws | (uintptr_t)b->grads) & 15) return MSIG_E_ALIGN;
  if (b->dropout_thr < 0 || b->dropout_thr > 256) return MSIG_E_SHAPE;
  if ((uintptr_t)b->loss_acc & 7) return MSIG_E_ALIGN;
  if (b->gru_layers < 0 || b->gru_layers > 2) return MSIG_E_SHAPE;
  if ((rc = msig_check_forms(b))) return rc;
  c.d = make_dims(b->shape);
  rc = msig_workspace_layout(&b->shape, b->training, c.w.off);
  if (rc) return rc;
  if (b->ws_bytes < c.w.off[MSIG_NWS]) return MSIG_E_WORKSPACE;
  c.w.base = (char*)b->ws;
  return msig_param_layout(b->shape.C, b->shape.K, c.po);
}

extern "C" int msig_frontend_fwd(const msig_batch* b, void* stream) {
  Ctx c; int rc = make_ctx(b, c, false); if (rc) return rc;
  return launch_frontend_fwd(b, c.d, c.w, c.po, single_fold(b), (hipStream_t)stream);
}
extern "C" int msig_gru_fwd(const msig_batch* b, void* stream) {
  Ctx c; int rc = make_ctx(b, c, false); if (rc) return rc;
  return launch_gru_fwd(b, c.d, c.w, c.po, single_fold(b), (hipStream_t)stream);
}
extern "C" int msig_head_ce_fwd(const msig_batch* b, void* stream) {
  Ctx c; int rc = make_ctx(b, c, false); if (rc) return rc;
  return launch_head_fwd(b, c.d, c.w, c.po, single_fold(b), (hipStream_t)stream);
}
extern "C" int msig_head_ce_bwd(const msig_batch* b, const float* dlogits, void* stream) {
  Ctx c; int rc = make_ctx(b, c, true); if (rc) return rc;
  if (!b->training) return MSIG_E_SHAPE;
  ColsumPlan plan;
  if ((rc = launch_head_bwd(b, dlogits, c.d, c.w, c.po, plan, single_fold(b), (hipStream_t)stream))) return rc;
  return launch_colsum_plan(plan, single_fold(b), (hipStream_t)stream);
}
extern "C" int msig_gru_bwd(const msig_batch* b, void* stream) {
  Ctx c; int rc = make_ctx(b, c, true); if (rc) return rc;
  if (!b->training) return MSIG_E_SHAPE;
  ColsumPlan plan;
  if ((rc = launch_gru_bwd(b, c.d, c.w, c.po, plan, single_fold(b), (hipStream_t)stream))) return rc;
  return launch_colsum_plan(plan, single_fold(b), (hipStream_t)stream);
}
extern "C" int msig_frontend_bwd(const msig_batch* b, void* stream) {
  Ctx c; int rc = make_ctx(b, c, true); if (rc) return rc;
  if (!b->training) return MSIG_E_SHAPE;
  ColsumPlan plan;
  if ((rc = launch_frontend_bwd(b, c.d, c.w, c.po, plan, single_fold(b), (hipStream_t)stream))) return rc;
  return launch_colsum_plan(plan, single_fold(b), (hipStream_t)stream);
}

static int forward_fc(const msig_batch* b, const FoldCtx& fc, hipStream_t st, bool with_head = true) {
  Ctx c; int rc = make_ctx(b, c, false); if (rc) return rc;
  if ((rc = msig_check_call_forms(b, c.d.NT, fc))) return rc;      // nothing has been launched: no model state has changed
  if ((rc = launch_frontend_fwd(b, c.d, c.w, c.po, fc, st))) return rc;
  if ((rc = launch_gru_fwd(b, c.d, c.w, c.po, fc, st))) return rc;
  return with_head ? launch_head_fwd(b, c.d, c.w, c.po, fc, st) : 0;
}
extern "C" int msig_forward(const msig_batch* b, void* stream) {
  if (!b) return MSIG_E_NULL;
  return forward_fc(b, single_fold(b), (hipStream_t)stream);
}

extern "C" int msig_backward(const msig_batch* b, const float* dlogits, void* stream) {
  Ctx c; int rc = make_ctx(b, c, true); if (rc) return rc;
  if (!b->training) return MSIG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const FoldCtx fc = single_fold(b);
  ColsumPlan plan;      // every weight-gradient reduction of the pass, done by one launch at the end
  if ((rc = launch_head_bwd(b, dlogits, c.d, c.w, c.po, plan, fc, st))) return rc;
  if ((rc = launch_gru_bwd(b, c.d, c.w, c.po, plan, fc, st))) return rc;
  if ((rc = launch_frontend_bwd(b, c.d, c.w, c.po, plan, fc, st))) return rc;
  return launch_colsum_plan(plan, fc, st);
}

extern "C" int msig_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq) return MSIG_E_NULL;
  if (n < 0 || (n & 3) || step < 1) return MSIG_E_SHAPE;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return MSIG_E_ALIGN;
  return launch_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream);
}

// fc.lr_over_bc1 is filled in here from `lr` (single fold) or from m->lr (fold batch)
static int train_step_fc(const msig_batch* b, FoldCtx fc, const float* lrs, const int64_t* steps, float* exp_avg, float* exp_avg_sq, float beta1,
                         float beta2, float eps, float weight_decay, int64_t step, hipStream_t st) {
  if (!b || !b->labels) return MSIG_E_NULL;
  if (!b->training) return MSIG_E_SHAPE;
  if (!exp_avg || !exp_avg_sq) return MSIG_E_NULL;
  if (step < 1) return MSIG_E_SHAPE;
  if (steps)
    for (int i = 0; i < fc.n; ++i) if (steps[i] < 0) return MSIG_E_SHAPE;
  if (((uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return MSIG_E_ALIGN;
  int rc;
  Ctx c;
  if ((rc = make_ctx(b, c, true))) return rc;          // every argument check of the step before its first launch
  if ((rc = msig_check_call_forms(b, c.d.NT, fc))) return rc;
  fc.fused_step = 1;        // forward and backward forms resolve from this one descriptor: gru_fwd_ws may store the two-vector stash
  // few windows: the head's forward, CrossEntropy and backward are one launch (head.hip head_step_kernel), its loss sums ride in the last one
  const bool head_step = head_step_applies(b, c.d);
  if ((rc = forward_fc(b, fc, st, !head_step))) return rc;
  // backward, then ONE launch that reduces every weight-gradient partial and applies Adam to each reduced element
  // (plus the few gradients their kernels write in place): the arithmetic of msig_backward + msig_adam_step
  ColsumPlan plan;
  if ((rc = head_step ? launch_head_step(b, c.d, c.w, c.po, plan, fc, st) : launch_head_bwd(b, nullptr, c.d, c.w, c.po, plan, fc, st))) return rc;
  if ((rc = launch_gru_bwd(b, c.d, c.w, c.po, plan, fc, st))) return rc;
  if ((rc = launch_frontend_bwd(b, c.d, c.w, c.po, plan, fc, st))) return rc;
  const int in_place[6] = {MSIG_P_GATE_W1, MSIG_P_GATE_W2, MSIG_P_BN1_G, MSIG_P_BN1_B, MSIG_P_BN2_G, MSIG_P_BN2_B};
  for (int i = 0; i < 6; ++i) {
    const int t = in_place[i];
    if (!plan.add_in_place(b->grads + c.po[t], (int)(c.po[t + 1] - c.po[t]))) return MSIG_E_SHAPE;
  }
  for (int i = 0; i < fc.n; ++i) {         // bias corrections per fold: folds of a batch may be at different step counts (msig_multi.step)
    const double s_i = (double)((steps && steps[i] > 0) ? steps[i] : step);
    const double bc1 = 1.0 - pow((double)beta1, s_i), bc2 = 1.0 - pow((double)beta2, s_i);
    fc.lr_over_bc1[i] = (float)((double)lrs[i] / bc1);
    fc.inv_sqrt_bc2[i] = (float)(1.0 / sqrt(bc2));
  }
  const AdamArgs ad{(float*)b->params, b->grads, exp_avg, exp_avg_sq, fc.lr_over_bc1[0], fc.inv_sqrt_bc2[0], beta1, beta2, eps, weight_decay};
  return launch_colsum_adam_plan(plan, ad, fc, st);
}

extern "C" int msig_train_step(const msig_batch* b, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                               float eps, float weight_decay, int64_t step, void* stream) {
  if (!b) return MSIG_E_NULL;
  return train_step_fc(b, single_fold(b), &lr, nullptr, exp_avg, exp_avg_sq, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream);
}

// ---- fold batching -------------------------------------------------------------------------------------------------
static int make_fold_ctx(const msig_batch* b, const msig_multi* m, FoldCtx& fc) {
  if (!b || !m) return MSIG_E_NULL;
  if (m->n < 1 || m->n > MSIG_MAX_FOLDS) return MSIG_E_SHAPE;
  if (m->stride_bytes <= 0 || (m->stride_bytes & 255)) return MSIG_E_ALIGN;
  if (m->form_folds < 0 || m->form_folds > MSIG_MAX_FOLDS) return MSIG_E_SHAPE;
  fc = FoldCtx{};
  fc.n = m->n; fc.stride = m->stride_bytes; fc.form_folds = m->form_folds ? m->form_folds : m->n;
  for (int i = 0; i < m->n; ++i) {
    if (m->slot[i] < 0) return MSIG_E_SHAPE;
    for (int j = 0; j < i; ++j) if (m->slot[j] == m->slot[i]) return MSIG_E_SHAPE;        // two launches into one arena would race
    fc.slot[i] = m->slot[i]; fc.key_gru[i] = m->key_gru[i]; fc.key_head[i] = m->key_head[i];
  }
  return 0;
}
extern "C" int msig_forward_multi(const msig_batch* b, const msig_multi* m, void* stream) {
  FoldCtx fc; int rc = make_fold_ctx(b, m, fc); if (rc) return rc;
  return forward_fc(b, fc, (hipStream_t)stream);
}
extern "C" int msig_train_step_multi(const msig_batch* b, const msig_multi* m, float* exp_avg, float* exp_avg_sq, float beta1, float beta2,
                                     float eps, float weight_decay, int64_t step, void* stream) {
  FoldCtx fc; int rc = make_fold_ctx(b, m, fc); if (rc) return rc;
  return train_step_fc(b, fc, m->lr, m->step, exp_avg, exp_avg_sq, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream);
}

extern "C" uint32_t msig_dropout_key(uint64_t seed, uint64_t step, uint32_t stream_id) {
  const uint32_t lo = (uint32_t)(seed & 0xFFFFFFFFu), hi = (uint32_t)(seed >> 32);
  const uint32_t a = (uint32_t)((step * 0x9E3779B9ull) & 0xFFFFFFFFull);
  const uint32_t b = (uint32_t)(((uint64_t)stream_id * 0x7F4A7C15ull) & 0xFFFFFFFFull);
  const uint32_t inner = fmix32(a + b + hi);
  return fmix32(lo ^ inner);
}

extern "C" int msig_gather_windows(const float* store, const int64_t* store_labels, const int64_t* idx, int32_t B,
                                   int64_t window_floats, float* out_x, int64_t* out_y, void* stream) {
  if (!store || !idx || !out_x) return MSIG_E_NULL;
  if (B < 1 || window_floats < 4 || (window_floats & 3)) return MSIG_E_SHAPE;
  if (((uintptr_t)store | (uintptr_t)out_x) & 15) return MSIG_E_ALIGN;
  return launch_gather(store, store_labels, idx, B, B, window_floats, out_x, out_y, single_fold(nullptr), (hipStream_t)stream);
}

extern "C" int msig_gather_windows_multi(const float* store, const int64_t* store_labels, const int64_t* idx, int64_t idx_row_stride, int32_t B,
                                         int64_t window_floats, float* out_x, int64_t* out_y, const msig_multi* m, void* stream) {
  if (!store || !idx || !out_x || !m) return MSIG_E_NULL;
  if (B < 1 || idx_row_stride < B || window_floats < 4 || (window_floats & 3)) return MSIG_E_SHAPE;
  if (((uintptr_t)store | (uintptr_t)out_x) & 15) return MSIG_E_ALIGN;
  msig_batch dummy{};
  FoldCtx fc; int rc = make_fold_ctx(&dummy, m, fc); if (rc) return rc;
  return launch_gather(store, store_labels, idx, idx_row_stride, B, window_floats, out_x, out_y, fc, (hipStream_t)stream);
}

int launch_channel_attention(const float* x, const float* W1, const float* W2, int B, int C, int T, float* out, float* s, float* scratch,
                             hipStream_t st);
extern "C" int msig_channel_attention(const float* x, const float* w1, const float* w2, int32_t B, int32_t C, int32_t T, float* out, float* s,
                                      float* scratch, void* stream) {
  if (!x || !out || !s || !scratch) return MSIG_E_NULL;
  if (B < 1 || C < 1 || C > MSIG_MAX_C || T < 1) return MSIG_E_SHAPE;
  if (C >= 4 && (!w1 || !w2)) return MSIG_E_NULL;
  return launch_channel_attention(x, w1, w2, B, C, T, out, s, scratch, (hipStream_t)stream);
}

extern "C" int64_t msig_normalise_scratch_bytes(void) { return (int64_t)(512 * 2 * MSIG_MAX_C + 2 * MSIG_MAX_C) * (int64_t)sizeof(double); }

extern "C" int msig_normalise_subject(const double* raw, int64_t N, int32_t T, int32_t C_all, const int32_t* cols, int32_t C,
                                      uint32_t log1p_mask, float* out, void* scratch, void* stream) {
  if (!raw || !cols || !out || !scratch) return MSIG_E_NULL;
  if (N < 1 || T < 1 || C_all < 1 || C < 1 || C > MSIG_MAX_C) return MSIG_E_SHAPE;
  for (int c = 0; c < C; ++c)
    if (cols[c] < 0 || cols[c] >= C_all) return MSIG_E_SHAPE;
  if (((uintptr_t)raw | (uintptr_t)scratch) & 7) return MSIG_E_ALIGN;
  return launch_normalise(raw, N, T, C_all, cols, C, log1p_mask, out, scratch, (hipStream_t)stream);
}
