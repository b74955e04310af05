// libmsig_prep.so — offline preprocessing in front of the training path: FFT resampling of whole recordings
// (hipFFT, float64) and sliding-window extraction.  See include/msig_prep.h.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <stdint.h>
#include <vector>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/msig_prep.h"

extern "C" int msig_prep_abi_version(void) { return MSIG_PREP_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------------------
// Resampling = DFT of length n, spectrum surgery, inverse DFT of length num, with n and num whatever the recording
// happens to be (n = 4 200 317, num = 768 057 ...).  Handing those lengths to hipFFT works, but every new length
// costs 1-4 s of plan creation (run-time kernel generation) for 14 ms of transforms.  So both DFTs are done as
// chirp-z (Bluestein) convolutions on POWER-OF-TWO complex FFTs: X[k] = w[k] * IFFT(FFT(x*w) . FFT(conj-chirp))[k],
// w[m] = exp(-i pi m^2 / n).  The power-of-two plans are cached for the life of the process (a whole WESAD run needs
// two or three sizes), and 288 GB of HBM make the zero-padded 2^23-point x 8-column work arrays (1 GB) a non-issue.
// m^2 is reduced mod 2n in 64-bit integers before it becomes an angle, so the chirp keeps full double precision.
// ------------------------------------------------------------------------------------------------------------
#include <map>
#include <mutex>
typedef hipfftDoubleComplex cplx;

__device__ __forceinline__ cplx chirp(int64_t m, int64_t n, double sign) {      // exp(sign * i * pi * m^2 / n)
  const int64_t r = (m * m) % (2 * n);                                           // m < 2^31: m*m < 2^62
  double sn, cs;
  sincospi((double)r / (double)n, &sn, &cs);
  return cplx{cs, sign * sn};
}
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return cplx{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// b[m] = exp(+i pi m^2 / n) for |m| < n (wrapped into [0, M)), 0 elsewhere
__global__ __launch_bounds__(256) void chirp_filter_kernel(cplx* __restrict__ b, int64_t n, int64_t M) {
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    const int64_t d = m < n ? m : (M - m < n ? M - m : -1);
    b[m] = d >= 0 ? chirp(d, n, +1.0) : cplx{0.0, 0.0};
  }
}
// a[c][m] = x[m][c] * exp(-i pi m^2 / n) for m < n, 0 for n <= m < M
__global__ __launch_bounds__(256) void fwd_pack_kernel(const double* __restrict__ x, cplx* __restrict__ a, int64_t n, int64_t M, int ncols) {
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    cplx w = cplx{0.0, 0.0};
    if (m < n) w = chirp(m, n, -1.0);
    for (int c = 0; c < ncols; ++c) {
      const double v = m < n ? x[m * ncols + c] : 0.0;
      a[(int64_t)c * M + m] = cplx{v * w.x, v * w.y};
    }
  }
}
__global__ __launch_bounds__(256) void cmul_kernel(cplx* __restrict__ A, const cplx* __restrict__ B, int64_t M, int ncols) {
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    const cplx bv = B[m];
    for (int c = 0; c < ncols; ++c) A[(int64_t)c * M + m] = cmul(A[(int64_t)c * M + m], bv);
  }
}
// Spectrum surgery + packing of the inverse transform.  conv[c][k] (k < keep) holds M1 * X[k] / w_n[k]; the output
// spectrum is Ysel[k] = X[k] * scale (Nyquist bin * nyq_scale) for k < keep, 0 above, mirrored to the full Hermitian
// spectrum of length L; the inverse DFT is Re(DFT(conj(Yfull))), so a2[c][k] = conj(Yfull[k]) * exp(-i pi k^2 / L).
__global__ __launch_bounds__(256) void spectrum_pack_kernel(const cplx* __restrict__ conv, cplx* __restrict__ a2, int64_t n, int64_t M1,
                                                            int64_t L, int64_t M2, int64_t keep, int64_t nyq_bin, double nyq_scale,
                                                            double scale, int ncols) {
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < M2; k += (int64_t)gridDim.x * 256) {
    const bool live = k < L;
    const int64_t src = live ? (k <= L / 2 ? k : L - k) : 0;           // bin of the half spectrum this entry mirrors
    const bool have = live && src < keep;
    cplx wn = cplx{0.0, 0.0}, wl = cplx{0.0, 0.0};
    if (have) { wn = chirp(src, n, -1.0); wl = chirp(k, L, -1.0); }
    const double sc = (src == nyq_bin ? scale * nyq_scale : scale) / (double)M1;
    for (int c = 0; c < ncols; ++c) {
      cplx v = cplx{0.0, 0.0};
      if (have) {
        cplx X = cmul(conv[(int64_t)c * M1 + src], wn);               // X[src] * M1
        X.x *= sc; X.y *= sc;
        // Yfull[k] = X for k <= L/2, conj(X) above; we need conj(Yfull[k])
        if (k <= L / 2) X.y = -X.y;
        v = cmul(X, wl);
      }
      a2[(int64_t)c * M2 + k] = v;
    }
  }
}
// y[j][c] = Re(exp(-i pi j^2 / L) * conv2[c][j]) / M2
__global__ __launch_bounds__(256) void out_kernel(const cplx* __restrict__ conv2, double* __restrict__ y, int64_t L, int64_t M2, int ncols) {
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < L; j += (int64_t)gridDim.x * 256) {
    const cplx w = chirp(j, L, -1.0);
    for (int c = 0; c < ncols; ++c) {
      const cplx v = conv2[(int64_t)c * M2 + j];
      y[j * ncols + c] = (w.x * v.x - w.y * v.y) / (double)M2;
    }
  }
}

static std::mutex g_plan_mu;
static std::map<std::pair<int64_t, int>, hipfftHandle> g_plans;      // (points, batch) -> Z2Z plan, kept for the process
static int get_plan(int64_t M, int batch, hipfftHandle* out) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  auto it = g_plans.find({M, batch});
  if (it != g_plans.end()) { *out = it->second; return 0; }
  hipfftHandle h = 0;
  int len[1] = {(int)M};
  const hipfftResult r = hipfftPlanMany(&h, 1, len, nullptr, 1, (int)M, nullptr, 1, (int)M, HIPFFT_Z2Z, batch);
  if (r != HIPFFT_SUCCESS) return MSIG_PREP_E_FFT - (int)r;
  g_plans[{M, batch}] = h;
  *out = h;
  return 0;
}
static int64_t pow2_at_least(int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return p; }
static int grid_for(int64_t n) { const int64_t g = (n + 255) / 256; return (int)(g < 8192 ? g : 8192); }

extern "C" int msig_prep_resample(const double* x, int64_t n, int32_t ncols, int64_t num, double* y, void* stream) {
  if (!x || !y) return MSIG_PREP_E_NULL;
  if (n < 2 || num < 2 || ncols < 1 || n > (1ll << 29) || num > (1ll << 29)) return MSIG_PREP_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t N = n < num ? n : num, keep = N / 2 + 1;
  const bool even = (N % 2) == 0;
  const double nyq_scale = !even ? 1.0 : (num < n ? 2.0 : (n < num ? 0.5 : 1.0));
  const int64_t M1 = pow2_at_least(2 * n - 1), M2 = pow2_at_least(2 * num - 1);
  const bool timing = getenv("MSIG_PREP_TIMING") != nullptr;      // diagnostic: plans vs transforms
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  hipfftHandle p1 = 0, p1b = 0, p2 = 0, p2b = 0;
  int rc;
  if ((rc = get_plan(M1, ncols, &p1)) || (rc = get_plan(M1, 1, &p1b)) || (rc = get_plan(M2, ncols, &p2)) || (rc = get_plan(M2, 1, &p2b))) return rc;
  const double t_plan = now() - t0;
  cplx *a1 = nullptr, *b1 = nullptr, *a2 = nullptr, *b2 = nullptr;
  hipError_t e;
  if ((e = hipMalloc(&a1, sizeof(cplx) * M1 * ncols)) != hipSuccess) return (int)e;
  if ((e = hipMalloc(&b1, sizeof(cplx) * M1)) != hipSuccess) { (void)hipFree(a1); return (int)e; }
  if ((e = hipMalloc(&a2, sizeof(cplx) * M2 * ncols)) != hipSuccess) { (void)hipFree(a1); (void)hipFree(b1); return (int)e; }
  if ((e = hipMalloc(&b2, sizeof(cplx) * M2)) != hipSuccess) { (void)hipFree(a1); (void)hipFree(b1); (void)hipFree(a2); return (int)e; }
  rc = 0;
  {
    // the cached plans are shared: one resample at a time per process
    std::lock_guard<std::mutex> lk(g_plan_mu);
    hipfftResult r = HIPFFT_SUCCESS;
    auto F = [&](hipfftHandle h, cplx* buf, int dir) { if (rc == 0 && r == HIPFFT_SUCCESS) { r = hipfftSetStream(h, st); if (r == HIPFFT_SUCCESS) r = hipfftExecZ2Z(h, buf, buf, dir); } };
    chirp_filter_kernel<<<grid_for(M1), 256, 0, st>>>(b1, n, M1);
    fwd_pack_kernel<<<grid_for(M1), 256, 0, st>>>(x, a1, n, M1, ncols);
    F(p1b, b1, HIPFFT_FORWARD);
    F(p1, a1, HIPFFT_FORWARD);
    cmul_kernel<<<grid_for(M1), 256, 0, st>>>(a1, b1, M1, ncols);
    F(p1, a1, HIPFFT_BACKWARD);
    chirp_filter_kernel<<<grid_for(M2), 256, 0, st>>>(b2, num, M2);
    spectrum_pack_kernel<<<grid_for(M2), 256, 0, st>>>(a1, a2, n, M1, num, M2, keep, even ? N / 2 : -1, nyq_scale, 1.0 / (double)n, ncols);
    F(p2b, b2, HIPFFT_FORWARD);
    F(p2, a2, HIPFFT_FORWARD);
    cmul_kernel<<<grid_for(M2), 256, 0, st>>>(a2, b2, M2, ncols);
    F(p2, a2, HIPFFT_BACKWARD);
    out_kernel<<<grid_for(num), 256, 0, st>>>(a2, y, num, M2, ncols);
    if (r != HIPFFT_SUCCESS) rc = MSIG_PREP_E_FFT - (int)r;
    if (rc == 0 && (e = hipGetLastError()) != hipSuccess) rc = (int)e;
    e = hipStreamSynchronize(st);
    if (rc == 0 && e != hipSuccess) rc = (int)e;
  }
  (void)hipFree(a1); (void)hipFree(b1); (void)hipFree(a2); (void)hipFree(b2);
  if (timing) fprintf(stderr, "[msig_prep_resample] n=%lld -> num=%lld x %d columns (chirp-z on 2^%d / 2^%d points): %.3f s total, of which plan lookup/creation %.3f s\n",
                      (long long)n, (long long)num, ncols, __builtin_ctzll((unsigned long long)M1), __builtin_ctzll((unsigned long long)M2), now() - t0, t_plan);
  return rc;
}

static int64_t windows_in(int64_t s, int64_t e, int64_t win, int64_t stride) {
  const int64_t last = e - win + 1;            // range(s, e - win + 1, stride)
  return last > s ? (last - s + stride - 1) / stride : 0;
}

extern "C" int64_t msig_prep_count_windows(const int64_t* seg_start, const int64_t* seg_end, int32_t nseg, int64_t win, int64_t stride) {
  if (!seg_start || !seg_end) return MSIG_PREP_E_NULL;
  if (nseg < 0 || win < 1 || stride < 1) return MSIG_PREP_E_SHAPE;
  int64_t n = 0;
  for (int i = 0; i < nseg; ++i) n += windows_in(seg_start[i], seg_end[i], win, stride);
  return n;
}

// one workgroup per window: a window is a contiguous run of win*ncols doubles of the resampled recording
__global__ __launch_bounds__(256) void window_kernel(const double* __restrict__ y, const int64_t* __restrict__ first, const int64_t* __restrict__ label,
                                                     int64_t per_window, int ncols, double* __restrict__ out_x, int64_t* __restrict__ out_y) {
  const int64_t w = blockIdx.x;
  const double* src = y + first[w] * ncols;
  double* dst = out_x + w * per_window;
  for (int64_t i = threadIdx.x; i < per_window; i += 256) dst[i] = src[i];
  if (threadIdx.x == 0) out_y[w] = label[w];
}

extern "C" int msig_prep_windows(const double* y, int64_t num, int32_t ncols, const int64_t* seg_start, const int64_t* seg_end,
                                 const int32_t* seg_label, int32_t nseg, int64_t win, int64_t stride, double* out_x, int64_t* out_y,
                                 void* stream) {
  if (!y || !seg_start || !seg_end || !seg_label || !out_x || !out_y) return MSIG_PREP_E_NULL;
  if (nseg < 0 || win < 1 || stride < 1 || ncols < 1 || num < 1) return MSIG_PREP_E_SHAPE;
  std::vector<int64_t> first, label;
  for (int i = 0; i < nseg; ++i)
    for (int64_t s = seg_start[i]; s < seg_end[i] - win + 1; s += stride) {
      if (s < 0 || s + win > num) return MSIG_PREP_E_SHAPE;
      first.push_back(s); label.push_back(seg_label[i]);
    }
  const int64_t nw = (int64_t)first.size();
  if (nw == 0) return 0;
  if (nw > 0x7fffffff) return MSIG_PREP_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  int64_t* meta = nullptr;
  hipError_t e;
  if ((e = hipMalloc(&meta, sizeof(int64_t) * 2 * nw)) != hipSuccess) return (int)e;
  int rc = 0;
  if ((e = hipMemcpyAsync(meta, first.data(), sizeof(int64_t) * nw, hipMemcpyHostToDevice, st)) != hipSuccess) rc = (int)e;
  if (!rc && (e = hipMemcpyAsync(meta + nw, label.data(), sizeof(int64_t) * nw, hipMemcpyHostToDevice, st)) != hipSuccess) rc = (int)e;
  if (!rc) {
    window_kernel<<<(unsigned)nw, 256, 0, st>>>(y, meta, meta + nw, win * ncols, ncols, out_x, out_y);
    if ((e = hipGetLastError()) != hipSuccess) rc = (int)e;
  }
  e = hipStreamSynchronize(st);                  // the host vectors and `meta` must outlive the copies / the kernel
  if (!rc && e != hipSuccess) rc = (int)e;
  (void)hipFree(meta);
  return rc;
}
