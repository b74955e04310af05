/* msig_prep.h — C ABI of the OFFLINE preprocessing stage in front of the training path:
 * FFT resampling of a whole recording and sliding-window extraction, i.e. the "raw" branch of the reference's
 * preprocess.py.  Separate library (libmsig_prep.so) so that the training library has no FFT dependency.
 * Everything is float64, as in the reference (numpy/scipy defaults); all pointers are DEVICE pointers unless
 * marked host.  Return convention as msig.h: 0 ok, > 0 a hipError_t, < 0 an MSIG_PREP_E_* code
 * (hipFFT failures are reported as MSIG_PREP_E_FFT - hipfftResult).  Unlike libmsig_hip.so this stage may
 * allocate (hipFFT work areas) and synchronises the stream before it returns. */
#ifndef MSIG_PREP_H
#define MSIG_PREP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MSIG_PREP_ABI_VERSION 1
#define MSIG_PREP_E_NULL  (-1)
#define MSIG_PREP_E_SHAPE (-2)
#define MSIG_PREP_E_FFT   (-100)

int msig_prep_abi_version(void);

/* scipy.signal.resample(x, num) along axis 0 (scipy 1.15.3, the reference's pin; used by resample_signal,
 * preprocess.py:70-75): y = irfft(Y, num) * num/n with Y[0 : min(n,num)/2 + 1] = rfft(x)[same], the bin at
 * min(n,num)/2 doubled when decimating and halved when interpolating if min(n,num) is even, every other bin 0.
 * x: [n][ncols] row-major (a recording's channel columns, e.g. the 8 RespiBAN columns), y: [num][ncols]. */
int msig_prep_resample(const double* x, int64_t n, int32_t ncols, int64_t num, double* y, void* stream);

/* Number of windows the reference's loop `range(start, end - win + 1, stride)` emits over all segments
 * (preprocess.py:184-200).  Host pointers. */
int64_t msig_prep_count_windows(const int64_t* seg_start /* host */, const int64_t* seg_end /* host */, int32_t nseg,
                                int64_t win, int64_t stride);

/* The windows themselves: out_x [n_windows][win][ncols] (the `{sid}_X.npy` layout, preprocess.py:217-222) and
 * out_y [n_windows] = the segment's raw protocol label, in segment order then time order.  Windows that would
 * reach beyond `num` samples are a caller error (MSIG_PREP_E_SHAPE), as the reference would emit ragged arrays. */
int msig_prep_windows(const double* y, int64_t num, int32_t ncols, const int64_t* seg_start /* host */,
                      const int64_t* seg_end /* host */, const int32_t* seg_label /* host */, int32_t nseg, int64_t win,
                      int64_t stride, double* out_x, int64_t* out_y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSIG_PREP_H */
