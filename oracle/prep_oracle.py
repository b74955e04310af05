"""CPU restatement of the reference's raw preprocessing branch — TEST INFRASTRUCTURE ONLY (imported by tests/
only; the product path is multimodalsignal_amd/preprocess.py -> libmsig_prep.so and has no CPU fallback).

Parity: PINNED.  `resample` restates scipy.signal.resample (scipy 1.15.3, the version the reference's
requirements.txt pins and the one installed here; real input, no window) and is checked against scipy's own
function in tests/test_prep.py; `windows` restates the loop of preprocess.py:184-200."""
import numpy as np


def resample(x: np.ndarray, num: int) -> np.ndarray:
    """scipy.signal.resample(x, num, axis=0) — preprocess.py:70-75 calls it per column."""
    x = np.asarray(x, dtype=np.float64)
    nx = x.shape[0]
    X = np.fft.rfft(x, axis=0)
    Y = np.zeros((num // 2 + 1,) + x.shape[1:], dtype=X.dtype)
    n = min(num, nx)
    nyq = n // 2 + 1
    Y[:nyq] = X[:nyq]
    if n % 2 == 0:
        if num < nx:
            Y[n // 2] *= 2.0
        elif nx < num:
            Y[n // 2] *= 0.5
    return np.fft.irfft(Y, num, axis=0) * (float(num) / float(nx))


def target_length(n: int, original_fs: float, target_fs: float) -> int:
    return int(n * (target_fs / original_fs))                 # preprocess.py:72,74


def segment_bounds(start_min: float, end_min: float, original_fs: float, target_fs: float):
    """preprocess.py:163-164,184-185: minutes -> original samples (int) -> resampled samples (int)."""
    so, eo = int(start_min * 60 * original_fs), int(end_min * 60 * original_fs)
    return int(so * (target_fs / original_fs)), int(eo * (target_fs / original_fs))


def windows(y: np.ndarray, segments, win: int, stride: int):
    """segments: [(start, end, label)] in resampled samples -> (X [n][win][cols], labels [n])."""
    xs, ls = [], []
    for s, e, lab in segments:
        for i in range(s, e - win + 1, stride):
            xs.append(y[i:i + win])
            ls.append(lab)
    if not xs:
        return np.zeros((0, win) + y.shape[1:]), np.zeros((0,), dtype=np.int64)
    return np.array(xs), np.array(ls, dtype=np.int64)
