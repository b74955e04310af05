"""The "raw" branch of the reference's ``preprocess.py`` on the GPU: whole-recording FFT
resampling (``resample_signal``, preprocess.py:70-75) and 60 s / 10 s sliding windows per protocol segment
(preprocess.py:184-200), written in the reference's ``{sid}_X.npy`` / ``{sid}_y.npy`` / ``_channel_names.txt``
format (preprocess.py:128-136,217-222).  Arithmetic is float64 like the reference's (numpy/scipy defaults) and runs
in libmsig_prep.so (hipFFT + two small kernels); there is no CPU fallback.  Reading WESAD's pickles (preprocess.py:60-68) stays
with the caller: `preprocess_recording` takes the decoded arrays; `parse_quest_csv` reads the protocol file (preprocess.py:41-58).

Out of scope, as in DESIGN.md §8: the hand-crafted feature branch (neurokit2) and `preprocess_check.py`.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Dict, Iterable, Sequence, Tuple

import numpy as np
import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libmsig_prep.so"
_lib = None

ORIGINAL_CHEST_FS = 700                       # preprocess.py:18
RAW_FS, RAW_WINDOW_SEC, RAW_STRIDE_SEC = 128, 60, 10          # preprocess.py:21-23
CHEST_CHANNELS = ["ACC", "ECG", "EDA", "EMG", "Resp", "Temp"]  # preprocess.py:27
TASK_TO_LABEL_MAP = {"Base": 1, "TSST": 2, "Fun": 3, "Medi1": 4, "Medi2": 4}   # preprocess.py:28
ALL_CHANNEL_NAMES = [f"chest_ACC_{ax}" for ax in "xyz"] + [f"chest_{c}" for c in ["ECG", "EDA", "EMG", "Resp", "Temp"]]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C multimodalsignal_amd/csrc`. There is no CPU fallback.")
        L = C.CDLL(str(LIB_PATH))
        vp, i64p, i32p = C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)
        L.msig_prep_resample.argtypes = [vp, C.c_int64, C.c_int32, C.c_int64, vp, vp]
        L.msig_prep_count_windows.argtypes = [i64p, i64p, C.c_int32, C.c_int64, C.c_int64]
        L.msig_prep_count_windows.restype = C.c_int64
        L.msig_prep_windows.argtypes = [vp, C.c_int64, C.c_int32, i64p, i64p, i32p, C.c_int32, C.c_int64, C.c_int64, vp, vp, vp]
        if L.msig_prep_abi_version() != 1:
            raise RuntimeError("libmsig_prep.so ABI version mismatch")
        _lib = L
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def resample_device(x: torch.Tensor, num: int) -> torch.Tensor:
    """scipy.signal.resample(x, num, axis=0) for a (n,) or (n, cols) float64 GPU tensor."""
    if not x.is_cuda or x.dtype != torch.float64 or x.dim() not in (1, 2):
        raise ValueError("resample_device needs a float64 GPU tensor of shape (n,) or (n, cols)")
    x2 = x.reshape(x.shape[0], -1).contiguous()
    y = torch.empty((int(num), x2.shape[1]), dtype=torch.float64, device=x.device)
    _check(lib().msig_prep_resample(x2.data_ptr(), x2.shape[0], x2.shape[1], int(num), y.data_ptr(), _stream(x.device)), "msig_prep_resample")
    return y.reshape((int(num),) + tuple(x.shape[1:]))


def resample_signal(signal_data: np.ndarray, original_fs, target_fs, device="cuda") -> np.ndarray:
    """preprocess.py:70-75, same signature and return type (numpy), computed on the GPU."""
    sig = np.asarray(signal_data, dtype=np.float64)
    num = int(len(sig) * (target_fs / original_fs))
    return resample_device(torch.from_numpy(np.ascontiguousarray(sig)).to(device), num).cpu().numpy()


def parse_quest_csv(subject_id: str, wesad_root) -> list:
    """preprocess.py:41-58: the protocol rows (task, start_min, end_min) of WESAD's ``SX_quest.csv`` — ';'-separated, the rows that
    start with '# ORDER', '# START', '# END'; empty cells dropped — with the reference's special case: for S2 and S6 the Base
    segment starts at its midpoint (preprocess.py:54-58).  Plain Python (the reference uses pandas); returns a list of tuples,
    which is what preprocess_recording takes."""
    path = Path(wesad_root) / subject_id / f"{subject_id}_quest.csv"
    rows = {}
    for line in path.read_text().splitlines():
        cells = line.split(";")
        for key in ("# ORDER", "# START", "# END"):
            if key in cells[0] and key not in rows:
                rows[key] = [c for c in cells[1:] if c.strip() != ""]
    tasks = list(rows["# ORDER"])
    starts, ends = [float(v) for v in rows["# START"]], [float(v) for v in rows["# END"]]
    if not (len(tasks) == len(starts) == len(ends)):
        raise ValueError(f"为受试者 {subject_id} 解析出的任务、开始、结束时间长度不匹配!")
    if subject_id in ("S2", "S6") and "Base" in tasks:
        i = tasks.index("Base")
        starts[i] = (starts[i] + ends[i]) / 2
    return list(zip(tasks, starts, ends))


def segment_bounds(start_min: float, end_min: float, original_fs=ORIGINAL_CHEST_FS, target_fs=RAW_FS) -> Tuple[int, int]:
    so, eo = int(start_min * 60 * original_fs), int(end_min * 60 * original_fs)                 # preprocess.py:163-164
    return int(so * (target_fs / original_fs)), int(eo * (target_fs / original_fs))            # preprocess.py:184-185


def windows_device(y: torch.Tensor, segments: Sequence[Tuple[int, int, int]], win: int, stride: int):
    """segments: (start, end, label) in resampled samples -> (X (n, win, cols) float64, labels (n,) int64), on the GPU."""
    if not y.is_cuda or y.dtype != torch.float64 or y.dim() != 2:
        raise ValueError("windows_device needs a (num, cols) float64 GPU tensor")
    y = y.contiguous()
    n = len(segments)
    s = (C.c_int64 * max(n, 1))(*[int(a) for a, _, _ in segments])
    e = (C.c_int64 * max(n, 1))(*[int(b) for _, b, _ in segments])
    lab = (C.c_int32 * max(n, 1))(*[int(c) for _, _, c in segments])
    nw = lib().msig_prep_count_windows(s, e, n, int(win), int(stride))
    if nw < 0:
        raise RuntimeError(f"msig_prep_count_windows failed with code {nw}")
    X = torch.empty((nw, int(win), y.shape[1]), dtype=torch.float64, device=y.device)
    L = torch.empty((nw,), dtype=torch.int64, device=y.device)
    if nw:
        _check(lib().msig_prep_windows(y.data_ptr(), y.shape[0], y.shape[1], s, e, lab, n, int(win), int(stride), X.data_ptr(), L.data_ptr(),
                                       _stream(y.device)), "msig_prep_windows")
    return X, L


def preprocess_recording(chest: Dict[str, np.ndarray], protocol: Iterable[Tuple[str, float, float]], device="cuda",
                         original_fs=ORIGINAL_CHEST_FS, target_fs=RAW_FS, window_sec=RAW_WINDOW_SEC, stride_sec=RAW_STRIDE_SEC):
    """One subject's raw branch (preprocess.py:141-154,158-164,184-200).  chest: {'ACC': (n,3), 'ECG': (n,1), ...} as in
    WESAD's pickle; protocol: (task, start_min, end_min) rows as parse_quest_csv returns them.  Returns X (N, T, 8)
    float64 and the raw protocol labels y (N,) as numpy arrays — exactly what the reference saves."""
    cols = [np.asarray(chest[c], dtype=np.float64).reshape(len(chest[c]), -1) for c in CHEST_CHANNELS]
    rec = torch.from_numpy(np.ascontiguousarray(np.concatenate(cols, axis=1))).to(device)      # (n, 8): ACC xyz, ECG, EDA, EMG, Resp, Temp
    num = int(rec.shape[0] * (target_fs / original_fs))
    y = resample_device(rec, num)
    segs = []
    for task, start_min, end_min in protocol:
        label = TASK_TO_LABEL_MAP.get(str(task).replace(" ", "").strip())
        if label is None:
            continue
        s, e = segment_bounds(start_min, end_min, original_fs, target_fs)
        segs.append((s, e, label))
    X, L = windows_device(y, segs, int(window_sec * target_fs), int(stride_sec * target_fs))
    return X.cpu().numpy(), L.cpu().numpy()


def save_subject(out_dir: Path, sid: str, X: np.ndarray, y: np.ndarray):
    """preprocess.py:128-136,217-222: the files WesadDataset reads."""
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    names = out_dir / "_channel_names.txt"
    if not names.exists():
        names.write_text("".join(f"{n}\n" for n in ALL_CHANNEL_NAMES))
    np.save(out_dir / f"{sid}_X.npy", X)
    np.save(out_dir / f"{sid}_y.npy", y)
