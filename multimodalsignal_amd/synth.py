"""Synthetic WESAD-shaped data (the real dataset is not redistributable and is absent from
the reference tree).  Writes the on-disk format preprocess.py:217-222
produces — ``{sid}_X.npy`` (N,T,C_all) float64, ``{sid}_y.npy`` raw protocol labels
{1,2,3,4}, ``_channel_names.txt`` — with a class-dependent planted signal so that
training has something to learn."""
from pathlib import Path

import numpy as np

CHANNELS6 = ["chest_ECG", "chest_EDA", "chest_Resp", "chest_EMG", "wrist_BVP", "wrist_EDA"]
ALL_SUBJECTS = [f"S{i}" for i in range(2, 18) if i != 12]          # main.py:67


def subject_window_counts(n_subjects, windows_per_subject, spread, seed=42):
    """Per-subject window counts: `windows_per_subject` +- `spread` (WESAD recordings differ in length by a few minutes, so the
    reference's per-subject `.npy` files differ by a few windows, dataset.py:17-27); spread 0 = equal sizes."""
    if not spread:
        return [int(windows_per_subject)] * n_subjects
    rs = np.random.RandomState(seed + 7919)
    return [int(windows_per_subject + v) for v in rs.randint(-spread, spread + 1, size=n_subjects)]


def make_synthetic_wesad(out_dir, subjects=ALL_SUBJECTS, windows_per_subject=270, T=3840, channels=CHANNELS6,
                         seed=42, fs=64.0, difficulty=1.0, window_spread=0):
    """`difficulty` >= 1 shrinks the class-dependent effects by 1/difficulty while the per-window,
    label-independent variability stays, so classes overlap more (1.0 is almost separable).
    `window_spread` > 0: subject i gets windows_per_subject +- window_spread windows (subject_window_counts)."""
    out = Path(out_dir)
    out.mkdir(parents=True, exist_ok=True)
    rs = np.random.RandomState(seed)
    C = len(channels)
    t = np.arange(T) / fs
    a = 1.0 / float(difficulty)
    # protocol mix of WESAD: baseline ~20 min, TSST ~10, amusement ~6.5, meditation 2x7
    probs = np.array([20.0, 10.0, 6.5, 14.0])
    probs /= probs.sum()
    counts = subject_window_counts(len(subjects), windows_per_subject, window_spread, seed)
    for si, sid in enumerate(subjects):
        n = counts[si]
        y = rs.choice([1, 2, 3, 4], size=n, p=probs)
        gain = 0.7 + 0.6 * rs.rand(C)                  # subject-specific scale / offset (what the z-score removes)
        offs = rs.randn(C)
        x = rs.randn(n, T, C)
        stress = (y == 2).astype(np.float64)[:, None]
        hr = 1.1 + 0.30 * a * stress + 0.12 * rs.randn(n, 1)                  # heart-rate like rhythm (Hz), jittered per window
        x[:, :, 0] += 1.5 * np.sin(2 * np.pi * hr * t[None, :] + rs.rand(n, 1) * 6.28)
        if C > 1:
            level = 1.0 + 0.5 * a * stress + 0.35 * rs.randn(n, 1)           # tonic EDA level, positive
            x[:, :, 1] = np.abs(x[:, :, 1]) * 0.2 + np.abs(level) + 0.0015 * a * t[None, :] * stress
        if C > 2:
            br = 0.25 + 0.06 * a * stress + 0.04 * rs.randn(n, 1)
            x[:, :, 2] += np.sin(2 * np.pi * br * t[None, :])
        for c in range(3, C):
            x[:, :, c] += 0.3 * a * stress * np.sin(2 * np.pi * (0.5 + 0.1 * c) * t[None, :])
        x = x * gain[None, None, :] + offs[None, None, :]
        if C > 1:
            x[:, :, 1] = np.abs(x[:, :, 1]) + 0.05
        np.save(out / f"{sid}_X.npy", x.astype(np.float64))
        np.save(out / f"{sid}_y.npy", y.astype(np.int64))
    (out / "_channel_names.txt").write_text("\n".join(channels) + "\n")
    return out
