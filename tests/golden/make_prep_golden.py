#!/usr/bin/env python3
"""Builds tests/golden/prep_ref.npz + prep_ref.json FROM THE REFERENCE's preprocess.py (build container only: imports
/root/reference; `neurokit2`, which the raw branch never calls, is replaced by an empty stand-in module exactly as `seaborn` is in
make_golden.py).  The fixtures are data: inputs and the reference's outputs.

    python tests/golden/make_prep_golden.py

What is recorded (VERDICT r4 item 6 / SURVEY §8f rank 4):
  * `resample_signal` (preprocess.py:70-75) on seeded signals: a 2-column 700 -> 128 Hz one, the same 700 -> 64 Hz, a 1-D one and a
    length that does not divide (int() truncation of the target length);
  * `parse_quest_csv` (preprocess.py:41-58) on a synthetic S2_quest.csv — the Base start moved to the segment's midpoint for S2 / S6 —
    and the window loop of `run_preprocessing` (preprocess.py:150-200, PROCESS_TARGETS = ['raw']) on a synthetic S2.pkl whose every
    channel is a sample-index ramp, with `resample_signal` replaced by an index-preserving stand-in of the reference's own target
    length: the saved windows then spell out, bit for bit, WHICH resampled samples each window holds (starts, lengths, channel
    order, labels).  The real resampler is pinned by the first item; the stand-in isolates the loop's integer arithmetic.
The reference module is imported with a scratch directory as the working directory: it creates ./data/... at import time.
"""
import json
import os
import pickle
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def main():
    sys.dont_write_bytecode = True       # nothing is written under /root/reference, not even a __pycache__ entry
    scratch = Path(tempfile.mkdtemp(prefix="msig_prep_golden_"))
    os.chdir(scratch)
    sys.path.insert(0, str(REF))
    sys.modules.setdefault("neurokit2", types.ModuleType("neurokit2"))
    import preprocess as rp          # noqa: E402  (creates ./data/chest_raw_align, ./data/chest_feature under the scratch dir)

    # ---- resample_signal ----
    rs = np.random.RandomState(20)
    sig2 = (rs.randn(2100, 2).cumsum(axis=0) * 0.05 + rs.randn(2100, 2)).astype(np.float64)      # 3 s of two RespiBAN channels
    sig1 = np.sin(np.arange(1751) / 37.0) + 0.1 * rs.randn(1751)                                 # 1-D, length not a multiple of 700
    arrays = {"sig2": sig2, "sig1": sig1,
              "sig2_to128": rp.resample_signal(sig2, 700, 128), "sig2_to64": rp.resample_signal(sig2, 700, 64),
              "sig1_to128": rp.resample_signal(sig1, 700, 128), "sig1_to64": rp.resample_signal(sig1, 700, 64)}

    # ---- parse_quest_csv + the window loop on a synthetic WESAD subject ----
    wesad = scratch / "WESAD"
    (wesad / "S2").mkdir(parents=True)
    n = 700 * 60 * 14 + 311                                   # 14 minutes of RespiBAN at 700 Hz
    ramp = np.arange(n, dtype=np.float64)
    chest = {b"ACC": np.stack([ramp, ramp + 0.25, ramp + 0.5], axis=1), b"ECG": ramp[:, None] + 1e6, b"EDA": ramp[:, None] + 2e6,
             b"EMG": ramp[:, None] + 3e6, b"Resp": ramp[:, None] + 4e6, b"Temp": ramp[:, None] + 5e6}
    with open(wesad / "S2" / "S2.pkl", "wb") as f:
        pickle.dump({b"signal": {b"chest": chest}, b"subject": b"S2"}, f)
    # the layout of WESAD's SX_quest.csv that parse_quest_csv reads: ';'-separated rows '# ORDER', '# START', '# END'
    order = ["Base", "TSST", "Medi 1", "Fun", "Medi 2", "sRead"]
    start = [0.40, 4.10, 7.33, 9.10, 11.30, 13.50]
    end = [3.60, 6.45, 8.90, 10.95, 13.26, 13.90]
    (wesad / "S2" / "S2_quest.csv").write_text("# Subj;S2;;;;;;\n# ORDER;" + ";".join(order) + ";\n# START;" + ";".join(f"{v:.2f}" for v in start) + ";\n# END;"
                                               + ";".join(f"{v:.2f}" for v in end) + ";\n")
    proto = rp.parse_quest_csv("S2", wesad)
    rp.WESAD_ROOT = wesad
    rp.PROCESS_TARGETS = ["raw"]
    rp.RAW_PATH = scratch / "out_raw"
    rp.RAW_PATH.mkdir()
    real_resample = rp.resample_signal

    def index_resample(signal_data, original_fs, target_fs):
        """the reference's target length (preprocess.py:72,74), values = index of the resampled sample + the channel's offset"""
        m = int(len(signal_data) * (target_fs / original_fs))
        idx = np.arange(m, dtype=np.float64)
        off = signal_data[0] if signal_data.ndim == 1 else signal_data[0:1, :]
        return idx + off if signal_data.ndim == 1 else idx[:, None] + off
    rp.resample_signal = index_resample
    rp.run_preprocessing()
    rp.resample_signal = real_resample
    X = np.load(rp.RAW_PATH / "S2_X.npy")
    y = np.load(rp.RAW_PATH / "S2_y.npy")
    names = (rp.RAW_PATH / "_channel_names.txt").read_text().split()
    assert X.shape[1:] == (rp.RAW_WINDOW_SEC * rp.RAW_FS, 8), X.shape
    # every window is a run of consecutive resampled samples in every column: record its first index and verify the rest here
    starts = X[:, 0, 0].astype(np.int64)
    offs = np.array([0.0, 0.25, 0.5, 1e6, 2e6, 3e6, 4e6, 5e6])
    want = starts[:, None, None] + np.arange(X.shape[1])[None, :, None] + offs[None, None, :]
    assert np.array_equal(X, want), "a window is not a run of consecutive samples in the reference's channel order"
    doc = {"what": "reference preprocess.py (17LiQi/MultimodalSignal) on synthetic inputs; generated by tests/golden/make_prep_golden.py",
           "n_samples_700hz": n, "raw_fs": rp.RAW_FS, "window_sec": rp.RAW_WINDOW_SEC, "stride_sec": rp.RAW_STRIDE_SEC,
           "quest": {"order": order, "start_min": start, "end_min": end},
           "protocol_after_parse": [[r["task"], float(r["start_min"]), float(r["end_min"])] for _, r in proto.iterrows()],
           "channel_names": names, "resampled_length": int(n * (rp.RAW_FS / 700)),
           "window_starts": starts.tolist(), "labels": y.astype(int).tolist(), "window_shape": list(X.shape[1:]),
           "column_offsets": offs.tolist()}
    np.savez_compressed(OUT / "prep_ref.npz", **arrays)
    (OUT / "prep_ref.json").write_text(json.dumps(doc, indent=1))
    print(f"{len(starts)} windows, labels {sorted(set(y.tolist()))}; protocol after parse: {doc['protocol_after_parse']}")
    print(f"resampled lengths: {[arrays[k].shape for k in ('sig2_to128', 'sig2_to64', 'sig1_to128', 'sig1_to64')]}")


if __name__ == "__main__":
    main()
