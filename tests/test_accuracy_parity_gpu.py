"""GPU tier, north_star's accuracy criterion: per-fold accuracy of the HIP path against the REFERENCE's
own CPU runs on identical windows, at the reference's defaults (B = 64, dropout 0.5, shuffled batches, Adam), over several
seeds per side — one seed cannot separate a few pp of per-fold difference from the spread of the random streams, which
differ by construction (torch's Philox dropout / DataLoader shuffling vs the counter-based masks and the device shuffler).

Fixture: tests/golden/loso_parity_ref.json (tests/golden/make_parity_fixture.py; the reference imported in the build
container).  The synthetic dataset is regenerated here from the generator arguments the fixture records."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1500)
def test_loso_accuracy_matches_reference_over_seeds(tmp_path_factory):
    sys.path.insert(0, str(ROOT / "tools"))
    import loso_parity as LP
    fx = json.loads((GOLDEN / "loso_parity_ref.json").read_text())
    ds, tr = fx["dataset"], fx["training"]
    data = Path("/tmp") / f"msig_parity_w{ds['windows_per_subject']}_t{ds['T']}_d{ds['difficulty']}"
    LP.ensure_data(data, ds["windows_per_subject"], ds["T"], ds["difficulty"])
    out = tmp_path_factory.mktemp("parity")
    folds = list(fx["runs"][0]["folds"])
    ref = {f: np.array([r["folds"][f]["acc"] for r in fx["runs"]]) for f in folds}
    got = {f: [] for f in folds}
    for run in fx["runs"]:
        res = LP.run_side("gpu", data, out, run["seed_base"], folds, tr["epochs"], tr["batch"], tr["dropout"], log=lambda *a: None)
        for r in res:
            got[r["subject"]].append(r["acc"])
    got = {f: np.array(v) for f, v in got.items()}
    n = len(fx["runs"])
    ref_mean, got_mean = float(np.mean([ref[f].mean() for f in folds])), float(np.mean([got[f].mean() for f in folds]))
    print(f"\nmean LOSO accuracy over {n} seeds x {len(folds)} folds: reference {ref_mean:.4f}  HIP {got_mean:.4f}  (diff {100 * (got_mean - ref_mean):+.2f} pp)")
    for f in folds:
        print(f"  {f}: reference {ref[f].mean():.3f} [{ref[f].min():.2f} .. {ref[f].max():.2f}]   HIP {got[f].mean():.3f} [{got[f].min():.2f} .. {got[f].max():.2f}]")
    # north_star: mean LOSO accuracy within +-0.5 pp of the reference's CPU run on identical windows
    assert abs(got_mean - ref_mean) <= 0.005, (got_mean, ref_mean)
    # and every fold's mean inside the interval the reference's own seeds span (widened by one test-window: 1 / n_test)
    slack = 1.0 / ds["windows_per_subject"]
    for f in folds:
        assert ref[f].min() - slack <= got[f].mean() <= ref[f].max() + slack, (f, got[f].mean(), ref[f].min(), ref[f].max())
