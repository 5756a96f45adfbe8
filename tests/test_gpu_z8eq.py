"""-m gpu: the one result the reference repository itself holds for this path -- `out/pca.csv`, the output of `example.jl:17-24`
(read_file -> preprocess -> sclens on data/Z8eq.csv.gz, Float32 GPU path): 3960 cells after QC, 9 robust signals, whose
eigenvalues are the squared column norms of that file (tests/golden/z8eq_expected.json, made by tests/golden/
make_z8eq_expected.py). The INPUT is not part of the reference tree (`.MISSING_LARGE_BLOBS:27`), so this test is skipped unless
the dataset is supplied: tests/golden/Z8eq.csv.gz, or the path in $SCLENS_Z8EQ (cells x genes, a leading `cell` column)."""
import json
import os

import numpy as np
import pytest

from sclens_amd import api

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _dataset():
    for p in (os.environ.get("SCLENS_Z8EQ", ""), os.path.join(HERE, "golden", "Z8eq.csv.gz")):
        if p and os.path.exists(p):
            return p
    return None


def test_z8eq_example_reproduces_the_reference_output(ctx):
    path = _dataset()
    if path is None:
        pytest.skip("data/Z8eq.csv.gz is not in the reference tree (.MISSING_LARGE_BLOBS:27); set SCLENS_Z8EQ to run")
    import pandas as pd

    want = json.load(open(os.path.join(HERE, "golden", "z8eq_expected.json")))
    df = pd.read_csv(path)  # scLENS.read_file (scLENS.jl:30-88): first column = cell ids, the others = genes
    genes = np.asarray(df.columns[1:])
    counts = df.iloc[:, 1:].to_numpy(dtype=np.float32)
    out = api.preprocess(counts, genes, cell_names=df.iloc[:, 0].to_numpy(), ctx=ctx)  # scLENS.preprocess defaults (:160-162)
    assert out is not None
    Xf, _, cells = out
    assert Xf.shape[0] == len(cells) == want["cells_after_qc"]
    res = api.sclens(Xf, seed=1, ctx=ctx)  # scLENS.sclens defaults: th = 60, p_step = 0.001, n_perturb = 20
    ev = np.asarray(res["signal_ev"])[np.asarray(res["sig_id"])]
    ref = np.asarray(want["robust_signal_eigenvalues"])
    assert res["pca_n1"].shape == (want["cells_after_qc"], len(ev))
    assert len(ev) == want["robust_signals"], (len(ev), ev)
    assert np.allclose(ev, ref, rtol=1e-2), (ev, ref)
    assert np.allclose((res["pca_n1"].astype(np.float64) ** 2).sum(axis=0), ev, rtol=1e-4)  # the identity the fixture rests on
