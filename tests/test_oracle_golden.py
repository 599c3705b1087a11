"""The oracle against the reference's own outputs (tests/golden, made by scripts/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import pqmf_oracle as po
from oracle import vicreg_oracle as vo
from helpers import randn, checks


def test_pqmf_filters(golden_dir):
    g = np.load(os.path.join(golden_dir, "pqmf_filters.npz"))
    for N in (3, 4, 64):
        H, G, U = po.design(N)
        assert np.array_equal(H.numpy(), g[f"H{N}"])
        assert np.array_equal(G.numpy(), g[f"G{N}"])
        assert np.array_equal(U.numpy(), g[f"updown{N}"])
    H, G, _ = po.design(8, 30, 0.07, 7.0)
    assert np.array_equal(H.numpy(), g["H8_t30"]) and np.array_equal(G.numpy(), g["G8_t30"])


def test_pqmf_analysis_small_and_edges(golden_dir):
    g = np.load(os.path.join(golden_dir, "pqmf_analysis.npz"))
    H3, _, _ = po.design(3)
    H4, _, _ = po.design(4)
    z = po.analysis(randn((4, 1, 16000), 101), H3, 3, 62)
    assert z.shape == (4, 3, 5334)
    np.testing.assert_allclose(z.numpy(), g["small_z3"], rtol=0, atol=2e-6)
    for T in (1, 31, 62, 63, 64, 1000, 1001):
        xe = randn((2, 1, T), 200 + T)
        np.testing.assert_allclose(po.analysis(xe, H4, 4, 62).numpy(), g[f"edge_T{T}_z4"], atol=2e-6)
        np.testing.assert_allclose(po.analysis(xe, H3, 3, 62).numpy(), g[f"edge_T{T}_z3"], atol=2e-6)


def test_pqmf_analysis_full_length(golden_dir):
    g = np.load(os.path.join(golden_dir, "pqmf_analysis.npz"))
    x = randn((2, 1, 176400), 102)
    for N, tag in ((3, "z3"), (64, "z64")):
        H, _, _ = po.design(N)
        z = po.analysis(x, H, N, 62)
        assert list(z.shape) == list(g[f"full_{tag}_shape"])
        np.testing.assert_allclose(z.flatten()[::97].numpy(), g[f"full_{tag}_sub"], atol=2e-6)
        np.testing.assert_allclose(checks(z), g[f"full_{tag}_checks"], rtol=1e-5)


def test_pqmf_synthesis(golden_dir):
    g = np.load(os.path.join(golden_dir, "pqmf_synthesis.npz"))
    x = randn((2, 1, 4096), 103)
    for N in (3, 4, 64):
        H, G, U = po.design(N)
        z = po.analysis(x, H, N, 62)
        np.testing.assert_allclose(z.numpy(), g[f"z{N}"], atol=2e-6)
        y = po.synthesis(torch.from_numpy(g[f"z{N}"]), G, U, N, 62)
        np.testing.assert_allclose(y.numpy(), g[f"y{N}"], atol=2e-5)


def test_preprocess(golden_dir):
    g = np.load(os.path.join(golden_dir, "audioembed_preprocess.npz"))
    H, _, _ = po.design(3)
    img = po.preprocess(randn((2, 1, 176400), 104), H, 3, 62, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    assert list(img.shape) == list(g["shape"])
    np.testing.assert_allclose(img.flatten()[::89].numpy(), g["sub"], atol=1e-5)
    np.testing.assert_allclose(checks(img), g["checks"], rtol=1e-5)


@pytest.mark.parametrize("tag", ["tiny", "b16", "b128", "denom_quirk", "b1024"])
def test_vicreg_loss(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "vicreg_loss.npz"))
    B, D, cfgB, s1, s2 = [int(v) for v in g[tag + "_meta"]]
    x, y = randn((B, D), s1), randn((B, D), s2) * 0.7 + 0.1
    if tag == "tiny":
        assert np.array_equal(x.numpy(), g["tiny_x"]) and np.array_equal(y.numpy(), g["tiny_y"])
    out = [o.item() for o in vo.loss(x, y, cfgB, D)]
    np.testing.assert_allclose(out, g[tag + "_out"], rtol=2e-6)


def test_off_diagonal(golden_dir):
    g = np.load(os.path.join(golden_dir, "vicreg_loss.npz"))
    assert np.array_equal(vo.off_diagonal(torch.from_numpy(g["offdiag_in"])).numpy(), g["offdiag_out"])
    assert np.array_equal(vo.off_diagonal(torch.from_numpy(g["offdiag7_in"])).numpy(), g["offdiag7_out"])
    assert g["offdiag_out"].tolist() == [1, 2, 3, 5, 6, 7]
