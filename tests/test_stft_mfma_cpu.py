"""Host logic of the matrix-core STFT kernel: the product's constant-block builder (``ias_stft_build_mtables``) run
through a lane-level numpy model of the kernel's data flow (tests/stft_mfma_model.py) against numpy.fft — the index
algebra of the chained MFMAs is pinned without a GPU."""
import numpy as np
import pytest
import torch

import stft_mfma_model as M
from inverse_audio_synthesis_amd import _lib
from inverse_audio_synthesis_amd.spectral import melscale_fbanks


def _csr(fb):
    start, count, woff, w = [], [], [], []
    for m in range(fb.shape[1]):
        nz = np.nonzero(fb[:, m])[0]
        if len(nz) == 0:
            start.append(0); count.append(0); woff.append(len(w))
            continue
        s, e = int(nz[0]), int(nz[-1]) + 1
        start.append(s); count.append(e - s); woff.append(len(w))
        w.extend(fb[s:e, m].tolist())
    return (np.array(start, np.int32), np.array(count, np.int32), np.array(woff, np.int32),
            np.array(w if w else [0.0], np.float32), fb.shape[1])


@pytest.mark.parametrize("n_fft", [512, 1024, 2048])
def test_lane_model_on_the_built_tables_matches_rfft(n_fft):
    lib = _lib.load()
    rng = np.random.default_rng(n_fft)
    win = torch.hann_window(n_fft).numpy()
    tab = M.build_mtables(lib, n_fft, win)
    for trial in range(2):
        x = rng.standard_normal(n_fft)
        if trial == 1:
            x = np.cos(2 * np.pi * 37.25 * np.arange(n_fft) / n_fft) + 0.3     # leakage + DC
        P = M.frame_power(tab, n_fft, x)
        ref = np.abs(np.fft.rfft(x * win)) ** 2
        assert np.abs(P - ref).max() <= 2e-6 * ref.max()


@pytest.mark.parametrize("n_fft,n_mels,sr", [(1024, 128, 44100), (512, 40, 16000), (2048, 100, 44100), (1024, 23, 16000)])
def test_mel_tiles_cover_the_filterbank(n_fft, n_mels, sr):
    lib = _lib.load()
    fb = melscale_fbanks(n_fft // 2 + 1, 0.0, sr / 2, n_mels, sr).numpy()
    tab = M.build_mtables(lib, n_fft, torch.hann_window(n_fft).numpy(), _csr(fb))
    Ps = np.random.default_rng(1).random((16, n_fft // 2 + 1))
    out, seen = M.mel_project(tab, n_fft, n_mels, Ps)
    ref = Ps @ fb
    assert np.abs(out - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1.0)
    # every tile that has a non-zero filter is in the waves' lists, whole (row-quarter mask 15) or in disjoint pieces
    live = {t for t in range((n_mels + 15) // 16) if np.any(fb[:, 16 * t:16 * t + 16])}
    assert {t for t, _ in seen} == live
    for t in live:
        masks = [m for tt, m in seen if tt == t]
        assert sum(masks) == np.bitwise_or.reduce(masks) <= 15          # disjoint


def test_unsupported_sizes_are_refused():
    lib = _lib.load()
    assert lib.ias_stft_mtables_len(768, None, None, 0) == -2


@pytest.mark.parametrize("n_mels,sr", [(128, 44100), (128, 16000), (80, 22050), (40, 16000), (23, 16000)])
def test_segment_major_mel_tables(n_mels, sr):
    """ias_stft_build_segtab: every bin is stored exactly once, stale scratch only meets zero weights, and the row-wise
    gather + U/D shift reproduces P @ fb; filterbanks whose segments do not fit report 'unsupported' (-> CSR path)."""
    lib = _lib.load()
    fb = melscale_fbanks(513, 0.0, sr / 2, n_mels, sr).numpy()
    tab = M.build_segtab(lib, 1024, _csr(fb))
    if tab is None:
        assert n_mels < 128       # wide filters: segments longer than the 17 rows of the scratch -> CSR path
        return
    rng = np.random.default_rng(3)
    for _ in range(3):
        Pb = rng.random(513)
        got = M.seg_mel(tab, Pb, n_mels)
        ref = Pb @ fb.astype(np.float64)
        assert np.abs(got - ref).max() <= 1e-12 * max(ref.max(), 1.0)


def test_segment_tables_refuse_non_triangular_filterbanks():
    lib = _lib.load()
    fb = np.zeros((513, 8), np.float32)
    fb[10:40, 0] = 1.0
    fb[10:40, 2] = 1.0            # two non-adjacent filters on the same bins
    assert M.build_segtab(lib, 1024, _csr(fb)) is None
