"""Lane-level numpy model of csrc/stft_mfma_kernels.hip (test infrastructure, never imported by the product).

It executes the kernel's data flow with 64 "lanes" and an emulated v_mfma_f32_16x16x4_f32 on the constant block that
``ias_stft_build_mtables`` (the product's host-side C builder) produces, so the CPU suite pins the builder and the index
algebra of the kernel (operand lane maps, permuted k-order of the chained MFMAs, twiddles, Hermitian unpack, banded mel
tiles) against ``numpy.fft.rfft`` without a GPU.
"""
import ctypes

import numpy as np

LANES = np.arange(64)
LO, G = LANES & 15, LANES >> 4


def mfma16x16x4(a, b, c):
    """a, b: [64] operands (A[l&15][l>>4], B[l>>4][l&15]); c: [64,4] (col l&15, row 4 (l>>4) + reg) -> d [64,4]"""
    A = np.zeros((16, 4), np.float64)
    B = np.zeros((4, 16), np.float64)
    A[LO, G] = a
    B[G, LO] = b
    D = A @ B
    d = c.copy()
    for r in range(4):
        d[:, r] += D[4 * G + r, LO]
    return d


def layout(n_fft):
    N2 = n_fft // 2
    Q = N2 // 16
    L = dict(n_fft=n_fft, N2=N2, Q=Q, NT=Q // 16, NB=Q // 8, VPL=Q // 2, NPAIR_IT=N2 // 128)
    e = 0
    for name, n in (("win", L["VPL"]), ("b1", 8), ("tw1", L["NT"] * 8), ("a2", 4), ("tw2", (L["NB"] - 1) * 4),
                    ("unp", L["NPAIR_IT"] * 2)):
        L["e_" + name] = e
        e += n
    L["n_entries"] = e
    bins = N2 + 1
    L["pstr"] = ((bins - 8 + 63) // 64) * 64 + 8
    L["off_desc"] = 64 * e
    L["off_mela"] = L["off_desc"] + 4 * (4 + 2 * 60)
    return L


def build_mtables(lib, n_fft, window, mel=None):
    """mel: None or (start, count, woff, w, n_out) numpy arrays (int32 / float32)"""
    window = np.ascontiguousarray(window, np.float32)
    P = ctypes.c_void_p
    if mel is None:
        args = (None, None, None, None, 0)
    else:
        s, c, o, w, n_out = mel
        args = (P(s.ctypes.data), P(c.ctypes.data), P(o.ctypes.data), P(w.ctypes.data), int(n_out))
    n = lib.ias_stft_mtables_len(n_fft, args[0], args[1], args[4])
    assert n > 0, n
    out = np.zeros(n, np.float32)
    st = lib.ias_stft_build_mtables(n_fft, P(window.ctypes.data), args[0], args[1], args[2], args[3], args[4],
                                    P(out.ctypes.data))
    assert st == 0, st
    return out


def n2_of(Q, t, row):
    if Q == 16:
        return row
    h, e = t >> 1, t & 1
    return 2 * (row + 16 * h) + e


def frame_power(tab, n_fft, frame):
    """frame: [n_fft] float samples (un-windowed) -> power [n_fft/2 + 1], through the kernel's lane-level data flow"""
    L = layout(n_fft)
    Q, N2, NT, NB = L["Q"], L["N2"], L["NT"], L["NB"]
    E = lambda e: tab[64 * e:64 * e + 64].astype(np.float64)
    # loads, in load order, windowed
    xv = np.zeros((64, L["VPL"]))
    for v in range(L["VPL"]):
        if Q == 16:
            s, c = v >> 1, v & 1
            sample = 2 * Q * (4 * s + G) + 2 * LO + c
        else:
            nh = Q // 32
            s, h, e4 = v // (4 * nh), (v // 4) % nh, v & 3
            sample = 2 * Q * (4 * s + G) + 64 * h + 4 * LO + e4
        xv[:, v] = frame[sample] * E(L["e_win"] + v)

    def a_elem(s, t, c):    # A value of lane (i, kq) for tile t, k-step (c, s)
        if Q == 16:
            return xv[:, 2 * s + c]
        nh = Q // 32
        h, e = t >> 1, t & 1
        return xv[:, (s * nh + h) * 4 + 2 * e + c]

    cosb = [E(L["e_b1"] + s) for s in range(4)]
    sinb = [E(L["e_b1"] + 4 + s) for s in range(4)]
    acc1 = np.zeros((NT, 2, 64, 4))
    for t in range(NT):
        for s in range(4):
            acc1[t, 0] = mfma16x16x4(a_elem(s, t, 0), cosb[s], acc1[t, 0])
            acc1[t, 0] = mfma16x16x4(a_elem(s, t, 1), sinb[s], acc1[t, 0])
            acc1[t, 1] = mfma16x16x4(a_elem(s, t, 0), -sinb[s], acc1[t, 1])
            acc1[t, 1] = mfma16x16x4(a_elem(s, t, 1), cosb[s], acc1[t, 1])
    # twiddle 1
    sp = np.zeros((NT, 2, 64, 4))
    for t in range(NT):
        for r in range(4):
            c, s_ = E(L["e_tw1"] + 2 * (4 * t + r)), E(L["e_tw1"] + 2 * (4 * t + r) + 1)
            sr, si = acc1[t, 0][:, r], acc1[t, 1][:, r]
            sp[t, 0][:, r] = sr * c + si * s_
            sp[t, 1][:, r] = si * c - sr * s_
    # stage 2a
    a2 = [[E(L["e_a2"] + 2 * c + x) for x in range(2)] for c in range(2)]
    acc2 = np.zeros((NB, 64, 4))
    for b in range(NB):
        for c in range(2):
            for x in range(2):
                if Q == 16:
                    t, r = 0, 2 * x + b
                elif Q == 32:
                    t, r = b & 1, 2 * x + (b >> 1)
                else:
                    t, r = 2 * x + (b & 1), b >> 1
                acc2[b] = mfma16x16x4(a2[c][x], sp[t, c][:, r], acc2[b])
    # twiddle 2 + radix NB;  acc2[b][:, 2 kl + c']
    Z = np.zeros(N2, np.complex128)
    for kl in range(2):
        tv = []
        for b in range(NB):
            tr, ti = acc2[b][:, 2 * kl], acc2[b][:, 2 * kl + 1]
            if b > 0:
                c, s_ = E(L["e_tw2"] + 2 * (2 * (b - 1) + kl)), E(L["e_tw2"] + 2 * (2 * (b - 1) + kl) + 1)
                tr, ti = tr * c + ti * s_, ti * c - tr * s_
            tv.append(tr + 1j * ti)
        tv = np.stack(tv)                                   # [NB, 64]
        W = np.exp(-2j * np.pi * np.outer(np.arange(NB), np.arange(NB)) / NB)
        Y = W @ tv                                          # [kb, 64]
        for kb in range(NB):
            Z[LO + 16 * ((2 * G + kl) + 8 * kb)] = Y[kb]
    # unpack: a lane pairs its own lower-half bins k = k1 + 16 (2 G + kl) + 128 kb (kb < NB/2) with Z[N2 - k]
    P = np.zeros(N2 + 1)
    for kl in range(2):
        for kb in range(NB // 2):
            k = LO + 16 * (2 * G + kl) + 128 * kb
            zk, zn = Z[k], Z[(N2 - k) % N2]
            e = L["e_unp"] + 2 * (kl * (NB // 2) + kb)
            wr, wi = E(e), E(e + 1)
            a, b = zk.real + zn.real, zk.imag - zn.imag
            d, s_ = zk.real - zn.real, zk.imag + zn.imag
            tx = wr * s_ + wi * d
            ty = wi * s_ - wr * d
            P[k] = 0.25 * ((a + tx) ** 2 + (b + ty) ** 2)
            P[N2 - k] = 0.25 * ((a - tx) ** 2 + (b - ty) ** 2)
    P[N2 // 2] = abs(Z[N2 // 2]) ** 2
    return P


def mel_project(tab, n_fft, n_out, Pslots):
    """Pslots: [16, bins] power values of 16 frames -> [16, n_out] through the tiled descriptors / A table"""
    L = layout(n_fft)
    pstr = L["pstr"]
    sP = np.zeros((16, pstr))
    sP[:, :Pslots.shape[1]] = Pslots
    desc = tab[L["off_desc"]:L["off_mela"]].view(np.int32)
    A = tab[L["off_mela"]:].astype(np.float64)
    out = np.zeros((16, 16 * ((n_out + 15) // 16)))
    seen = set()
    DW = 4 + 2 * 60
    for w in range(4):
        dw = desc[w * DW:(w + 1) * DW]
        nblk, ablk = int(dw[0]), int(dw[1])
        acc = None
        firsts = []
        for b in range(nblk):
            poff, flags = int(dw[4 + 2 * b]), int(dw[5 + 2 * b])
            t = flags & 255
            if flags & 0x100:
                key = (t, (flags >> 10) & 15)
                assert key not in seen and acc is None
                seen.add(key)
                firsts.append(b)
                acc = np.zeros((64, 4))
            bq = np.stack([sP[LO, poff + 4 * G + s] for s in range(4)], 1)   # the 16-byte read of lane (g, j)
            for s in range(4):
                acc = mfma16x16x4(A[256 * (ablk + b) + 64 * s:256 * (ablk + b) + 64 * s + 64], bq[:, s], acc)
            if flags & 0x200:
                emit = ((flags >> (10 + G)) & 1).astype(bool)   # row quarter G (rows 4 G + r) belongs to the (sub)tile
                for r in range(4):
                    out[LO[emit], 16 * t + 4 * G[emit] + r] = acc[emit, r]
                acc = None
        assert acc is None
        # "next tile" links: the first tile in dw[2], then bits 16.. of every block of the tile before
        tiles_w = [int(dw[5 + 2 * b]) & 255 for b in firsts]
        assert int(dw[3]) == max(1, max(-(-int(desc[v * DW]) // 12) for v in range(4)))    # passes of 12 register blocks
        if tiles_w:
            assert int(dw[2]) == tiles_w[0]
            for i, b in enumerate(firsts):
                assert (int(dw[5 + 2 * b]) >> 16) & 255 == (tiles_w[i + 1] if i + 1 < len(tiles_w) else 255)
    return out[:, :n_out], seen


# ---------------------------------------------------------------------------------------------------------------------
# segment-major mel projection of csrc/stft2_kernels.hip (ias_stft_build_segtab)
def build_segtab(lib, n_fft, mel):
    s, c, o, w, n_out = mel
    P = ctypes.c_void_p
    n = lib.ias_stft_segtab_len(n_fft, P(s.ctypes.data), P(c.ctypes.data), P(o.ctypes.data), P(w.ctypes.data), int(n_out))
    if n < 0:
        return None
    out = np.zeros(n, np.float32)
    st = lib.ias_stft_build_segtab(n_fft, P(s.ctypes.data), P(c.ctypes.data), P(o.ctypes.data), P(w.ctypes.data),
                                   int(n_out), P(out.ctypes.data))
    assert st == 0, st
    return out


SEG_STRIDE = 72          # csrc/stft2_kernels.hip: IAS_SEG_STRIDE (floats per row of the segment-major power buffer)


def seg_mel(tab, Pbins, n_out):
    """Pbins [513] -> mel [n_out] through the lanes' store offsets, the row-wise gather and the U/D shift"""
    hdr = tab[:16].view(np.int32)
    assert hdr[0] == 0x5e67ab
    rows, rA, rB, rC, s0, nseg = (int(v) for v in hdr[1:7])
    addr = tab[16:16 + 9 * 64].view(np.int32).reshape(9, 64)
    wt = tab[16 + 9 * 64:].astype(np.float64).reshape(rows, 64, 2)
    buf = np.full(17 * SEG_STRIDE + 64, 1e30)             # stale scratch: must only ever meet zero weights
    seen = set()
    for e in range(4):
        k = (LANES >> 3) + 8 * (LANES & 7) + 64 * e
        buf[addr[e]] = Pbins[k]
        buf[addr[4 + e]] = Pbins[512 - k]
        seen |= set(k.tolist()) | set((512 - k).tolist())
    buf[addr[8][0]] = Pbins[256]
    seen.add(256)
    assert seen == set(range(513))
    U = np.zeros((3, 64)); D = np.zeros((3, 64))
    base = [0, rA, rA + rB]
    for g, rg in enumerate((rA, rB, rC)):
        for t in range(rg):
            r = base[g] + t
            w = wt[r]
            pv = buf[r * SEG_STRIDE + LANES]
            pv = np.where((w[:, 0] == 0) & (w[:, 1] == 0), 0.0, pv)      # 0 x stale = 0 in the kernel too (finite stale)
            U[g] += w[:, 0] * pv
            D[g] += w[:, 1] * pv
    Uf, Df = U.reshape(-1), D.reshape(-1)                  # slot i = 64 g + lane  <->  segment s0 + i
    mel = np.zeros(n_out)
    for m in range(n_out):
        iu, idn = m - s0, m + 1 - s0
        if 0 <= iu < nseg:
            mel[m] += Uf[iu]
        if 0 <= idn < nseg:
            mel[m] += Df[idn]
    return mel
