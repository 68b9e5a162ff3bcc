"""The identity behind vag_ic_photon_kernel's electron-energy walk (vegasafterglow_amd/csrc/vag_ic_kernels.h).

The reference (accumulate_IC, src/radiation/inverse-compton.h:483-527) forms, for every electron energy i, the scattering CDF
over the seed bins, c_j(i) = sum_{m >= j} ex_m(i), and lets output node kk take dNe_i * (c_{j+1}(i) + term_j(i)) at
j = n_lo - 2 i + kk -- c_0(i) alone for j < 0, nothing past the last bin.  The kernel instead adds dNe_i ex_m(i) and
dNe_i term_m(i) to two histograms over the diagonals d = m + 2 i and takes ONE suffix sum per cell:

    I[kk] = sum_{d > n_lo + kk} D[d] + E[n_lo + kk].

This file checks that rearrangement on random inputs in plain numpy (no GPU, no library): it is host-side arithmetic only.
"""
import numpy as np
import pytest


def reference_order(dNe, ex, term, n_lo, n_ic):
    g, nb = ex.shape  # electron energies, seed bins (nu_last)
    out = np.zeros(n_ic)
    for i in range(g):
        if not dNe[i] > 0:
            continue
        cdf = np.concatenate([np.cumsum(ex[i][::-1])[::-1], [0.0]])  # cdf[j] = sum_{m >= j} ex[m], cdf[nb] = 0
        if not cdf[0] > 0:
            continue
        for kk in range(n_ic):
            j = n_lo - 2 * i + kk
            if j < 0:
                out[kk] += dNe[i] * cdf[0]
            elif j < nb:
                out[kk] += dNe[i] * (cdf[j + 1] + term[i, j])
    return out


def diagonal_order(dNe, ex, term, n_lo, n_ic):
    g, nb = ex.shape
    nd = nb + 2 * g
    D, E = np.zeros(nd + 1), np.zeros(nd + 1)
    for i in range(g):
        if not dNe[i] > 0:
            continue
        for m in range(nb):
            D[m + 2 * i] += dNe[i] * ex[i, m]
            E[m + 2 * i] += dNe[i] * term[i, m]
    suffix = np.concatenate([np.cumsum(D[::-1])[::-1], [0.0]])  # suffix[d] = sum_{d' >= d} D[d']
    out = np.zeros(n_ic)
    for kk in range(n_ic):
        d0 = n_lo + kk
        above = suffix[0] if d0 + 1 <= 0 else (suffix[d0 + 1] if d0 + 1 <= nd else 0.0)
        at = E[d0] if 0 <= d0 <= nd else 0.0
        out[kk] = above + at
    return out


@pytest.mark.parametrize("g,nb,n_lo,n_ic", [(34, 49, -12, 70), (45, 76, -30, 83), (2, 1, 0, 2), (20, 63, 5, 192), (64, 127, -100, 192)])
def test_diagonal_histograms_reproduce_the_per_energy_cdf_sums(g, nb, n_lo, n_ic):
    rng = np.random.default_rng(g * 1000 + nb)
    dNe = np.exp(rng.uniform(-30, 5, g))
    dNe[rng.random(g) < 0.15] = 0.0                      # energies without electrons are skipped
    ex = np.exp(rng.uniform(-40, 3, (g, nb)))            # bin integrals: positive, many decades apart
    ex[:, rng.random(nb) < 0.1] = 0.0                    # empty seed bins
    term = ex * rng.uniform(0.5, 1.5, (g, nb))           # edge terms: the bin integral above the KN split, trapezoid * ratio below
    a = reference_order(dNe, ex, term, n_lo, n_ic)
    b = diagonal_order(dNe, ex, term, n_lo, n_ic)
    scale = np.maximum(np.abs(a), 1e-300)
    assert np.all(np.abs(a - b) <= 1e-12 * scale)        # all terms are positive: the two orders differ by rounding only
    assert np.any(a > 0)


def test_an_energy_whose_cdf_vanishes_contributes_nothing_in_either_order():
    # the reference skips an energy with cdf[0] <= 0; every ex of it is then 0, and so is every term (a trapezoid of zeros)
    g, nb = 5, 9
    dNe = np.ones(g)
    ex = np.ones((g, nb))
    term = np.ones((g, nb))
    ex[2], term[2] = 0.0, 0.0
    assert np.allclose(reference_order(dNe, ex, term, -3, 20), diagonal_order(dNe, ex, term, -3, 20), rtol=1e-14, atol=0)
