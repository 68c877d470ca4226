#!/usr/bin/env python3
"""Statistics of the counter-based dropout function of pacednegatives_amd/csrc/common.h, restated in numpy.

drop_values(seed, idx) gives the 32-bit value of every element index; an element is kept when value >= thresh16 << 16.
Run as a script it reports, for random seeds and start offsets, at p = 0.1, 0.5 and 0.01:
  * the keep rate of each of the four elements of a quad, as a z score against the binomial,
  * the correlation of keep decisions between the elements of a quad and between quads 1 .. 65536 apart,
  * the dependence of the keep rate on every bit of the quad number,
  * the correlation between neighbouring seeds,
  * chi-square of the byte histograms of the 16-bit fields (255 degrees of freedom).
About 3500 z scores are taken, so the largest should be near 4 for a function without structure.
tests/test_kernels_gpu.py::test_dropout_mask_equals_the_numpy_restatement imports drop_values."""
import numpy as np

U = np.uint64
M32 = U(0xffffffff)
MULTIPLIERS = (0x1b873b, 0x6ca68b, 0xb2ae35, 0xd4e12d)


def _mad24(a, c, add):
    return ((a & U(0xffffff)) * U(c) + add) & M32


def drop_state(seed, quad):
    """state shared by the four elements of every quad number in `quad` (uint64 array)"""
    quad = np.asarray(quad, dtype=np.uint64)
    hi = (quad >> U(32)) & U(0xffffff)
    x = _mad24(hi, 0x9e3779, (quad & M32) ^ U(seed))
    x ^= x >> U(16); x = _mad24(x, 0xeb352d, x)
    x ^= x >> U(13); x = _mad24(x, 0x2c1b3d, x)
    x ^= x >> U(12)
    return x


def drop_values(seed, idx):
    """32-bit value of every element index in `idx`: kept iff value >= round(p * 65536) << 16"""
    idx = np.asarray(idx, dtype=np.uint64)
    y = drop_state(seed, idx >> U(2))
    mul = np.asarray(MULTIPLIERS, dtype=np.uint64)[(idx & U(3)).astype(np.int64)]
    return ((y & U(0xffffff)) * mul + y) & M32


def report(n=1 << 22, trials=6):
    rng = np.random.default_rng(1)
    worst_rate, worst_corr, worst = 0.0, 0.0, None
    for trial in range(trials):
        seed = int(rng.integers(0, 2 ** 32))
        base = U(int(rng.integers(0, 2 ** 31))) + (U(trial % 3) << U(32))
        q = np.arange(n, dtype=np.uint64) + base
        y = drop_state(seed, q)
        vals = [((y & U(0xffffff)) * U(c) + y) & M32 for c in MULTIPLIERS]
        for p in (0.1, 0.5, 0.01):
            limit = U(int(p * 65536 + 0.5)) << U(16)
            keep = [(v >= limit).astype(np.float64) - (1 - p) for v in vals]
            sig = np.sqrt(p * (1 - p) / n)
            for k in keep:
                worst_rate = max(worst_rate, abs(k.mean() / sig))
            pairs = [(f"e{i}e{j}", keep[i], keep[j]) for i in range(4) for j in range(i + 1, 4)]
            for d in (1, 2, 3, 4, 8, 16, 64, 65, 256, 1024, 4096, 65536):
                pairs += [(f"d{d}e{i}e{j}", keep[i][:-d], keep[j][d:]) for i in range(4) for j in range(4)]
            for name, a, b in pairs:
                z = np.mean(a * b) / (p * (1 - p)) * np.sqrt(len(a))
                if abs(z) > worst_corr:
                    worst_corr, worst = abs(z), (name, p, trial)
            for bit in range(22):
                sel = ((q >> U(bit)) & U(1)) == 1
                for i, k in enumerate(keep):
                    z = (k[sel].mean() - k[~sel].mean()) / (2 * sig)
                    if abs(z) > worst_corr:
                        worst_corr, worst = abs(z), (f"bit{bit}e{i}", p, trial)
    idx = np.arange(4 * n, dtype=np.uint64)
    seeds = 0.0
    for s in range(4):
        a = (drop_values(12345 + s, idx) >> U(16)) >= U(6554)
        b = (drop_values(12346 + s, idx) >> U(16)) >= U(6554)
        seeds = max(seeds, abs(np.mean((a - .9) * (b - .9)) / .09 * np.sqrt(len(idx))))
    f = drop_values(999, idx) >> U(16)
    chi = []
    for e in range(4):
        for sh in (0, 8):
            cnt = np.bincount(((f[e::4] >> U(sh)) & U(0xff)).astype(np.int64), minlength=256)
            chi.append(((cnt - n / 256) ** 2 / (n / 256)).sum())
    print(f"largest |z|: keep rate {worst_rate:.2f}, pair correlation / index bit {worst_corr:.2f} {worst}, "
          f"neighbouring seeds {seeds:.2f}; byte histograms chi2 (255 dof, sd 22.6) max {max(chi):.0f}")


if __name__ == "__main__":
    report()
