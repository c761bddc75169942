"""The chr22 study of BASELINE.json configs[2..4] as a host-side workload description.

Measured SNPs -- positions, alleles and z-scores -- come from the reference's own study file
(`data/PGC2_Chr22_ilmn1M_Z.txt`, 13 362 SNPs over 16.05-51.21 Mb, committed as input data under
tests/golden/); only what the reference does not ship is synthesised: the unmeasured panel SNPs (up to
~100 000 in total) and the genotypes (Balding-Nichols frequencies + Gaussian-copula
AR(1) LD, generated on the device by `gauss_synth_device`).  Windows follow the vignette's call pattern
(`docs/articles/dist_example.md:144-153`): 1 Mb prediction windows `start = k*10^6 + 1`, 500 kb wings,
membership as in `dist.cpp:132-140`, the ">10" guards of `dist.cpp:145-146`.

Real positions matter for measurement: the measured count of the extended windows spans 156-1 213, so
window cost varies ~10x -- the skew the farm's LPT sharding exists for (uniform positions hide it).
"""
import os

import numpy as np

from . import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHR22_STUDY = os.path.join(ROOT, "tests", "golden", "PGC2_Chr22_ilmn1M_Z.txt")
WINDOW_BP = 1_000_000


def read_study(path=CHR22_STUDY):
    """(rsid, bp, a1, a2, z) of the study file, sorted by position (stable: file order among equal bp)."""
    rsid, bp, a1, a2, z = [], [], [], [], []
    with open(path) as f:
        f.readline()
        for line in f:
            t = line.split()
            if len(t) >= 6:
                rsid.append(t[0]); bp.append(int(t[2])); a1.append(t[3]); a2.append(t[4]); z.append(float(t[5]))
    bp = np.array(bp, dtype=np.int64)
    o = np.argsort(bp, kind="stable")
    return (np.array(rsid)[o], bp[o], np.array(a1)[o], np.array(a2)[o], np.array(z, dtype=np.float64)[o])


def populations(mode):
    """Selected populations: the 21 PGC2 populations (distmix, N = 32 147) or the EUR super-population (dist, 20 281)."""
    if mode == "dist":
        return [p for p in synth.POPS_33KG if p[2] == "EUR"]
    return [p for p in synth.POPS_33KG if p[0] in synth.PGC2_WEIGHTS]


def make_chromosome(snps=100_000, mode="distmix", seed=20260216, sample_scale=1.0, study=CHR22_STUDY):
    """Host-side description of the chromosome: positions, measured mask, z, per-population thresholds of the
    genotype generator.  `snps` < 100 000 thins measured and unmeasured SNPs alike (debug sizes)."""
    from scipy.stats import norm
    rng = np.random.default_rng(seed)
    pops = populations(mode)
    if sample_scale != 1.0:
        pops = [(a, max(30, int(n * sample_scale)), s) for a, n, s in pops]
    w = np.array([synth.PGC2_WEIGHTS.get(p[0], 1.0) for p in pops])
    off = synth.pop_offsets([p[1] for p in pops])
    _, mbp, _, _, mz = read_study(study)
    if snps < 100_000:
        keep = np.sort(rng.choice(len(mbp), size=max(12, int(round(len(mbp) * snps / 100_000))), replace=False))
        mbp, mz = mbp[keep], mz[keep]
    lo, hi = int(mbp[0]), int(mbp[-1])
    n_un = max(0, snps - len(mbp))
    cand = np.setdiff1d(np.arange(lo, hi + 1, dtype=np.int64), mbp)
    ubp = np.sort(rng.choice(cand, size=n_un, replace=False))
    bp = np.concatenate([mbp, ubp])
    measured = np.concatenate([np.ones(len(mbp), bool), np.zeros(len(ubp), bool)])
    z = np.concatenate([mz, np.zeros(len(ubp))])
    o = np.argsort(bp, kind="stable")
    bp, measured, z = bp[o], measured[o], z[o]
    S = len(bp)
    # Balding-Nichols frequencies (as gauss_amd/synth.py), kept inside (0.02, 0.98)
    p0 = rng.uniform(0.03, 0.5, S)
    p0 = np.where(rng.random(S) < 0.5, 1 - p0, p0)
    sups = sorted(set(p[2] for p in pops))

    def bn(p, f):
        return np.clip(rng.beta(p * (1 - f) / f, (1 - p) * (1 - f) / f), 0.02, 0.98)
    psup = {s: bn(p0, 0.15) for s in sups}
    ppop = np.stack([bn(psup[p[2]], 0.05) for p in pops], axis=1)
    thr = norm.ppf(ppop).astype(np.float32)
    rho = np.ones(S, dtype=np.float32)
    rho[1:] = np.exp(-np.diff(bp) / 50e3)
    return dict(pops=pops, w=w, off=off, bp=bp, measured=measured, thr=thr, rho=rho, z=z, mode=mode,
                study=os.path.relpath(study, ROOT) if os.path.isabs(study) else study)


def windows_of(ch, wing=500_000, limit=0):
    """[(start_bp, measured row indices, unmeasured row indices)] of every window that passes the guards."""
    bp, meas = ch["bp"], ch["measured"]
    out = []
    start = (int(bp[0]) // WINDOW_BP) * WINDOW_BP + 1
    while start <= bp[-1]:
        end = start + WINDOW_BP - 1
        i0, i1 = np.searchsorted(bp, [start - wing, end + wing + 1])
        j0, j1 = np.searchsorted(bp, [start, end + 1])
        mi = i0 + np.nonzero(meas[i0:i1])[0]
        ui = j0 + np.nonzero(~meas[j0:j1])[0]
        if len(mi) > 10 and len(ui) > 10:                                  # dist.cpp:145-146
            out.append((start, mi, ui))
        start += WINDOW_BP
    return out[:limit] if limit else out


def window_flops(n_samples, m, u):
    """Algorithmic LD flops of one window: N M (M+1) + 2 N U M."""
    return float(n_samples) * m * (m + 1.0 + 2.0 * u)


def shard(wins, n_samples, world):
    """Longest-processing-time assignment of whole windows to ranks (farm.assign_windows) by LD flops.
    Returns (owner per window, per-rank cost list)."""
    from . import farm
    costs = [window_flops(n_samples, len(mi), len(ui)) for _, mi, ui in wins]
    owner = farm.assign_windows(costs, world)
    load = [0.0] * world
    for c, r in zip(costs, owner):
        load[r] += c
    return owner, load


def shard_balanced(wins, n_samples, world, granule=64, contiguous=False):
    """Shares with window cutting: LPT + levelling (farm.level_windows) or contiguous equal-cost stretches
    (farm.balance_windows).  Returns (shares[r] = [(window, u0, u1)], per-rank cost list)."""
    from . import farm
    mu = [(len(mi), len(ui)) for _, mi, ui in wins]
    if contiguous:
        return farm.balance_windows(mu, n_samples, world, granule)
    # GAUSS_PLAN_ADJACENCY=1: the planner also prices neighbouring windows on one rank (farm.adjacency_saving: their common B11
    # tile pairs are multiplied once, gauss_plan.cpp clusters).  Measured on the 8-rank emulation (round 4): nine neighbour pairs end
    # up together and the MEAN share drops 1 % (5.40 -> 5.35 ms), but the slowest share stays at 5.44-5.46 ms -- the ranks' times
    # scatter +-2 % around any flop model, more than the saving -- so the predicted efficiency does not move and it is off.
    shared = None
    if os.environ.get("GAUSS_PLAN_ADJACENCY", "0") != "0":
        shared = [int(len(np.intersect1d(wins[k][1], wins[k + 1][1], assume_unique=True))) for k in range(len(wins) - 1)]
    return farm.level_windows(mu, n_samples, world, granule, shared=shared)


def pieces_of(wins, share):
    """The window tuples of one rank's share: a cut window keeps its measured rows and a slice of the unmeasured ones."""
    return [(wins[k][0], wins[k][1], wins[k][2][u0:u1]) for k, u0, u1 in share]

