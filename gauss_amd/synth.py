"""Seeded synthetic reference panels in the shape of the 33KG panel.

The real 33KG panel (32 953 samples, 29 populations; docs/articles/ref_33KG.md:15-43 of the
reference) is an external download, so every test and benchmark here runs on synthetic
genotypes: Balding-Nichols per-population allele frequencies and a Gaussian-copula AR(1)
haplotype model along base-pair position, which gives realistic LD blocks.
"""
import numpy as np

# (abbreviation, subjects, super-population) -- docs/articles/ref_33KG.md:15-43
POPS_33KG = [
    ("ACB", 164, "AFR"), ("ASW", 162, "AFR"), ("BEB", 86, "SAS"), ("CCE", 3409, "ASN"),
    ("CCS", 2613, "ASN"), ("CDX", 95, "ASN"), ("CEU", 6360, "EUR"), ("CLM", 98, "AMR"),
    ("CNE", 2330, "ASN"), ("CSE", 2020, "ASN"), ("ESN", 140, "AFR"), ("FIN", 3529, "EUR"),
    ("GBR", 2020, "EUR"), ("GIH", 110, "SAS"), ("GWD", 113, "AFR"), ("IBS", 1309, "EUR"),
    ("ITU", 95, "SAS"), ("JPT", 107, "ASN"), ("KHV", 226, "ASN"), ("LWK", 99, "AFR"),
    ("MSL", 87, "AFR"), ("MXL", 187, "AMR"), ("ORK", 5772, "EUR"), ("PEL", 110, "AMR"),
    ("PJL", 121, "SAS"), ("PUR", 138, "AMR"), ("STU", 110, "SAS"), ("TSI", 1291, "EUR"),
    ("YRI", 52, "AFR"),
]

# data/PGC2_SCZ_ANC_Prop.RData decoded in SURVEY.md section 8 (21 populations, sum = 1.061)
PGC2_WEIGHTS = {
    "ACB": .006, "ASW": .036, "BEB": .005, "CCE": .008, "CCS": .004, "CDX": .018, "CEU": .165,
    "CLM": .025, "CNE": .003, "CSE": .012, "FIN": .138, "GBR": .165, "GIH": .006, "IBS": .099,
    "JPT": .011, "KHV": .017, "MXL": .030, "ORK": .166, "PJL": .016, "PUR": .045, "TSI": .086,
}


def pop_table(scale=1.0, min_size=4):
    """33KG population table, optionally with every population shrunk by `scale`."""
    return [(a, max(min_size, int(round(n * scale))), s) for a, n, s in POPS_33KG]


def pop_offsets(sizes):
    return np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.int64))]).astype(np.int32)


def synth_genotypes(bp, pops, seed=20260213, f_within=0.05, f_across=0.15, ld_scale_bp=50e3,
                    maf_lo=0.01):
    """Genotypes {0,1,2} for SNPs at positions `bp` and populations `pops`.

    Returns (G uint8 [S, N] SNP-major with populations concatenated in `pops` order,
             af [S, P] realised per-population frequency of the counted allele).
    """
    rng = np.random.default_rng(seed)
    bp = np.asarray(bp, dtype=np.float64)
    S = len(bp)
    sizes = [p[1] for p in pops]
    sups = [p[2] for p in pops]
    P = len(pops)
    N = int(sum(sizes))
    # ancestral frequency, then super-population, then population (Balding-Nichols)
    p0 = rng.uniform(maf_lo, 0.5, size=S)
    flip = rng.random(S) < 0.5
    p0 = np.where(flip, 1.0 - p0, p0)

    def bn(p, f):
        a = np.maximum(p * (1 - f) / f, 1e-3)
        b = np.maximum((1 - p) * (1 - f) / f, 1e-3)
        return np.clip(rng.beta(a, b), 0.002, 0.998)

    sup_names = sorted(set(sups))
    p_sup = {s: bn(p0, f_across) for s in sup_names}
    p_pop = np.stack([bn(p_sup[sups[k]], f_within) for k in range(P)], axis=1)  # [S, P]
    # AR(1) latent along position, shared correlation structure for every haplotype
    rho = np.ones(S)
    rho[1:] = np.exp(-np.abs(np.diff(bp)) / ld_scale_bp)
    from scipy.stats import norm
    thr = norm.ppf(p_pop)  # allele = latent < thr
    G = np.zeros((S, N), dtype=np.uint8)
    off = pop_offsets(sizes)
    for h in range(2):
        z = rng.standard_normal(N)
        for s in range(S):
            if s:
                z = rho[s] * z + np.sqrt(max(0.0, 1.0 - rho[s] ** 2)) * rng.standard_normal(N)
            for k in range(P):
                G[s, off[k]:off[k + 1]] += (z[off[k]:off[k + 1]] < thr[s, k]).astype(np.uint8)
    af = np.stack([G[:, off[k]:off[k + 1]].sum(1) / (2.0 * sizes[k]) for k in range(P)], axis=1)
    return G, af
