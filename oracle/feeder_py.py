"""Python restatement of the reference's host data layer + drivers -- TEST INFRASTRUCTURE ONLY.

Follows src/gauss.cpp (ReadInputZ 121-190, ReadReferenceIndex 293-399, ReadReferenceIndexAll
431-518, MakeSnpVec 543-604, MakeSnpVecMix 631-693, ReadGenotype 720-785, read_ref_desc 951-993,
init_pop_flag_vec 1019-1066, init_pop_flag_wgt_vec 1093-1117, ReadAnnotation 1275-1361,
MakeGeneStartEndVec 1383-1439) and the drivers (computeLD.cpp, dist.cpp, distmix.cpp, jepeg.cpp,
jepegmix.cpp); the numeric part calls the C oracle.  Written independently of
gauss_amd/csrc/host (C++), so agreement between the two checks both.
"""
import math
import struct
import zlib

import numpy as np

from . import oracle_c as oc


class Bgzf:
    """Minimal BGZF text reader with virtual-offset seek (bgzf.c:702-727 convention)."""

    def __init__(self, path):
        self.f = open(path, "rb")
        self.addr = None
        self.data = b""
        self.next = 0

    def _load(self, addr):
        self.f.seek(addr)
        hdr = self.f.read(12)
        if len(hdr) < 12:
            self.addr, self.data, self.next = addr, b"", addr
            return
        xlen = struct.unpack("<H", hdr[10:12])[0]
        extra = self.f.read(xlen)
        bsize, p = None, 0
        while p + 4 <= xlen:
            slen = struct.unpack("<H", extra[p + 2:p + 4])[0]
            if extra[p:p + 2] == b"BC":
                bsize = struct.unpack("<H", extra[p + 4:p + 6])[0]
            p += 4 + slen
        total = bsize + 1
        comp = self.f.read(total - 12 - xlen - 8)
        self.f.read(8)
        self.addr, self.data, self.next = addr, zlib.decompress(comp, -15), addr + total

    def line_at(self, voff):
        """The text line starting at virtual offset voff ('' past EOF, like the reference)."""
        if voff < 0:
            return ""
        addr, off = voff >> 16, voff & 0xFFFF
        if self.addr != addr:
            self._load(addr)
        out = b""
        while True:
            if not self.data:
                return out.decode()
            i = self.data.find(b"\n", off)
            if i >= 0:
                return (out + self.data[off:i]).decode()
            out += self.data[off:]
            self._load(self.next)
            off = 0

    def lines(self):
        addr = 0
        buf = b""
        while True:
            self._load(addr)
            if not self.data:
                break
            buf += self.data
            addr = self.next
        text = buf.decode()
        return text.split("\n")[:-1] if text.endswith("\n") else text.split("\n")


class Snp:
    def __init__(self):
        self.rsid, self.chr, self.bp, self.a1, self.a2 = ".", -1, -1, ".", "."
        self.af1mix = self.af1ref = -1.0
        self.z, self.info, self.type, self.fpos = 0.0, -1.0, -1, -1
        self.geneid, self.categ, self.geno = ".", {}, None


class Args:
    lam, min_abs_eig = 0.1, 1e-5
    min_measured = min_unmeasured = 10
    categ_cor_cutoff, denorm_norm_w = 0.8, 3


def read_ref_desc(path):
    pops = []
    with open(path) as f:
        f.readline()
        for line in f:
            t = line.split()
            if len(t) >= 3:
                pops.append((t[0], int(t[1]), t[2]))
    return pops


def pop_flags(pops, study_pop):
    names, sups = [p[0] for p in pops], [p[2] for p in pops]
    in_pop, in_sup = names.count(study_pop), sups.count(study_pop)
    if in_pop == 0 and in_sup == 0:
        raise ValueError(f"ERROR: invalid population name '{study_pop}'")
    pv = names if in_pop else sups
    return [1 if x == study_pop else 0 for x in pv]


def pop_flags_wgt(pops, names, wgts):
    m = {str(n).upper(): float(w) for n, w in zip(names, wgts)}
    flags, w = [], []
    for p in pops:
        if p[0] in m:
            flags.append(1)
            w.append(m[p[0]])
        else:
            flags.append(0)
    return flags, w


def read_input_z(path, chr_, lo, hi, All):
    m = {}
    with open(path) as f:
        f.readline()
        for line in f:
            t = line.split()
            if len(t) < 6:
                continue
            rsid, c, bp, a1, a2, z = t[0], int(t[1]), int(t[2]), t[3], t[4], float(t[5])
            if not All:
                if chr_ > 0 and chr_ != c:
                    continue
                if lo > bp or hi < bp:
                    continue
            s = Snp()
            s.rsid, s.chr, s.bp, s.a1, s.a2, s.z, s.info, s.type = rsid, c, bp, a1, a2, z, 1.0, 2
            m[(c, bp, a1, a2)] = s
    return m


def read_reference_index(m, path, chr_, lo, hi, All):
    for line in Bgzf(path).lines():
        t = line.split()
        if len(t) < 7:
            continue
        rsid, c, bp, a1, a2, fpos = t[0], int(t[1]), int(t[2]), t[3], t[4], int(t[6])
        if not All:
            if chr_ > 0 and chr_ != c:
                continue
            if lo > bp or hi < bp:
                continue
        k1, k2 = (c, bp, a1, a2), (c, bp, a2, a1)
        i1, i2 = k1 in m, k2 in m
        if i1 and not i2:
            s = m[k1]
            s.rsid, s.type, s.fpos = rsid, 1, fpos
        elif i2 and not i1:
            s = m.pop(k2)
            s.rsid, s.a1, s.a2, s.z, s.type, s.fpos = rsid, a1, a2, -s.z, 1, fpos
            m[k1] = s
        elif not i1 and not i2:
            if not All:
                s = Snp()
                s.rsid, s.chr, s.bp, s.a1, s.a2, s.type, s.fpos = rsid, c, bp, a1, a2, 0, fpos
                m[k1] = s
        else:
            raise ValueError("ERROR: input file contains duplicates")


CATEG = {"PROTEIN": 0, "TFBS": 1, "WTH_HAIR": 2, "WTH_TARGET": 3, "CIS_EQTL": 4, "TRANS_EQTL": 5}


def read_annotation(m, path):
    num = 0
    with open(path) as f:
        f.readline()
        for line in f:
            t = line.split()
            if len(t) < 8:
                continue
            c, bp, a1, a2, geneid, categ, wgt = int(t[1]), int(t[2]), t[3], t[4], t[5], t[6], float(t[7])
            num = CATEG.get(categ, num)
            k1, k2 = (c, bp, a1, a2), (c, bp, a2, a1)
            if k1 in m and k2 not in m:
                m[k1].geneid = geneid
                m[k1].categ[num] = wgt
            elif k1 not in m and k2 in m:
                s = m.pop(k2)
                s.a1, s.a2, s.af1ref, s.z, s.geneid = a1, a2, 1 - s.af1ref, -s.z, geneid
                s.categ[num] = wgt
                m[k1] = s


def _parse_line(line, flags):
    t = line.split()
    P = len(flags)
    geno = [t[k] if k < len(t) else "" for k in range(P)]
    afs = [float(t[P + k]) if P + k < len(t) else 0.0 for k in range(P)]
    return [g for g, f in zip(geno, flags) if f], [a for a, f in zip(afs, flags) if f]


def make_snp_vec(m, data_path, flags, cutoff, wgt=None):
    """MakeSnpVec (wgt None) / MakeSnpVecMix; also caches the genotype strings (ReadGenotype)."""
    bg = Bgzf(data_path)
    out = []
    for key in sorted(m.keys()):
        s = m[key]
        geno, afs = _parse_line(bg.line_at(s.fpos), flags)
        s.geno = geno
        if wgt is None:
            n = sum(len(g) for g in geno)
            cnt = float(sum(sum(ord(ch) - 48 for ch in g) for g in geno))
            af = cnt / (2 * n) if n else float("nan")
            af = math.ceil(af * 100000.0) / 100000.0 if af == af else af
            s.af1ref = af
            if af > cutoff and af < 1 - cutoff:
                out.append(s)
        else:
            af = 0.0
            for a, w in zip(afs, wgt):
                af += a * w
            if af > cutoff and af < 1 - cutoff:
                s.af1mix = af
                out.append(s)
    return out


def _matrix(snps):
    if not snps:
        return np.zeros((0, 0), dtype=np.uint8)
    return np.array([np.frombuffer("".join(s.geno).encode(), dtype=np.uint8) for s in snps])


def _selected_off(pops, flags):
    off = [0]
    for p, f in zip(pops, flags):
        if f:
            off.append(off[-1] + p[1])
    return np.array(off, dtype=np.int32)


def _impute(kind_mix, chr_, start_bp, end_bp, wing, study_pop, pop_wgt, files, af1_cutoff):
    input_file, index, data, desc = files
    cutoff = 0.01 if af1_cutoff is None else af1_cutoff
    pops = read_ref_desc(desc)
    if kind_mix:
        flags, w = pop_flags_wgt(pops, *pop_wgt)
    else:
        flags, w = pop_flags(pops, study_pop), None
    lo, hi = start_bp - wing, end_bp + wing
    m = read_input_z(input_file, chr_, lo, hi, False)
    read_reference_index(m, index, chr_, lo, hi, False)
    vec = make_snp_vec(m, data, flags, cutoff, w)
    meas = [s for s in vec if s.type == 1]
    unme = [s for s in vec if s.type == 0 and start_bp <= s.bp <= end_bp]
    if len(meas) <= Args.min_measured or len(unme) <= Args.min_unmeasured:
        raise ValueError("Not enough number of SNPs loaded")
    off = _selected_off(pops, flags)
    res = oc.run_impute(1 if kind_mix else 0, _matrix(meas), _matrix(unme), off, w, [s.z for s in meas],
                        Args.lam, Args.min_abs_eig)
    for s, z, info in zip(unme, res["z"], res["info"]):
        s.z, s.info = float(z), float(info)
    rows = [s for s in vec if start_bp <= s.bp <= end_bp]
    return dict(rsid=[s.rsid for s in rows], chr=[s.chr for s in rows], bp=[s.bp for s in rows],
                a1=[s.a1 for s in rows], a2=[s.a2 for s in rows],
                af=[(s.af1mix if kind_mix else s.af1ref) for s in rows], z=[s.z for s in rows],
                pval=[2 * oc.pnorm_upper(abs(s.z)) for s in rows], info=[s.info for s in rows],
                type=[s.type for s in rows], n_measured=len(meas), n_unmeasured=len(unme), mpd=res["mpd"])


def dist(chr_, start_bp, end_bp, wing, study_pop, input_file, index, data, desc, af1_cutoff=None):
    return _impute(False, chr_, start_bp, end_bp, wing, study_pop, None, (input_file, index, data, desc), af1_cutoff)


def distmix(chr_, start_bp, end_bp, wing, pop_wgt, input_file, index, data, desc, af1_cutoff=None):
    return _impute(True, chr_, start_bp, end_bp, wing, None, pop_wgt, (input_file, index, data, desc), af1_cutoff)


def _qcat(kind_mix, chr_, start_bp, end_bp, wing, study_pop, pop_wgt, files, af1_cutoff):
    """qcat.cpp:30-262 / qcatmix.cpp:30-297: the dist / distmix feeder, then run_qcat[mix]."""
    import math
    input_file, index, data, desc = files
    cutoff = (0.01 if kind_mix else 0.05) if af1_cutoff is None else af1_cutoff     # qcatmix.cpp:61-65, qcat.cpp:53-57
    pops = read_ref_desc(desc)
    if kind_mix:
        flags, w = pop_flags_wgt(pops, *pop_wgt)
    else:
        flags, w = pop_flags(pops, study_pop), None
    lo, hi = start_bp - wing, end_bp + wing
    m = read_input_z(input_file, chr_, lo, hi, False)
    read_reference_index(m, index, chr_, lo, hi, False)
    vec = make_snp_vec(m, data, flags, cutoff, w)
    meas = [s for s in vec if s.type == 1]
    unme = [s for s in vec if s.type == 0 and start_bp <= s.bp <= end_bp]
    n_head = sum(1 for s in meas if s.bp < start_bp)
    n_pred = sum(1 for s in meas if start_bp <= s.bp <= end_bp)
    if len(meas) <= Args.min_measured or (kind_mix and len(unme) <= Args.min_unmeasured):
        raise ValueError("Not enough number of SNPs loaded")        # qcat.cpp:157, qcatmix.cpp:168
    off = _selected_off(pops, flags)
    res = oc.run_qcat(1 if kind_mix else 0, _matrix(meas), _matrix(unme) if unme else None, off, w,
                      [s.z for s in meas], n_head, n_pred, Args.lam, 0.01)
    ne = res["num_eig"]
    for s in vec:
        s.qcat_m, s.qcat_t, s.qcat_chisq = 0, 0.0, 0.0              # snp.cpp:26-28
    tested = meas[n_head:n_head + n_pred] + unme
    for s, r in zip(tested, res["r"]):
        s.qcat_m = ne
        s.qcat_t = (math.sqrt(ne - 3) if ne >= 3 else float("nan")) * float(r)
        s.qcat_chisq = (ne - 3) * float(r) * float(r)
    rows = [s for s in vec if start_bp <= s.bp <= end_bp]
    return dict(rsid=[s.rsid for s in rows], chr=[s.chr for s in rows], bp=[s.bp for s in rows],
                a1=[s.a1 for s in rows], a2=[s.a2 for s in rows],
                af=[(s.af1mix if kind_mix else s.af1ref) for s in rows], z=[s.z for s in rows],
                qcat_m=[s.qcat_m for s in rows], qcat_t=[s.qcat_t for s in rows],
                qcat_chisq=[s.qcat_chisq for s in rows],
                qcat_pval=[oc.pchisq_upper(s.qcat_chisq, 1) for s in rows],
                type=[s.type for s in rows], n_measured=len(meas), n_unmeasured=len(unme),
                n_head=n_head, n_pred=n_pred, num_eig=ne)


def qcat(chr_, start_bp, end_bp, wing, study_pop, input_file, index, data, desc, af1_cutoff=None):
    return _qcat(False, chr_, start_bp, end_bp, wing, study_pop, None, (input_file, index, data, desc), af1_cutoff)


def qcatmix(chr_, start_bp, end_bp, wing, pop_wgt, input_file, index, data, desc, af1_cutoff=None):
    return _qcat(True, chr_, start_bp, end_bp, wing, None, pop_wgt, (input_file, index, data, desc), af1_cutoff)


def prep_qcat(chr_, start_bp, end_bp, wing, study_pop, input_file, index, data, desc, af1_cutoff=None):
    """prep_qcat.cpp:16-205."""
    cutoff = 0.01 if af1_cutoff is None else af1_cutoff
    pops = read_ref_desc(desc)
    flags = pop_flags(pops, study_pop)
    lo, hi = start_bp - wing, end_bp + wing
    m = read_input_z(input_file, chr_, lo, hi, False)
    read_reference_index(m, index, chr_, lo, hi, False)
    vec = make_snp_vec(m, data, flags, cutoff, None)
    pred = [s for s in vec if s.type != 2 and start_bp <= s.bp <= end_bp]
    meas = [s for s in vec if s.type == 1]
    if len(meas) <= Args.min_measured:
        raise ValueError("Not enough number of SNPs loaded")
    res = oc.ld_blocks(0, _matrix(meas), _matrix(pred), _selected_off(pops, flags), None, 1.0, (0,))
    return dict(rsid=[s.rsid for s in vec], bp=[s.bp for s in vec], af=[s.af1ref for s in vec],
                z=[s.z for s in vec], type=[s.type for s in vec],
                z_vec=np.array([s.z for s in meas]), cor_mat1=res["b11"], cor_mat2=res["b21"])


def prep_recessive_impute(chr_, start_bp, end_bp, wing, pop_wgt, input_file, index, data, desc, af1_cutoff=None):
    """prep_qcatmix.cpp:36-316 incl. UpdateSnpToMinorAllele (gauss.cpp:1137-1184)."""
    cutoff = 0.01 if af1_cutoff is None else af1_cutoff
    pops = read_ref_desc(desc)
    flags, w = pop_flags_wgt(pops, *pop_wgt)
    lo, hi = start_bp - wing, end_bp + wing
    m = read_input_z(input_file, chr_, lo, hi, False)
    read_reference_index(m, index, chr_, lo, hi, False)
    vec = make_snp_vec(m, data, flags, cutoff, w)
    flip = str.maketrans("012", "210")
    for s in vec:
        if s.af1mix > 0.5:
            s.af1mix = 1 - s.af1mix
            s.z = -s.z
            s.a1, s.a2 = s.a2, s.a1
            s.geno = [g.translate(flip) for g in s.geno]
    pred = [s for s in vec if s.type != 2 and start_bp <= s.bp <= end_bp]
    meas = [s for s in vec if s.type == 1]
    if len(meas) <= Args.min_measured:
        raise ValueError("Not enough number of SNPs loaded")
    res = oc.ld_blocks(1, _matrix(meas), _matrix(pred), _selected_off(pops, flags), w, 1.0, (0, 1, 2))
    A = len(pred)
    return dict(rsid=[s.rsid for s in pred], bp=[s.bp for s in pred], a1=[s.a1 for s in pred], a2=[s.a2 for s in pred],
                af=[s.af1mix for s in pred], z=[s.z for s in pred], type=[s.type for s in pred],
                zvec=np.array([s.z for s in meas]), cormat=res["b11"], cormat_add=res["b21"][:A],
                cormat_dom=res["b21"][A:2 * A], cormat_rec=res["b21"][2 * A:])


def r_quantile7(x, p):
    """stats::quantile(x, p), type 7, as written in R's quantile.default."""
    import math
    x = sorted(float(v) for v in x)
    n = len(x)
    if n == 0:
        return float("nan")
    if any(math.isnan(v) for v in x):
        raise ValueError("missing values and NaN's not allowed if 'na.rm' is FALSE")
    index = 1 + (n - 1) * p
    lo, hi = math.floor(index), math.ceil(index)
    qs = x[lo - 1]
    if index > lo and x[hi - 1] != qs:
        h = index - lo
        qs = (1 - h) * qs + h * x[hi - 1]
    return qs


def prep_zmix5(input_file, index, data, desc, percentile=None, interval=None):
    """zmix.cpp:44-190 with stats::quantile(type 7) from numpy (same definition, 'linear')."""
    pct = 0.99 if percentile is None else percentile
    step = interval or 1
    pops = read_ref_desc(desc)
    flags = [1] * len(pops)
    m = read_input_z(input_file, 0, 0, 0, True)
    read_reference_index(m, index, 0, 0, 0, True)
    measured = [s for _, s in sorted(m.items()) if s.type == 1]
    vec = measured[::step]
    bg = Bgzf(data)
    nv = []
    for s in vec:
        toks = bg.line_at(s.fpos).split()
        af = np.array([float(x) for x in toks[len(pops):2 * len(pops)]])
        mean = 0.0
        for v in af:
            mean += v
        mean /= len(af)
        sq = 0.0
        for v in af:
            sq += v * v
        nv.append((sq / len(af) - mean * mean) / (mean * (1 - mean)))
        s.geno = toks[:len(pops)]
    nv = np.array(nv)
    cutoff = r_quantile7(nv, pct)
    sub = [s for s, v in zip(vec, nv) if v > cutoff]
    S = len(sub)
    rows = S * (S - 1) // 2
    out = np.zeros((rows, 1 + len(pops)))
    if S > 1:
        off = _selected_off(pops, flags)
        r = oc.ld_per_pop(_matrix(sub), off)
        out[:, 1:] = r.T
        k = 0
        for i in range(S):
            for j in range(i + 1, S):
                out[k, 0] = sub[i].z * sub[j].z
                k += 1
    return dict(data_mat=out, rsid=[s.rsid for s in sub], cutoff=cutoff, norm_var=[float(v) for s, v in zip(vec, nv) if v > cutoff])


def computeLD(chr_, start_bp, end_bp, pop_wgt, input_file, index, data, desc, af1_cutoff=None):
    cutoff = 0.01 if af1_cutoff is None else af1_cutoff
    pops = read_ref_desc(desc)
    flags, w = pop_flags_wgt(pops, *pop_wgt)
    m = read_input_z(input_file, chr_, start_bp, end_bp, False)
    read_reference_index(m, index, chr_, start_bp, end_bp, False)
    vec = make_snp_vec(m, data, flags, cutoff, w)
    meas = [s for s in vec if s.type == 1]
    if len(meas) <= Args.min_measured:
        raise ValueError("Not enough number of SNPs loaded")
    cor = oc.compute_ld(_matrix(meas), _selected_off(pops, flags), w)
    return dict(rsid=[s.rsid for s in meas], bp=[s.bp for s in meas], a1=[s.a1 for s in meas],
                a2=[s.a2 for s in meas], af1mix=[s.af1mix for s in meas], cormat=cor)


def _jepeg(kind_mix, study_pop, pop_wgt, input_file, annot, index, data, desc, af1_cutoff):
    cutoff = 0.01 if af1_cutoff is None else af1_cutoff
    pops = read_ref_desc(desc)
    if kind_mix:
        flags, w = pop_flags_wgt(pops, *pop_wgt)
    else:
        flags, w = pop_flags(pops, study_pop), None
    m = read_input_z(input_file, 0, 0, 0, True)
    read_reference_index(m, index, 0, 0, 0, True)
    read_annotation(m, annot)
    vec = make_snp_vec(m, data, flags, cutoff, w)
    gsnps = [s for s in vec if s.geneid != "." and s.type == 1]
    gsnps.sort(key=lambda s: s.geneid)      # std::sort is unstable: order inside a gene may differ
    off = _selected_off(pops, flags)
    out = []
    i = 0
    while i < len(gsnps):
        j = i
        while j < len(gsnps) and gsnps[j].geneid == gsnps[i].geneid:
            j += 1
        gs = gsnps[i:j]
        G = _matrix(gs)
        if kind_mix:
            corg = oc.compute_ld(G, off, w)
            np.fill_diagonal(corg, 1.0 + Args.lam)
        else:
            corg = oc.ld_pooled(G, off, 1.0 + Args.lam)
        has = np.zeros((len(gs), 6), dtype=np.int32)
        wg = np.zeros((len(gs), 6))
        for r, s in enumerate(gs):
            for c, v in s.categ.items():
                has[r, c], wg[r, c] = 1, v
        r = oc.jepeg_gene_tail(corg, [s.z for s in gs], [s.info for s in gs], has, wg, Args.min_abs_eig,
                               Args.categ_cor_cutoff, Args.denorm_norm_w)
        r["geneid"] = gs[0].geneid if r["df"] else "."
        r["snps"] = sorted(s.rsid for s in gs)
        r["top_snp_id"] = gs[r["top_snp"]].rsid if r["top_snp"] >= 0 else "."
        out.append(r)
        i = j
    return out


def jepeg(study_pop, input_file, annot, index, data, desc, af1_cutoff=None):
    return _jepeg(False, study_pop, None, input_file, annot, index, data, desc, af1_cutoff)


def jepegmix(pop_wgt, input_file, annot, index, data, desc, af1_cutoff=None):
    return _jepeg(True, None, pop_wgt, input_file, annot, index, data, desc, af1_cutoff)


def _pearson_strings(xs, ys):
    """CalCor on one population's genotype strings (util.cpp:153-169) / CalCorSup on several concatenated
    (zmix.cpp:1221-1246): plain sums over the samples, then (n Sxy - Sx Sy) / (sqrt(n Sxx - Sx^2) sqrt(n Syy - Sy^2))."""
    x = np.frombuffer(xs.encode(), dtype=np.uint8).astype(np.float64) - 48.0
    y = np.frombuffer(ys.encode(), dtype=np.uint8).astype(np.float64) - 48.0
    n = len(x)
    sx, sy, sxx, syy, sxy = x.sum(), y.sum(), (x * x).sum(), (y * y).sum(), (x * y).sum()
    with np.errstate(invalid="ignore", divide="ignore"):
        return (n * sxy - sx * sy) / (np.sqrt(n * sxx - sx * sx) * np.sqrt(n * syy - sy * sy))


def prep_zmix_variant(variant, input_file, index, data, desc, percentile=None, interval=None, p2=None):
    """The other prep_zmix selectors, loop for loop: "zmix" zmix.cpp:940-1076, "zmix2" :651-760, "zmix3" :511-650,
    "zmix4" :363-510, "zmix5_sup" :201-361.  Returns dict(data_mat, pairs as (rsid_i, rsid_j), groups)."""
    pops = read_ref_desc(desc)
    P = len(pops)
    m = read_input_z(input_file, 0, 0, 0, True)
    read_reference_index(m, index, 0, 0, 0, True)
    measured = [s for _, s in sorted(m.items()) if s.type == 1]
    n = len(measured)
    bg = Bgzf(data)

    def load(s):
        if s.geno is None:
            toks = bg.line_at(s.fpos).split()
            s.geno = toks[:P]
            s.af = [float(x) for x in toks[P:2 * P]]
        return s

    step = interval if interval else (1 if variant in ("zmix", "zmix5_sup") else 1000)
    par2 = p2 if p2 else (5 if variant == "zmix3" else 3)
    pairs, lead = [], []
    if variant in ("zmix", "zmix3", "zmix5_sup"):
        sub = measured[::step]
        if variant == "zmix5_sup":
            pct = 0.99 if percentile is None else percentile
            nv = []
            for s in sub:
                af = load(s).af
                mean = 0.0
                for v in af:
                    mean += v
                mean /= len(af)
                sq = 0.0
                for v in af:
                    sq += v * v
                nv.append((sq / len(af) - mean * mean) / (mean * (1 - mean)))
            cutoff = r_quantile7(nv, pct)
            sub = [s for s, v in zip(sub, nv) if v > cutoff]
        S = len(sub)
        for i in range(S):
            for j in range(i + 1, min(i + 1 + par2, S) if variant == "zmix3" else S):
                pairs.append((sub[i], sub[j]))
    elif variant == "zmix2":
        i = 0
        while i + par2 < n:
            pairs.append((measured[i], measured[i + par2]))
            i += step
    elif variant == "zmix4":
        for h in range(step):
            i = h
            while i < n and i + par2 < n:
                pairs.append((measured[i], measured[i + par2]))
                lead.append(float(h))
                i += step
    else:
        raise ValueError(variant)
    if variant == "zmix5_sup":
        groups = []
        for p in pops:
            if p[2] not in groups:
                groups.append(p[2])
        members = [[k for k, p in enumerate(pops) if p[2] == g] for g in groups]
    else:
        groups = [p[0] for p in pops]
        members = [[k] for k in range(P)]
    nl = 1 if variant == "zmix4" else 0
    out = np.zeros((len(pairs), nl + 1 + len(groups)))
    for r, (a, b) in enumerate(pairs):
        load(a)
        load(b)
        if nl:
            out[r, 0] = lead[r]
        out[r, nl] = a.z * b.z
        for g, mem in enumerate(members):
            out[r, nl + 1 + g] = _pearson_strings("".join(a.geno[k] for k in mem), "".join(b.geno[k] for k in mem))
    return dict(data_mat=out, pairs=[(a.rsid, b.rsid) for a, b in pairs], groups=groups)
