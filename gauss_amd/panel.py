"""Writers for the reference's on-disk formats, used to build synthetic panels.

* population description: header + ``pop_abbr n_subj super_pop``           (gauss.cpp:970-984)
* panel data  (BGZF text): one line per SNP = P genotype strings + P allele frequencies
                                                                          (gauss.cpp:755-763, 660-674)
* panel index (BGZF text): ``rsid chr bp a1 a2 af1ref fpos``, fpos = BGZF virtual offset of the
  SNP's data line                                                          (gauss.cpp:328-330)
* GWAS input: header + ``rsid chr bp a1 a2 z``                            (gauss.cpp:146-152)
* annotation: header + ``rsid chr bp a1 a2 geneid categ wgt``             (gauss.cpp:1308)

BGZF = concatenated gzip members (<= 64 KiB each) with a "BC" extra field; a virtual offset is
(member file offset << 16) | offset inside the member (bgzf.c:702-727).
"""
import struct
import zlib

import numpy as np

_BLOCK = 0xFF00


class BgzfTextWriter:
    def __init__(self, path, level=6):
        self.f = open(path, "wb")
        self.level = level
        self.buf = bytearray()
        self.addr = 0

    def tell(self):
        return (self.addr << 16) | len(self.buf)

    def write(self, data: bytes):
        mv = memoryview(data)
        while len(mv):
            room = _BLOCK - len(self.buf)
            take = min(room, len(mv))
            self.buf += mv[:take]
            mv = mv[take:]
            if len(self.buf) == _BLOCK:
                self._flush()

    def _flush(self):
        raw = bytes(self.buf)
        co = zlib.compressobj(self.level, zlib.DEFLATED, -15)
        comp = co.compress(raw) + co.flush()
        total = len(comp) + 26
        hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, total - 1)
        self.f.write(hdr + comp + struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw)))
        self.addr += total
        self.buf = bytearray()

    def close(self):
        if self.buf:
            self._flush()
        self._flush()          # empty member = BGZF end-of-file marker
        self.f.close()


def write_pop_desc(path, pops):
    with open(path, "w") as f:
        f.write("Population_Abbreviation Number_of_Subjects Super_Population\n")
        for abbr, n, sup in pops:
            f.write(f"{abbr} {n} {sup}\n")


def write_panel(index_path, data_path, rsid, chr_, bp, a1, a2, G, af, pop_sizes):
    """G: uint8 [S, N] genotypes over ALL panel populations (panel order); af: [S, P]."""
    off = np.concatenate([[0], np.cumsum(pop_sizes)]).astype(int)
    dw = BgzfTextWriter(data_path)
    iw = BgzfTextWriter(index_path)
    asc = (np.asarray(G, dtype=np.uint8) + ord("0"))
    for s in range(len(rsid)):
        fpos = dw.tell()
        row = asc[s].tobytes()
        parts = [row[off[k]:off[k + 1]] for k in range(len(pop_sizes))]
        line = b" ".join(parts) + b" " + " ".join(repr(float(x)) for x in af[s]).encode() + b"\n"
        dw.write(line)
        af1ref = float(np.asarray(G[s], dtype=np.float64).sum() / (2.0 * G.shape[1]))
        iw.write(f"{rsid[s]} {chr_[s]} {bp[s]} {a1[s]} {a2[s]} {af1ref:.6f} {fpos}\n".encode())
    dw.close()
    iw.close()


def _bgzf_member(raw, level):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(raw) + co.flush()
    total = len(comp) + 26
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, total - 1)
    return hdr + comp + struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw))


def write_bgzf_parallel(path, data, threads=8, level=1):
    """`data` (bytes-like) as one BGZF file: members of 0xFF00 bytes deflated on a thread pool (zlib releases the GIL),
    then written in order.  Returns the file offset of every member (for virtual offsets: (member offset << 16) | offset
    inside the member, bgzf.c:702-727) and the compressed size."""
    from concurrent.futures import ThreadPoolExecutor
    mv = memoryview(data)
    starts = list(range(0, len(mv), _BLOCK))
    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        members = list(pool.map(lambda o: _bgzf_member(bytes(mv[o:o + _BLOCK]), level), starts, chunksize=64))
    offs, addr = [], 0
    with open(path, "wb") as f:
        for m in members:
            offs.append(addr)
            f.write(m)
            addr += len(m)
        f.write(_bgzf_member(b"", level))                  # empty member = BGZF end-of-file marker
        addr += 28
    return np.array(offs, dtype=np.int64), addr


def write_panel_fast(index_path, data_path, rsid, chr_, bp, a1, a2, G, af, pop_sizes, threads=8, level=1, slab=20_000):
    """write_panel for panels of chromosome size (tens of thousands of 33 kB lines): the genotype part of every line is built
    as one array operation, the members are deflated on a thread pool.  Same format, same virtual offsets
    (gauss.cpp:328-330, 755-763).  G: the (S, N) genotype matrix, or a callable G(a, b) that returns rows [a, b) (a chromosome's
    text is 3.4 GB: it is built and deflated `slab` lines at a time, every slab starting a new BGZF member -- members are
    independent, so the file is what one pass would write except for one short member per slab).  Returns the bytes of
    inflated data text."""
    from concurrent.futures import ThreadPoolExecutor
    S = len(bp)
    P = len(pop_sizes)
    N = int(sum(pop_sizes))
    off = np.concatenate([[0], np.cumsum(pop_sizes)]).astype(int)
    rows = G if callable(G) else (lambda a, b: np.asarray(G[a:b], dtype=np.uint8))
    idx_lines = []
    total, addr = 0, 0
    with open(data_path, "wb") as f, ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        for s0 in range(0, S, slab):
            s1 = min(S, s0 + slab)
            Gs = np.asarray(rows(s0, s1), dtype=np.uint8)
            n = s1 - s0
            geno = np.full((n, N + P), ord(" "), dtype=np.uint8)      # P genotype strings, a blank behind each
            for k in range(P):
                geno[:, off[k] + k:off[k + 1] + k] = Gs[:, off[k]:off[k + 1]] + ord("0")
            tails = [(" ".join(repr(float(x)) for x in af[s0 + i]) + "\n").encode() for i in range(n)]
            lens = np.array([N + P + len(t) for t in tails], dtype=np.int64)
            line_off = np.concatenate([[0], np.cumsum(lens)])
            buf = bytearray(int(line_off[-1]))
            mv = memoryview(buf)
            for i in range(n):
                o = int(line_off[i])
                mv[o:o + N + P] = geno[i].tobytes()
                mv[o + N + P:o + int(lens[i])] = tails[i]
            del geno
            starts = list(range(0, len(mv), _BLOCK))
            members = list(pool.map(lambda o: _bgzf_member(bytes(mv[o:o + _BLOCK]), level), starts, chunksize=64))
            member_off = np.empty(len(members), dtype=np.int64)
            for j, m in enumerate(members):
                member_off[j] = addr
                f.write(m)
                addr += len(m)
            blk = line_off[:-1] // _BLOCK
            fpos = (member_off[blk] << 16) | (line_off[:-1] - blk * _BLOCK)
            cnt = Gs.sum(axis=1, dtype=np.int64)
            idx_lines.append("".join(f"{rsid[s0 + i]} {chr_[s0 + i]} {bp[s0 + i]} {a1[s0 + i]} {a2[s0 + i]} {cnt[i] / (2.0 * N):.6f} {int(fpos[i])}\n"
                                     for i in range(n)))
            total += int(line_off[-1])
            del buf, mv, members
        f.write(_bgzf_member(b"", level))                  # empty member = BGZF end-of-file marker
    write_bgzf_parallel(index_path, "".join(idx_lines).encode(), threads, level)
    return total


def write_gwas(path, rsid, chr_, bp, a1, a2, z):
    with open(path, "w") as f:
        f.write("rsid chr bp a1 a2 z\n")
        for i in range(len(rsid)):
            f.write(f"{rsid[i]} {chr_[i]} {bp[i]} {a1[i]} {a2[i]} {float(z[i])!r}\n")


def write_annotation(path, rows):
    """rows: iterable of (rsid, chr, bp, a1, a2, geneid, categ, wgt)."""
    with open(path, "w") as f:
        f.write("rsid chr bp a1 a2 geneid categ wgt\n")
        for r in rows:
            f.write(" ".join(str(x) if not isinstance(x, float) else repr(x) for x in r) + "\n")


CATEGS = ["PROTEIN", "TFBS", "WTH_HAIR", "WTH_TARGET", "CIS_EQTL", "TRANS_EQTL"]


def make_synthetic_study(outdir, pops, n_snp=400, chr_=22, bp_lo=1_000_000, bp_hi=3_000_000, frac_measured=0.35,
                         frac_swapped=0.1, frac_not_in_panel=0.02, n_genes=0, seed=11, prefix="syn"):
    """Write a complete synthetic study (panel + GWAS [+ annotation]) and return its description."""
    import os
    from . import synth
    rng = np.random.default_rng(seed)
    bp = np.sort(rng.choice(np.arange(bp_lo, bp_hi), size=n_snp, replace=False))
    G, af = synth.synth_genotypes(bp, pops, seed=seed + 1)
    alle = np.array(list("ACGT"))
    a1 = alle[rng.integers(0, 4, n_snp)]
    a2 = alle[(np.searchsorted(alle, a1) + rng.integers(1, 4, n_snp)) % 4]
    rsid = np.array([f"rs{100000 + i}" for i in range(n_snp)])
    chrs = np.full(n_snp, chr_)
    sizes = [p[1] for p in pops]
    paths = {k: os.path.join(outdir, f"{prefix}_{k}") for k in ("desc.txt", "index.gz", "data.gz", "gwas.txt", "annot.txt")}
    write_pop_desc(paths["desc.txt"], pops)
    write_panel(paths["index.gz"], paths["data.gz"], rsid, chrs, bp, a1, a2, G, af, sizes)
    meas = np.nonzero(rng.random(n_snp) < frac_measured)[0]
    z = rng.standard_normal(len(meas)) * 2.0
    g_rsid, g_chr, g_bp, g_a1, g_a2 = rsid[meas].copy(), chrs[meas].copy(), bp[meas].copy(), a1[meas].copy(), a2[meas].copy()
    swap = rng.random(len(meas)) < frac_swapped          # GWAS reports the alleles the other way round
    g_a1[swap], g_a2[swap] = a2[meas][swap], a1[meas][swap]
    n_extra = int(round(len(meas) * frac_not_in_panel))  # GWAS SNPs the panel does not have (type 2)
    if n_extra:
        ebp = rng.choice(np.setdiff1d(np.arange(bp_lo, bp_hi), bp), size=n_extra, replace=False)
        g_rsid = np.concatenate([g_rsid, [f"rsX{i}" for i in range(n_extra)]])
        g_chr = np.concatenate([g_chr, np.full(n_extra, chr_)])
        g_bp = np.concatenate([g_bp, ebp])
        g_a1 = np.concatenate([g_a1, np.full(n_extra, "A")])
        g_a2 = np.concatenate([g_a2, np.full(n_extra, "G")])
        z = np.concatenate([z, rng.standard_normal(n_extra)])
    order = rng.permutation(len(g_rsid))                 # GWAS files are not position sorted
    write_gwas(paths["gwas.txt"], g_rsid[order], g_chr[order], g_bp[order], g_a1[order], g_a2[order], z[order])
    annot = []
    if n_genes:
        for g in range(n_genes):
            k = int(rng.integers(1, 9))
            for s in rng.choice(meas, size=min(k, len(meas)), replace=False):
                for c in rng.choice(6, size=int(rng.integers(1, 3)), replace=False):
                    annot.append((rsid[s], int(chrs[s]), int(bp[s]), a1[s], a2[s], f"GENE{g:03d}", CATEGS[c],
                                  float(np.round(rng.uniform(0.2, 2.0), 3))))
        write_annotation(paths["annot.txt"], annot)
    return dict(paths=paths, bp=bp, rsid=rsid, a1=a1, a2=a2, G=G, af=af, pops=pops, measured=meas, annot=annot)


def make_panel_for_gwas(outdir, pops, gwas_file, n_extra=800, frac_swapped=0.1, seed=7, prefix="cfg"):
    """Synthetic panel (BGZF text index + data, population description) around an EXISTING GWAS summary file:
    every GWAS SNP is in the panel (a fraction with the two alleles the other way round), plus n_extra panel-only
    SNPs at random positions of the same range.  Returns dict(paths=...) like make_synthetic_study."""
    import os
    from . import synth
    rng = np.random.default_rng(seed)
    rows = [l.split() for l in open(gwas_file).read().splitlines()[1:] if l.strip()]
    g_rsid = np.array([r[0] for r in rows]); g_chr = np.array([int(r[1]) for r in rows])
    g_bp = np.array([int(r[2]) for r in rows]); g_a1 = np.array([r[3] for r in rows]); g_a2 = np.array([r[4] for r in rows])
    ebp = rng.choice(np.setdiff1d(np.arange(g_bp.min(), g_bp.max()), g_bp), size=n_extra, replace=False)
    alle = np.array(list("ACGT"))
    e_a1 = alle[rng.integers(0, 4, n_extra)]
    e_a2 = alle[(np.searchsorted(alle, e_a1) + rng.integers(1, 4, n_extra)) % 4]
    swap = rng.random(len(rows)) < frac_swapped
    p_a1 = np.where(swap, g_a2, g_a1); p_a2 = np.where(swap, g_a1, g_a2)
    bp = np.concatenate([g_bp, ebp]); order = np.argsort(bp, kind="stable")
    rsid = np.concatenate([g_rsid, [f"rsE{i}" for i in range(n_extra)]])[order]
    chrs = np.concatenate([g_chr, np.full(n_extra, g_chr[0])])[order]
    a1 = np.concatenate([p_a1, e_a1])[order]; a2 = np.concatenate([p_a2, e_a2])[order]
    bp = bp[order]
    G, af = synth.synth_genotypes(bp, pops, seed=seed + 1)
    paths = {k: os.path.join(outdir, f"{prefix}_{k}") for k in ("desc.txt", "index.gz", "data.gz")}
    paths["gwas.txt"] = gwas_file
    write_pop_desc(paths["desc.txt"], pops)
    write_panel(paths["index.gz"], paths["data.gz"], rsid, chrs, bp, a1, a2, G, af, [p[1] for p in pops])
    return dict(paths=paths, bp=bp, rsid=rsid, pops=pops, n_swapped=int(swap.sum()))


# ------------------------------------------------------------------------------------------
# 2-bit packed genotype rows (include/gauss_hip.h, GAUSS_GENO_2BIT): every population block starts on
# a 16-byte boundary and is zero padded to a multiple of 64 samples; sample s of a block sits at bits
# 2*(s % 4) of byte s // 4.
# ------------------------------------------------------------------------------------------
def pack2bit_layout(pop_sizes):
    """Byte offset of every population block and the row stride (a multiple of 16)."""
    off, o = [], 0
    for m in pop_sizes:
        off.append(o)
        o += ((int(m) + 63) // 64) * 16
    return np.array(off, dtype=np.int32), max(o, 16)


def pack2bit(G, pop_off):
    """G: (S, N) codes 0..2 (or ASCII digits); pop_off: (P+1,) column ranges.  Returns (rows, src_off):
    rows (S, ld) uint8 in the 2-bit layout and the byte offset of each population block."""
    G = np.asarray(G, dtype=np.uint8) & 0x0F
    pop_off = np.asarray(pop_off)
    sizes = np.diff(pop_off)
    src_off, ld = pack2bit_layout(sizes)
    rows = np.zeros((G.shape[0], ld), dtype=np.uint8)
    for q, m in enumerate(sizes):
        blk = np.zeros((G.shape[0], ((int(m) + 63) // 64) * 64), dtype=np.uint8)
        blk[:, :m] = G[:, pop_off[q]:pop_off[q + 1]]
        b4 = blk.reshape(G.shape[0], -1, 4)
        rows[:, src_off[q]:src_off[q] + b4.shape[1]] = b4[:, :, 0] | (b4[:, :, 1] << 2) | (b4[:, :, 2] << 4) | (b4[:, :, 3] << 6)
    return rows, src_off


def unpack2bit(rows, pop_sizes, src_off=None):
    """Inverse of pack2bit for the listed populations: (S, sum(pop_sizes)) codes."""
    rows = np.asarray(rows, dtype=np.uint8)
    if src_off is None:
        src_off, _ = pack2bit_layout(pop_sizes)
    out = []
    for q, m in enumerate(pop_sizes):
        nb = (int(m) + 3) // 4
        b = rows[:, src_off[q]:src_off[q] + nb]
        codes = np.stack([(b >> (2 * k)) & 3 for k in range(4)], axis=2).reshape(rows.shape[0], -1)
        out.append(codes[:, :m])
    return np.concatenate(out, axis=1) if out else np.zeros((rows.shape[0], 0), dtype=np.uint8)


def write_packed_panel(path, pops, rsid, chr_, bp, a1, a2, rows2bit, af, cnt, sorted_flag=None):
    """Write a packed panel file (GAUSSPK1, gauss_amd/csrc/host/packed_panel.h) directly from arrays:
    pops [(name, size, super)], per-SNP strings / positions, rows2bit (S, row_bytes) uint8 in the 2-bit
    layout of pack2bit(), af (S, P) float64 and cnt (S, P) int32.  The C++ converter
    (api.pack_panel) produces the same file from a BGZF text panel."""
    import struct
    S, P = len(rsid), len(pops)
    rows2bit = np.ascontiguousarray(rows2bit, dtype=np.uint8)
    row_bytes = rows2bit.shape[1]
    src_off, ld = pack2bit_layout([q[1] for q in pops])
    assert row_bytes == ld and row_bytes % 16 == 0 and rows2bit.shape[0] == S
    if sorted_flag is None:
        key = np.asarray(chr_, dtype=np.int64) * (1 << 40) + np.asarray(bp, dtype=np.int64)
        sorted_flag = bool(np.all(np.diff(key) >= 0))
    strings = bytearray()
    def add(sv):
        o = len(strings)
        strings.extend(str(sv).encode() + b"\0")
        return o
    snp = np.zeros(S, dtype=[("chr", "<i4"), ("rsid", "<u4"), ("a1", "<u4"), ("a2", "<u4"), ("bp", "<i8")])
    for i in range(S):
        snp[i] = (int(chr_[i]), add(rsid[i]), add(a1[i]), add(a2[i]), int(bp[i]))
    popb = b"".join(struct.pack("<24s24sII", q[0].encode(), q[2].encode(), int(q[1]), int(src_off[k])) for k, q in enumerate(pops))
    al = lambda v, a: (v + a - 1) // a * a
    off = 128
    off_pops = off; off = al(off + len(popb), 64)
    off_snps = off; off = al(off + snp.nbytes, 64)
    off_str = off; off = al(off + len(strings), 64)
    off_af = off; off = al(off + S * P * 8, 64)
    off_cnt = off; off = al(off + S * P * 4, 4096)
    off_geno = off; total = off + S * row_bytes
    hdr = struct.pack("<8sIIQQQQQQQQQI36x", b"GAUSSPK1", 1, P, S, row_bytes, off_pops, off_snps, off_str, off_af, off_cnt,
                      off_geno, total, 1 if sorted_flag else 0)
    assert len(hdr) == 128
    with open(path, "wb") as f:
        f.write(hdr)
        for o, b in ((off_pops, popb), (off_snps, snp.tobytes()), (off_str, bytes(strings)),
                     (off_af, np.ascontiguousarray(af, dtype="<f8").tobytes()),
                     (off_cnt, np.ascontiguousarray(cnt, dtype="<i4").tobytes())):
            f.seek(o)
            f.write(b)
        f.seek(off_geno)
        rows2bit.tofile(f)
        if f.tell() < total:
            f.truncate(total)
    return total
