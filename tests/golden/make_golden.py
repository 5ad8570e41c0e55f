#!/usr/bin/env python3
"""Generates tests/golden/*.json from the REFERENCE's own functions (build container only).

The reference scripts cannot be imported as modules (argparse + main loop at import time, `import pysam`),
so — as SURVEY.md §8c describes — the top-level FunctionDefs of each script are extracted with `ast`,
compiled, and executed against duck-typed read objects behind a fake `pysam.AlignmentFile`. Nothing of
the reference's text is written to the repo: the fixtures hold only INPUT records and the OUTPUT
signature/call tables the reference functions returned.

Tie rule: `sort_sig` (extract_contig_signature_Hifi.py:170-179) uses numpy's default (unstable) argsort.
Fixtures named *_tiefree were produced with the unmodified numpy and contain no equal positions in any
sorted list (checked at generation time); fixtures named *_stable were produced with an injected numpy
proxy whose argsort forces kind='stable' — the one declared deviation (SURVEY.md §7 hard part 1).

Usage:  python tests/golden/make_golden.py            (needs /root/reference)
"""
import ast
import json
import os
import sys
import tempfile
import types
from collections import Counter

import numpy as np

REF = os.environ.get("VSV_REFERENCE", "/root/reference")
LI = os.path.join(REF, "bin/VolcanoSV-vc/Large_INDEL")
HERE = os.path.dirname(os.path.abspath(__file__))

SCRIPTS = {
    "Hifi": "extract_contig_signature_Hifi.py",
    "ONT": "extract_contig_signature_ONT.py",
    "CLR": "extract_contig_signature_CLR.py",
    "READS": "extract_reads_signature.py",
}


# ------------------------------------------------------------------------------------------------
# fake pysam + duck-typed reads
# ------------------------------------------------------------------------------------------------
class FakeRead:
    def __init__(self, chrom, pos, qname, mapq, reverse, cigar, seq_len=None):
        self.reference_name = chrom
        self.pos = pos
        self.qname = qname
        self.mapq = mapq
        self.is_reverse = bool(reverse)
        self.cigar = [tuple(c) for c in cigar]
        # pysam reference_end: M, D, N, =, X consume the reference
        self.reference_end = pos + sum(l for op, l in self.cigar if op in (0, 2, 3, 7, 8))
        self.seq = None if seq_len is None else "N" * int(seq_len)      # `if read.seq: assert len(read.seq)==offset_contig` (H:397-398)


class FakeAlignmentFile:
    store = {}

    def __init__(self, path, *a, **k):
        self.path = path

    def fetch(self, chrom):
        return iter(FakeAlignmentFile.store.get(chrom, []))


class StableNumpy:
    """numpy proxy: argsort forced to kind='stable' (declared deviation for tie-rich fixtures)."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def argsort(a, *args, **kw):
        kw["kind"] = "stable"
        return np.argsort(a, *args, **kw)


class _Logger:
    def info(self, *a, **k):
        pass


def load_functions(script, stable, tie_log):
    src = open(os.path.join(LI, script)).read()
    tree = ast.parse(src)
    fdefs = [n for n in tree.body if isinstance(n, ast.FunctionDef)]
    mod = ast.Module(body=fdefs, type_ignores=[])
    ns = {
        "np": StableNumpy() if stable else np,
        "Counter": Counter,
        "tqdm": lambda x, **k: x,
        "logger": _Logger(),
        "pysam": types.SimpleNamespace(AlignmentFile=FakeAlignmentFile),
        "os": os,
        "min_cigar_mapq": 50,
        "min_split_mapq": 50,
    }
    exec(compile(mod, script, "exec"), ns)
    # record whether any sort_sig call saw equal positions (to certify tie-free fixtures)
    orig_sort = ns["sort_sig"]

    def sort_sig_logged(sig_list):
        pl = [s[2] for s in sig_list]
        if len(set(pl)) != len(pl):
            tie_log.append(1)
        return orig_sort(sig_list)

    ns["sort_sig"] = sort_sig_logged
    return ns


# ------------------------------------------------------------------------------------------------
# synthetic contig-like / read-like records
# ------------------------------------------------------------------------------------------------
def make_contig_records(seed, n_sites=40, n_chrom=2, contigs_per_hap=3, split_pairs=12, style="Hifi",
                        tie_rich=False):
    """Contig-vs-reference shaped records: per chromosome and hap a few overlapping contigs whose CIGARs
    carry jittered copies of shared INS/DEL sites (so clustering and hp1/hp2 pairing both trigger),
    close INS pairs (intra-read fold), low-mapq and no-hp records, and clipped split pairs."""
    rng = np.random.default_rng(seed)
    recs = []
    for c in range(n_chrom):
        chrom = "chr%d" % (c + 1)
        L = 400000
        sites = []
        p = 5000
        for _ in range(n_sites):
            p += int(rng.integers(150, 9000))
            typ = int(rng.integers(1, 3))  # 1 INS, 2 DEL
            ln = int(rng.choice([35, 60, 120, 260, 330, 800, 2500]))
            sites.append((p, typ, ln))
        for hp in (1, 2):
            for k in range(contigs_per_hap):
                start = int(rng.integers(0, 3000)) + k * 500
                end = L - int(rng.integers(0, 3000))
                qname = "PS%d_hp%d_ctg%d" % (100 + c, hp, k)
                if tie_rich:
                    jit = lambda: 0
                    ljit = lambda l: l
                else:
                    jit = lambda: int(rng.integers(-130, 131))
                    ljit = lambda l: max(30, int(l * rng.uniform(0.4, 1.7)))
                cig = []
                if rng.random() < 0.4:
                    cig.append((5, int(rng.integers(10, 500))))
                elif rng.random() < 0.5:
                    cig.append((4, int(rng.integers(10, 500))))
                cur = start
                for (sp, typ, ln) in sites:
                    if rng.random() < 0.25:
                        continue  # this contig lacks the site
                    ep = sp + jit()
                    if ep <= cur + 20:
                        continue
                    # small indel noise before the event
                    m = ep - cur
                    if m > 200 and rng.random() < 0.5:
                        a = int(rng.integers(20, m - 100))
                        cig.append((0, a)); cig.append((int(rng.integers(1, 3)), int(rng.integers(1, 29))))
                        if cig[-1][0] == 2:
                            a += cig[-1][1]
                        m = ep - cur - a
                        if m <= 0:
                            m = 1
                    cig.append((0, m))
                    l2 = ljit(ln)
                    cig.append((typ, l2))
                    cur = ep + (l2 if typ == 2 else 0)
                    # close second INS to trigger the intra-read fold (H:108-138)
                    if typ == 1 and l2 > 100 and rng.random() < 0.5:
                        gap = int(rng.integers(5, 420))
                        cig.append((0, gap)); cig.append((1, int(rng.integers(90, 500))))
                        cur += gap
                cig.append((0, max(1, end - cur)))
                if rng.random() < 0.5:
                    cig.append((4, int(rng.integers(5, 300))))
                mapq = 60 if rng.random() < 0.85 else int(rng.integers(0, 50))
                recs.append((chrom, start, qname, mapq, bool(rng.random() < 0.5), cig))
            # a record without hp tag (ignored by the contig path) and one with both tags
        recs.append((chrom, 100, "PS_unphased_%d" % c, 60, False, [(0, 1000), (2, 80), (0, 1000)]))
        recs.append((chrom, 7000, "PSX_hp1_hp2_both%d" % c, 60, False, [(0, 500), (1, 90), (0, 700), (2, 45), (0, 30)]))
        # noisy contigs that fail the CLR gate (C:53-70: ins_pct > 0.13 and mean M length < 200), one that
        # passes on ins_pct alone and one that passes on var_dist alone
        recs.append((chrom, 52000, "PS%d_hp1_noisy_fail" % (300 + c), 60, False,
                     [(0, 100), (1, 40), (0, 100), (1, 50), (0, 120), (2, 60), (0, 90)]))
        recs.append((chrom, 61000, "PS%d_hp2_noisy_pct_ok" % (300 + c), 60, True,
                     [(0, 150), (1, 39), (0, 150), (2, 70), (0, 150)]))
        recs.append((chrom, 73000, "PS%d_hp1_noisy_dist_ok" % (300 + c), 60, False,
                     [(0, 250), (1, 100), (0, 250), (2, 33), (0, 100)]))
        # split pairs: same qname, rec1 ends with clip, rec2 starts with clip, equal read length
        for k in range(split_pairs):
            hp = 1 + (k & 1)
            qname = "PS%d_hp%d_split%d" % (200 + c, hp, k)
            rl = int(rng.integers(20000, 60000))
            a_len = int(rng.integers(3000, rl - 3000))
            pos1 = int(rng.integers(10000, L - 100000))
            rev = bool(rng.random() < 0.5)
            kind = int(rng.integers(0, 4))
            if kind == 0:      # DEL-like: reference gap, small read overlap/gap
                olp = int(rng.integers(-60, 61)); dd = int(rng.integers(30, 9000))
            elif kind == 1:    # INS-like: read gap
                olp = int(rng.integers(-600, 601)); dd = -int(rng.integers(30, 9000))
            elif kind == 2:    # threshold probes around r*Diffdis (ONT/CLR float products)
                dd = int(rng.choice([30, 31, 33, 37, 100, 101, 333, 1001, 3333]))
                olp = int(round(dd * rng.choice([0.3, 0.5, -0.3, -0.5]))) + int(rng.integers(-1, 2))
            else:
                dd = -int(rng.choice([30, 31, 33, 37, 100, 101, 333, 1001, 3333]))
                olp = int(round(dd * rng.choice([0.3, 0.5, 0.8, -0.3, -0.5]))) + int(rng.integers(-1, 2))
            # rec1: a_len M then clip (rl - a_len); Read1e = a_len. rec2: clip c2 then M. Read2s = c2
            if dd >= 30:
                c2 = a_len - olp        # Diffolp = Read1e - Read2s = olp
                ref_gap = dd + (c2 - a_len)  # Diffdis = (Ref2s-Ref1e) - (Read2s-Read1e)
            else:
                ref_gap = -olp          # Diffolp = Ref1e - Ref2s = olp
                c2 = a_len + (ref_gap - dd)
            if c2 <= 0 or c2 >= rl - 10:
                continue
            ref1e = pos1 + a_len
            pos2 = ref1e + ref_gap
            if pos2 < pos1:
                continue
            clip1 = 4 if rng.random() < 0.5 else 5
            clip2 = 4 if rng.random() < 0.5 else 5
            mq1 = 60 if rng.random() < 0.9 else 20
            recs.append((chrom, pos1, qname, mq1, rev, [(0, a_len), (clip1, rl - a_len)]))
            recs.append((chrom, pos2, qname, 60, rev if rng.random() < 0.9 else (not rev), [(clip2, c2), (0, rl - c2)]))
            if rng.random() < 0.3:  # third segment of the same name
                c3 = c2 + 500
                if c3 < rl - 10:
                    recs.append((chrom, pos2 + 20000, qname, 60, rev, [(4, c3), (0, rl - c3)]))
    chrom_order = {"chr%d" % (i + 1): i for i in range(n_chrom)}
    recs.sort(key=lambda r: (chrom_order[r[0]], r[1]))
    return recs


def make_read_records(seed, n=600, n_chrom=2, tie_rich=False):
    """Read-shaped records for the RS path: ops {0,1,2,3,4,5,7,8}, some >=30 bp events, split pairs."""
    rng = np.random.default_rng(seed)
    recs = []
    for c in range(n_chrom):
        chrom = "chr%d" % (c + 1)
        for i in range(n):
            pos = int(rng.integers(0, 40)) * 5000 if tie_rich else int(rng.integers(0, 200000))
            k = int(rng.poisson(6))
            cig = []
            if rng.random() < 0.2:
                cig.append((5, int(rng.integers(1, 200))))
            if rng.random() < 0.3:
                cig.append((4, int(rng.integers(1, 200))))
            for j in range(k):
                cig.append((int(rng.choice([0, 0, 7, 8])), 500 if tie_rich else int(rng.integers(20, 900))))
                op = int(rng.choice([1, 2, 2, 1, 3]))
                ln = int(rng.integers(1, 8)) if rng.random() < 0.85 else int(rng.integers(30, 4000))
                cig.append((op, ln))
            cig.append((0, int(rng.integers(20, 900))))
            if rng.random() < 0.3:
                cig.append((4, int(rng.integers(1, 200))))
            mapq = 60 if rng.random() < 0.8 else int(rng.integers(0, 50))
            recs.append((chrom, pos, "read%d_%d" % (c, i), mapq, bool(rng.random() < 0.5), cig))
        for k in range(40):
            qname = "split%d_%d" % (c, k)
            rl = int(rng.integers(5000, 30000))
            a_len = int(rng.integers(1000, rl - 1000))
            pos1 = int(rng.integers(1000, 150000))
            rev = bool(rng.random() < 0.5)
            dd = int(rng.integers(-6000, 6000))
            olp = int(rng.integers(-100, 60))
            ref_gap = -olp
            c2 = a_len + (ref_gap - dd)
            if c2 <= 0 or c2 >= rl - 10 or ref_gap + a_len < 0:
                continue
            recs.append((chrom, pos1, qname, int(rng.integers(0, 61)), rev, [(7, a_len), (4, rl - a_len)]))
            recs.append((chrom, pos1 + a_len + ref_gap, qname, int(rng.integers(0, 61)), rev, [(5, c2), (0, rl - c2)]))
    chrom_order = {"chr%d" % (i + 1): i for i in range(n_chrom)}
    recs.sort(key=lambda r: (chrom_order[r[0]], r[1]))
    return recs


# ------------------------------------------------------------------------------------------------
# run the reference functions
# ------------------------------------------------------------------------------------------------
def jsonable(x):
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    return x


def synth_seq(name, n):
    """Deterministic pseudo-random ACGT string for a contig / chromosome name (tests regenerate it from the name)."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return "".join(np.array(list("ACGT"))[rng.integers(0, 4, n)])


def vcf_digest(lines):
    """VCF record lines -> compact rows: REF/ALT replaced by (length, crc32)."""
    import zlib
    rows = []
    for l in lines:
        if l.startswith("#"):
            continue
        f = l.rstrip("\n").split("\t")
        rows.append([f[0], int(f[1]), f[2], len(f[3]), zlib.crc32(f[3].encode()), len(f[4]), zlib.crc32(f[4].encode()), f[5], f[6], f[7], f[8], f[9]])
    return rows


SEQ_LEN = 410000


def run_contig(style, recs, stable, dumps=False):
    tie_log = []
    ns = load_functions(SCRIPTS[style], stable, tie_log)
    chroms = []
    for r in recs:
        if r[0] not in chroms:
            chroms.append(r[0])
    FakeAlignmentFile.store = {c: [FakeRead(*r) for r in recs if r[0] == c] for c in chroms}
    out = {"chroms": chroms, "per_chrom": {}}
    split_raw = []
    orig_split = ns["extract_sig_from_split"]

    def logged_split(read1, read2, min_mapq, max_svlen):
        d, i = orig_split(read1, read2, min_mapq, max_svlen)
        split_raw.append([d, i])
        return d, i

    ns["extract_sig_from_split"] = logged_split
    with tempfile.TemporaryDirectory() as td:
        for c in chroms:
            pc = {}
            # per-record fold outputs of extract_sig_from_cigar (all records, no filter): [del_sig, ins_sig, off_ref, off_contig]
            pc["per_record"] = [jsonable(list(ns["extract_sig_from_cigar"](rd, 30))) for rd in FakeAlignmentFile.store[c]]
            finals = {}
            for hp in ("hp1", "hp2"):
                del split_raw[:]
                dc, ic = ns["extract_signature_from_cigar"]("x.bam", c, td, hp, 50)
                ds, is_ = ns["extract_sig_from_split_reads"]("x.bam", c, td, hp, 50)
                pc["cluster1_%s" % hp] = jsonable({"del_cigar": dc, "ins_cigar": ic, "del_split": ds, "ins_split": is_})
                pc["split_raw_%s" % hp] = jsonable(split_raw)
                finals[hp] = ns["merge_all"](dc, ic, ds, is_)
                if dumps:   # the files write_sig_cigar / write_sig_split left in the output directory (H:404-405, 461-462)
                    for typ in ("DEL", "INS"):
                        for src in ("cigar", "split"):
                            fn = "%s_%s_contig_%s_%s.txt" % (c, typ, src, hp)
                            pc.setdefault("dumps", {})[fn] = open(os.path.join(td, fn)).read()
                pc["merged_%s" % hp] = jsonable(finals[hp])
            paired = ns["pair_sig"](finals["hp1"], finals["hp2"], 1000, 200, 0.5, 0.5)
            pc["paired"] = jsonable(paired)
            # write_vcf (H:678-714) with deterministic sequences; contigs named *_both* are absent from the FASTA (H:648)
            names = sorted({r[2] for r in recs if "both" not in r[2]})
            ns["dc_contig"] = {n: synth_seq(n, SEQ_LEN) for n in names}
            ns["ref_seq"] = synth_seq(c, SEQ_LEN)
            hdr = os.path.join(td, "header")
            with open(hdr, "w") as fh:
                fh.write("##fileformat=VCFv4.2\n")
            ns["header_path"] = hdr
            if style != "Hifi":   # ONT/CLR load the chromosome with load_seq(ref_path) (O:662,704)
                ns["load_seq"] = lambda path: ns["ref_seq"]
            vp = os.path.join(td, "out_%s.vcf" % c)
            ns["write_vcf"](paired, vp, "ref.fa", "contig.fa")
            pc["vcf"] = vcf_digest(open(vp).readlines())
            out["per_chrom"][c] = pc
    out["had_ties"] = bool(tie_log)
    return out


def run_reads(recs, stable, dumps=False):
    tie_log = []
    ns = load_functions(SCRIPTS["READS"], stable, tie_log)
    chroms = []
    for r in recs:
        if r[0] not in chroms:
            chroms.append(r[0])
    FakeAlignmentFile.store = {c: [FakeRead(*r) for r in recs if r[0] == c] for c in chroms}
    out = {"chroms": chroms, "per_chrom": {}}
    with tempfile.TemporaryDirectory() as td:
        for c in chroms:
            dcs, ics = ns["extract_signature_from_cigar"]("x.bam", c, td)
            dss, iss = ns["extract_sig_from_split_reads"]("x.bam", c, td)
            merged = ns["merge_all"](dcs + ics + dss + iss, td, c)
            out["per_chrom"][c] = {"merged": jsonable(merged)}
            if dumps:   # the four side files (RS:130-131, 242-243)
                out["per_chrom"][c]["dumps"] = {fn % c: open(os.path.join(td, fn % c)).read() for fn in
                                                ("%s_DEL_reads_cigar.txt", "%s_INS_reads_ciga.txt", "%s_DEL_reads_split.txt", "%s_INS_reads_split.txt")}
    out["had_ties"] = bool(tie_log)
    return out


def write_fixture(name, dtype, recs, expected, stable):
    path = os.path.join(HERE, name + ".json")
    doc = {
        "dtype": dtype,
        "numpy_argsort": "stable-shim" if stable else "unmodified",
        "had_ties": expected.pop("had_ties"),
        "generator": "tests/golden/make_golden.py",
        "records": jsonable(recs),
        "expected": expected,
    }
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote %s (%d records, %d bytes, ties=%s)" % (path, len(recs), os.path.getsize(path), doc["had_ties"]))
    return doc


def main():
    if not os.path.isdir(LI):
        sys.exit("reference not found at %s" % LI)
    for style in ("Hifi", "ONT", "CLR"):
        # tie-free: unmodified numpy; search a seed whose sorted lists never hold equal positions
        for seed in range(1000, 1100):
            recs = make_contig_records(seed, style=style)
            exp = run_contig(style, recs, stable=False)
            if not exp["had_ties"]:
                write_fixture("contig_%s_tiefree" % style.lower(), style, recs, exp, stable=False)
                break
        else:
            sys.exit("no tie-free seed found for %s" % style)
        recs = make_contig_records(77, style=style, tie_rich=True)
        exp = run_contig(style, recs, stable=True)
        assert exp["had_ties"]
        write_fixture("contig_%s_stable" % style.lower(), style, recs, exp, stable=True)
    for seed in range(2000, 2100):
        recs = make_read_records(seed, n=300)
        exp = run_reads(recs, stable=False)
        if not exp["had_ties"]:
            write_fixture("reads_tiefree", "READS", recs, exp, stable=False)
            break
    else:
        print("no tie-free reads seed; only the stable fixture is written")
    recs = make_read_records(5, n=600, tie_rich=True)
    exp = run_reads(recs, stable=True)
    assert exp["had_ties"]
    write_fixture("reads_stable", "READS", recs, exp, stable=True)


if __name__ == "__main__":
    main()
