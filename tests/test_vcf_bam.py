"""Host-side pieces either side of the hot path: BAM ingest (C-ABI, no compute) and VCF text, pinned against the
reference's own write_vcf output stored in the contig_* fixtures (tests/golden/make_golden.py)."""
import os
import zlib

import numpy as np
import pytest

from helpers import load_fixture
from volcanosv_amd import bam, vcf

CONTIG = ["contig_hifi_tiefree", "contig_hifi_stable", "contig_ont_tiefree", "contig_clr_tiefree"]
SEQ_LEN = 410000


def synth_seq(name, n=SEQ_LEN):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return "".join(np.array(list("ACGT"))[rng.integers(0, 4, n)])


def digest(lines):
    rows = []
    for l in lines:
        f = l.rstrip("\n").split("\t")
        rows.append([f[0], int(f[1]), f[2], len(f[3]), zlib.crc32(f[3].encode()), len(f[4]), zlib.crc32(f[4].encode()), f[5], f[6], f[7], f[8], f[9]])
    return rows


def fixture_inputs(doc):
    names = sorted({r[2] for r in doc["records"] if "both" not in r[2]})
    return {n: synth_seq(n) for n in names}


@pytest.mark.parametrize("name", CONTIG)
def test_vcf_lines_match_reference_write_vcf(name):
    from oracle import oracle
    doc, soa, dtype = load_fixture(name)
    st, tabs = oracle.run(soa, dtype=dtype)
    assert st == 0
    dc_contig = fixture_inputs(doc)
    for t, chrom in enumerate(doc["expected"]["chroms"]):
        calls = tabs["calls"][tabs["calls"]["sig"]["tid"] == t]
        lines = vcf.vcf_lines(soa, calls, tabs["merged"], synth_seq(chrom), dc_contig)
        assert digest(lines) == doc["expected"]["per_chrom"][chrom]["vcf"]
        cols = vcf.integer_columns(lines)
        assert all(c[4] is None for c in cols)            # Large_INDEL records carry no END key


def test_default_header_shape():
    h = vcf.default_header()
    assert len(h) == 35 and h[0] == "##fileformat=VCFv4.2\n" and h[13] == "##contig=<ID=chr10,length=135534747>\n"
    assert h[-1].startswith("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t")


def fixture_to_bam(doc, path):
    chroms = doc["expected"]["chroms"]
    refs = [(c, SEQ_LEN) for c in chroms]
    recs = [dict(tid=chroms.index(r[0]), pos=r[1], qname=r[2], mapq=r[3], flag=16 if r[4] else 0, cigar=[tuple(c) for c in r[5]])
            for r in doc["records"]]
    bam.write_bam(path, refs, recs)


def test_bam_ingest_roundtrip(tmp_path):
    doc, soa, _ = load_fixture("contig_hifi_tiefree")
    p = str(tmp_path / "x.bam")
    fixture_to_bam(doc, p)
    got = bam.read_bam(p)
    for k in ("pos", "tid", "qid", "cigar_off", "mapq", "flag", "cigar"):
        assert np.array_equal(getattr(soa, k), getattr(got, k)), k
    assert got.tid_names == doc["expected"]["chroms"]
    one = bam.read_bam(p, "chr2")
    assert one.n_records == int((soa.tid == 1).sum()) and set(one.tid) == {1}
    with pytest.raises(KeyError):
        bam.read_bam(p, "chrZ")


def test_bam_long_cigar_and_sa_tag(tmp_path):
    cig = [(0, 3), (1, 1)] * 40000 + [(0, 5)]             # 80 001 ops > 65535: CG:B,I tag
    recs = [dict(tid=0, pos=7, qname="ctg_hp1_long", mapq=60, flag=0, cigar=cig, seq_len=10),
            dict(tid=0, pos=9, qname="r2", mapq=3, flag=0x810, cigar=[(4, 5), (0, 5)], tags={b"SA": "chr1,8,+,5M5S,60,0;"})]
    p = str(tmp_path / "l.bam")
    bam.write_bam(p, [("chr1", 1000000)], recs)
    s = bam.read_bam(p, "chr1")
    assert s.n_records == 2 and int(s.cigar_off[1]) == len(cig)
    assert [(int(w) & 15, int(w) >> 4) for w in s.cigar[:4]] == cig[:4]
    assert s.sa_tags == ["", "chr1,8,+,5M5S,60,0;"]
    assert s.flag[1] & 1 and s.flag[1] & 2 and s.flag[0] & 4      # reverse, supplementary, hp1


@pytest.mark.gpu
@pytest.mark.parametrize("name,style", [("contig_hifi_tiefree", "Hifi"), ("contig_ont_tiefree", "ONT"), ("contig_clr_tiefree", "CLR")])
def test_cli_plumbing_config1(tmp_path, name, style):
    """BASELINE config 1 plumbing: BAM + contig FASTA + reference FASTA -> extract_contig_signature_<dtype>.py (CLI,
    same flags as the reference) -> volcano_variant_chr<N>.vcf identical to the reference's write_vcf output."""
    import subprocess
    import sys
    doc, soa, _ = load_fixture(name)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    b = str(tmp_path / "contigs.sorted.bam")
    fixture_to_bam(doc, b)
    with open(tmp_path / "contigs.fa", "w") as f:
        for n, s in fixture_inputs(doc).items():
            f.write(">%s\n%s\n" % (n, s))
    with open(tmp_path / "ref.fa", "w") as f:
        for c in doc["expected"]["chroms"]:
            f.write(">%s\n%s\n" % (c, synth_seq(c)))
    hdr = tmp_path / "header"
    hdr.write_text("".join(vcf.default_header()))
    for i, chrom in enumerate(doc["expected"]["chroms"]):
        ref = tmp_path / "ref.fa"
        if style != "Hifi":                      # ONT/CLR take a single-chromosome FASTA (O:653-662)
            ref = tmp_path / ("ref_%s.fa" % chrom)
            ref.write_text(">%s\n%s\n" % (chrom, synth_seq(chrom)))
        cmd = [sys.executable, os.path.join(root, "volcanosv_amd", "cli", "extract_contig_signature_%s.py" % style),
               "-bam", b, "-contig", str(tmp_path / "contigs.fa"), "-header", str(hdr), "-ref", str(ref), "-o", str(tmp_path / "out"),
               "-chr", str(i + 1)]
        subprocess.check_call(cmd)
        lines = open(tmp_path / "out" / ("volcano_variant_chr%d.vcf" % (i + 1))).readlines()
        assert lines[:35] == vcf.default_header()
        assert digest(lines[35:]) == doc["expected"]["per_chrom"][chrom]["vcf"]
        # the per-source signature lists next to the VCF (H:402-405, 459-462): the rows of the fixture's pre-cluster lists are not
        # stored, but their clustered survivors are a subset of the dumped rows (text pinned live in tests/test_reference_live.py)
        pc = doc["expected"]["per_chrom"][chrom]
        for hp in ("hp1", "hp2"):
            for typ, key in (("DEL", "del_cigar"), ("INS", "ins_cigar")):
                dumped = set(open(tmp_path / "out" / "signature" / ("%s_%s_contig_cigar_%s.txt" % (chrom, typ, hp))).read().splitlines())
                for row in pc["cluster1_%s" % hp][key]:
                    assert "\t".join(str(x) for x in row) in dumped


def test_reads_signature_lines_column_wise_equal_row_wise():
    """sigtable.reads_sig_lines (whole table, column by column) writes the text of the per-row field lists (RS:251-265)."""
    from volcanosv_amd import sigtable, synth
    from volcanosv_amd.abi import DTYPE_READS, M_DEL, M_SPLIT, SIG_DTYPE
    t, nq, _ = synth.generate(3000, "hifi", seed=3)
    soa = synth.to_soa(t, nq)
    rng = np.random.default_rng(0)
    n = 5000
    tab = np.zeros(n, SIG_DTYPE)
    tab["pos"] = np.sort(rng.integers(-50, 1 << 27, n)); tab["svlen"] = rng.integers(30, 50000, n)
    tab["rec"] = rng.integers(0, soa.n_records, n); tab["rec2"] = rng.integers(0, soa.n_records, n)
    tab["meta"] = rng.integers(0, 2, n) * M_DEL | (rng.random(n) < 0.3) * M_SPLIT
    tab["q_start"] = rng.integers(0, 20000, n); tab["q_end"] = tab["q_start"] + rng.integers(1, 9000, n)
    tab["tid"] = rng.integers(0, 3, n)
    for names in (None, ["chr7", "chrX", "scaffold_12"]):
        soa.tid_names = names
        want = ["\t".join(str(x) for x in sigtable.sig_fields(soa, s, DTYPE_READS)) + "\n" for s in tab]
        assert sigtable.reads_sig_lines(soa, tab) == want
    assert sigtable.reads_sig_lines(soa, tab[:0]) == []


def test_host_reader_survives_corrupted_files():
    """Semantically corrupted BAM streams (valid BGZF framing; random bytes, length-like fields, deleted stretches inside the
    records, the header or a CG:B,I long CIGAR) either load or raise — the native reader never crashes or reads outside a record.
    The worker runs in its own process so a crash is an exit code."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_corrupt_bam_worker.py")
    for seed in (1, 2, 3):
        r = subprocess.run([sys.executable, worker, str(seed), "120"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (seed, r.returncode, r.stderr[-400:])
        ok, err = int(r.stdout.split()[1]), int(r.stdout.split()[3])
        assert ok + err == 120 and err > 20


@pytest.mark.gpu
def test_cli_reads_signature(tmp_path):
    import subprocess
    import sys
    doc, soa, _ = load_fixture("reads_tiefree")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    chroms = doc["expected"]["chroms"]
    recs = [dict(tid=chroms.index(r[0]), pos=r[1], qname=r[2], mapq=r[3], flag=16 if r[4] else 0, cigar=[tuple(c) for c in r[5]])
            for r in doc["records"]]
    b = str(tmp_path / "reads.bam")
    bam.write_bam(b, [(c, 1000000) for c in chroms], recs)
    for i, chrom in enumerate(chroms):
        subprocess.check_call([sys.executable, os.path.join(root, "volcanosv_amd", "cli", "extract_reads_signature.py"),
                               "-i", b, "-o", str(tmp_path / "out"), "-chr", str(i + 1)])
        got = [l.rstrip("\n").split("\t") for l in open(tmp_path / "out" / "reads_signature" / ("%s_reads_sig.txt" % chrom))]
        want = [[str(x) for x in row] for row in doc["expected"]["per_chrom"][chrom]["merged"]]
        assert got == want
        # the four side files (RS:130-131, 242-243; text pinned live in tests/test_reference_live.py): together they hold the same rows
        side = []
        for fn in ("%s_DEL_reads_cigar.txt", "%s_INS_reads_ciga.txt", "%s_DEL_reads_split.txt", "%s_INS_reads_split.txt"):
            side += [l.rstrip("\n").split("\t") for l in open(tmp_path / "out" / "reads_signature" / (fn % chrom))]
        assert sorted(side) == sorted(want)


def _cigar_for(seg, primary):
    """[tid, rs, re, qa, qb, L, rev] -> CIGAR ops realising exactly these numbers (soft clips around M + I/D)."""
    tid, rs, re_, qa, qb, L, rev = seg
    qlen, rlen = qb - qa, re_ - rs
    ops = []
    if qa > 0:
        ops.append((4, qa))
    m = min(qlen, rlen)
    ops.append((0, m))
    if qlen > rlen:
        ops.append((1, qlen - rlen))
    elif rlen > qlen:
        ops.append((2, rlen - qlen))
    if L - qb > 0:
        ops.append((4, L - qb))
    return ops


@pytest.mark.gpu
def test_cli_svim_bnd_and_filter_tra(tmp_path):
    """Complex_SV plumbing: hp1/hp2 contig BAMs with SA tags -> svim-asm diploid drop-in (BND branch) -> variants.vcf equal to
    the reference svim functions' output; then the filter_tra.py drop-in."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = json.load(open(os.path.join(root, "tests", "golden", "bnd_a.json")))
    contigs = [tuple(c) for c in doc["contigs"]]
    names = [c[0] for c in contigs]
    ops = "MIDNSHP=X"
    for hap in (1, 2):
        recs = []
        for r in doc["reads"]:
            if r["hap"] != hap:
                continue
            p = r["segs"][0]
            sa = ""
            for s in r["segs"][1:]:
                cg = "".join("%d%s" % (l, ops[o]) for o, l in _cigar_for(s, False))
                sa += "%s,%d,%s,%s,60,0;" % (names[s[0]], s[1] + 1, "-" if s[6] else "+", cg)
            recs.append(dict(tid=p[0], pos=p[1], qname=r["name"], mapq=60, flag=16 if p[6] else 0, cigar=_cigar_for(p, True), seq_len=p[5],
                             tags={b"SA": sa}))
        recs.sort(key=lambda x: (x["tid"], x["pos"]))
        bam.write_bam(str(tmp_path / ("assembly_hp%d.bam" % hap)), contigs, recs)
    out = tmp_path / "Raw_Detection"
    subprocess.check_call([sys.executable, os.path.join(root, "volcanosv_amd", "cli", "svim_asm_bnd.py"), "diploid", str(out),
                           str(tmp_path / "assembly_hp1.bam"), str(tmp_path / "assembly_hp2.bam"), "ref.fa", "--query_names"])
    lines = [l.rstrip("\n") for l in open(out / "variants.vcf") if not l.startswith("#")]
    # the BAM route orders reads by coordinate, the fixture by event: READS order inside a 1/1 record is (hp1, hp2) in both;
    # ids follow the natural sort, which is unique here
    assert lines == doc["expected"]["vcf"]
    subprocess.check_call([sys.executable, os.path.join(root, "volcanosv_amd", "cli", "filter_tra.py"), "-vcf", str(out / "variants.vcf"),
                           "-o", str(tmp_path / "TRA"), "-bam", "x.bam"])
    tra = [l for l in open(tmp_path / "TRA" / "TRA_final.vcf") if not l.startswith("#")]
    assert 0 < len(tra) <= len(lines)


@pytest.mark.gpu
def test_all_chromosomes_parse_once_on_the_device(tmp_path):
    """contig_signature.run without -chr (chr1..chr22, H:729-742): the device reader inflates and parses the file once and every
    chromosome runs on its record range; the VCFs equal those of the host reader, chromosome by chromosome."""
    from volcanosv_amd import contig_signature
    doc, soa, _ = load_fixture("contig_hifi_tiefree")
    chroms = doc["expected"]["chroms"]
    b = str(tmp_path / "contigs.sorted.bam")
    recs = [dict(tid=chroms.index(r[0]), pos=r[1], qname=r[2], mapq=r[3], flag=16 if r[4] else 0, cigar=[tuple(c) for c in r[5]])
            for r in doc["records"]]
    bam.write_bam(b, [("chr%d" % i, 1000000) for i in range(1, 23)], recs)
    with open(tmp_path / "contigs.fa", "w") as f:
        for n, s in fixture_inputs(doc).items():
            f.write(">%s\n%s\n" % (n, s))
    with open(tmp_path / "ref.fa", "w") as f:
        for i in range(1, 23):
            f.write(">chr%d\n%s\n" % (i, synth_seq("chr%d" % i)))
    outs = {}
    for mode in (True, False):
        outs[mode] = contig_signature.run("Hifi", b, str(tmp_path / "contigs.fa"), str(tmp_path / "ref.fa"), str(tmp_path / ("out%d" % mode)),
                                          chr_number=None, device_ingest=mode, log=lambda *a: None)
    assert outs[True] == outs[False] and sum(len(v) for v in outs[True].values()) > 10
    for c in chroms:
        assert digest(outs[True][c]) == doc["expected"]["per_chrom"][c]["vcf"]
