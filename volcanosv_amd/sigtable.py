"""Host-side views of the C-ABI tables in the reference's own list-of-fields form.

The reference's only data type on this path is a Python list of 10 fields
(extract_contig_signature_Hifi.py:80,84): [chrom, 'DEL'|'INS', pos, svlen, qname, q_start, q_end,
'+'|'-', 'cigar'|'split-alignment', mapq | "m1-m2"], extended by pair_sig (H:573-592) with
[GT, contig_info, strands, sources, mapqs]. The C-ABI carries integers only (include/volcanosv.h
`vsv_sig`, `vsv_call`); these helpers re-attach the strings from the caller's SoA.
"""
from .abi import DTYPE_READS, M_DEL, M_HP2, M_SPLIT


def _chrom(soa, tid):
    if soa.tid_names is not None:
        return soa.tid_names[int(tid)]
    return "chr%d" % (int(tid) + 1)


def sig_fields(soa, s, dtype=None):
    """One vsv_sig row -> the reference's field list (10 fields; READS: 8/9 fields, RS:72-76,191-195)."""
    chrom = _chrom(soa, s["tid"])
    is_del = bool(s["meta"] & M_DEL)
    typ = "DEL" if is_del else "INS"
    rec = int(s["rec"])
    qname, strand = soa.qname(rec), soa.strand(rec)
    if dtype == DTYPE_READS:
        if s["meta"] & M_SPLIT:
            # the reference's own spellings (RS:191, RS:194)
            return [chrom, typ, int(s["pos"]), int(s["svlen"]), qname, int(s["q_start"]), int(s["q_end"]), strand,
                    "split-alignemnt" if is_del else "split_alignment"]
        return [chrom, typ, int(s["pos"]), int(s["svlen"]), qname, int(s["q_start"]), strand, "cigar"]
    if s["meta"] & M_SPLIT:
        src = "split-alignment"
        mapq = "%d-%d" % (int(soa.mapq[rec]), int(soa.mapq[int(s["rec2"])]))
    else:
        src = "cigar"
        mapq = int(soa.mapq[rec])
    return [chrom, typ, int(s["pos"]), int(s["svlen"]), qname, int(s["q_start"]), int(s["q_end"]), strand, src, mapq]


def reads_sig_lines(soa, table):
    """The text lines of chr<N>_reads_sig.txt (RS:251-265: tab-joined str() of the READS field lists) for a whole table,
    built column by column: the same text as '\\t'.join(str(x) for x in sig_fields(soa, s, DTYPE_READS)) per row, at
    about a sixth of the cost (the table of one chromosome of a 30x read set has 10^5-10^6 rows)."""
    if len(table) == 0:
        return []
    import numpy as np
    rec = table["rec"].astype(np.int64)
    qid = np.asarray(soa.qid)[rec].tolist()
    qn = soa.qnames
    names = [qn[q] for q in qid] if qn is not None else ["q%d" % q for q in qid]
    rev = (np.asarray(soa.flag)[rec] & 1).tolist()                      # VSV_F_REVERSE
    tids = table["tid"].tolist()
    if soa.tid_names is not None:
        tn = soa.tid_names
        chrom = [tn[t] for t in tids]
    else:
        chrom = ["chr%d" % (t + 1) for t in tids]
    meta = table["meta"]
    is_del = ((meta & M_DEL) != 0).tolist()
    is_split = ((meta & M_SPLIT) != 0).tolist()
    pos, svlen, qs, qe = table["pos"].tolist(), table["svlen"].tolist(), table["q_start"].tolist(), table["q_end"].tolist()
    out = []
    for i in range(len(pos)):
        strand = "-" if rev[i] else "+"
        if is_split[i]:       # the reference's own spellings (RS:191, RS:194)
            out.append("%s\t%s\t%d\t%d\t%s\t%d\t%d\t%s\t%s\n" % (chrom[i], "DEL" if is_del[i] else "INS", pos[i], svlen[i], names[i], qs[i], qe[i],
                                                             strand, "split-alignemnt" if is_del[i] else "split_alignment"))
        else:
            out.append("%s\t%s\t%d\t%d\t%s\t%d\t%s\tcigar\n" % (chrom[i], "DEL" if is_del[i] else "INS", pos[i], svlen[i], names[i], qs[i], strand))
    return out


def call_fields(soa, c, merged):
    """One vsv_call row -> the 15-field paired signature of pair_sig (H:571-592)."""
    base = sig_fields(soa, c["sig"])
    a, b = int(c["a"]), int(c["b"])

    def info(s):
        f = sig_fields(soa, s)
        return "%s:%d-%d" % (f[4], f[5], f[6]), f[7], f[8], str(f[9])

    if a >= 0 and b >= 0:
        i1, i2 = info(merged[a]), info(merged[b])
        return base + ["1/1", i1[0] + "," + i2[0], i1[1] + "," + i2[1], i1[2] + "," + i2[2], i1[3] + "," + i2[3]]
    i1 = info(merged[a] if a >= 0 else merged[b])
    return base + ["0/1", i1[0], i1[1], i1[2], i1[3]]


def hap_of(s):
    return "hp2" if s["meta"] & M_HP2 else "hp1"


def signature_dump_texts(soa, cigar_table, split_table, chrom, tid=None):
    """The four per-haplotype dump files extract_signature_one_hap leaves in <out>/signature/ (H:402-405 write_sig_cigar, H:459-462
    write_sig_split): for each haplotype and type the pre-cluster list, sorted by position (stable), one tab-joined 10-field row per
    signature. Returns {file name: text} for `chrom`. cigar_table / split_table: the engine's "cigar" / "split" tables."""
    import numpy as np
    out = {}
    for src_name, table in (("cigar", cigar_table), ("split", split_table)):
        if tid is not None and len(table):
            table = table[table["tid"] == tid]
        for hp, want_hp2 in (("hp1", False), ("hp2", True)):
            for typ, want_del in (("DEL", True), ("INS", False)):
                m = (((table["meta"] & M_HP2) != 0) == want_hp2) & (((table["meta"] & M_DEL) != 0) == want_del)
                rows = table[m]
                rows = rows[np.argsort(rows["pos"], kind="stable")]
                text = "".join("\t".join(str(x) for x in sig_fields(soa, r)) + "\n" for r in rows)
                out["%s_%s_contig_%s_%s.txt" % (chrom, typ, src_name, hp)] = text
    return out


def reads_dump_texts(soa, cigar_table, split_table, chrom, tid=None):
    """The four side files of extract_reads_signature.py (RS:130-131, 242-243; the INS cigar file really is spelled `_reads_ciga.txt`
    there): the CIGAR and split signature lists per type, sorted by position (stable), tab-joined str() fields."""
    import numpy as np
    out = {}
    for (table, names) in ((cigar_table, {True: "%s_DEL_reads_cigar.txt", False: "%s_INS_reads_ciga.txt"}),
                           (split_table, {True: "%s_DEL_reads_split.txt", False: "%s_INS_reads_split.txt"})):
        if tid is not None and len(table):
            table = table[table["tid"] == tid]
        for want_del, pattern in names.items():
            rows = table[((table["meta"] & M_DEL) != 0) == want_del]
            rows = rows[np.argsort(rows["pos"], kind="stable")]
            out[pattern % chrom] = "".join("\t".join(str(x) for x in sig_fields(soa, r, DTYPE_READS)) + "\n" for r in rows)
    return out
