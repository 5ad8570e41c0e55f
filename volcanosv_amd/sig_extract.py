"""Host side of Large_INDEL/sig_extract.py (the cuteSV-derived read-signature collector behind filter_GT_correction.py:119-127).

The per-read work of parse_read (SE:438-493) runs on the GPU: the CIGAR scan with the script's op table
(VSV_DTYPE_CUTESV) and the in-read merging of generate_combine_sigs (SE:373-435) come back as two tables — raw (one row per
>= min_siglength I/D op) and combined (one row per merged signal: summed length, number of pieces, raw index of the first).
What stays on the host is text: read names, the inserted sequence (sliced out of the read per piece and concatenated,
SE:468-469, 397), the `sort -u | sort -k2,2 -k3,3n` of the final files (SE:637-638)."""
import numpy as np

from .abi import DTYPE_CUTESV, F_REVERSE, F_SECONDARY, F_SKIP, F_SUPP, F_UNMAPPED, M_DEL
from .engine import default_params


def flag_bits(sam_flag, query_length, min_read_len=500):
    """SAM flag + read length -> record flag byte (VSV_F_*); reads shorter than min_read_len are skipped (SE:439)."""
    f = 0
    if sam_flag & 0x10:
        f |= F_REVERSE
    if sam_flag & 0x800:
        f |= F_SUPP
    if sam_flag & 0x100:
        f |= F_SECONDARY
    if sam_flag & 0x4:
        f |= F_UNMAPPED
    if query_length < min_read_len:
        f |= F_SKIP
    return f


def params(min_siglength=10, min_mapq=20, merge_del_threshold=0, merge_ins_threshold=100):
    """vsv_params for the collector; defaults = the script's (SE:703-747)."""
    p = default_params(DTYPE_CUTESV)
    p.min_svlen, p.min_cigar_mapq = int(min_siglength), int(min_mapq)
    p.merge_ins_threshold, p.merge_del_threshold = int(merge_ins_threshold), int(merge_del_threshold)
    return p


def leading_hardclip(soa, rec):
    w = int(soa.cigar[int(soa.cigar_off[rec])])
    return (w >> 4) if (w & 15) == 5 else 0


def cigar_candidates(soa, raw, combined, seq_of, chr_name):
    """Per-record candidate lists in the reference's layout and order (SE:476-477: the read's INS signals, then its DEL
    signals): INS [pos, len, read_name, seq, 'INS', chr], DEL [pos, len, read_name, 'DEL', chr].
    `seq_of(rec)` returns the read's stored sequence (query_sequence)."""
    out = {}
    for row in combined:
        rec = int(row["rec"])
        ins, dele = out.setdefault(rec, ([], []))
        name = soa.qname(rec)
        if int(row["meta"]) & M_DEL:
            dele.append([int(row["pos"]), int(row["svlen"]), name, "DEL", chr_name])
        else:
            seq, hc, left, k = seq_of(rec), leading_hardclip(soa, rec), int(row["q_end"]), int(row["rec2"])
            parts = []
            while left:                                     # pieces = the next `q_end` INS rows of this record in raw order
                r = raw[k]
                if int(r["rec"]) == rec and not (int(r["meta"]) & M_DEL):
                    a = int(r["q_start"]) - hc              # SE:468-469: [shift_ins_read - len - hardclip_left : shift_ins_read - hardclip_left]
                    parts.append(seq[a:a + int(r["svlen"])])       # same Python slice as the reference, same numbers
                    left -= 1
                k += 1
            ins.append([int(row["pos"]), int(row["svlen"]), name, "".join(parts), "INS", chr_name])
    return {rec: a + b for rec, (a, b) in out.items()}


# ---- split-read branch (organize_split_signal SE:341-371 on the host, analysis_split_read SE:193-319 on the GPU) ---------
import ctypes as _C
import re as _re

from .abi import M_QREV, Segments

_CIG = _re.compile(r"(\d+)([MIDNSHP=X])")


def acquire_clip_pos(deal_cigar):
    """SE:329-344: [leading S length, trailing S length, reference span (M, D, =, X)] of an SA-tag CIGAR string."""
    seq = [(int(n), op) for n, op in _CIG.findall(deal_cigar)]
    first_pos = seq[0][0] if seq[0][1] == 'S' else 0
    last_pos = seq[-1][0] if seq[-1][1] == 'S' else 0
    return [first_pos, last_pos, sum(n for n, op in seq if op in "MD=X")]


class SplitSegments:
    """vsv_segments for vsv_cutesv_split: reads = [(record index, query_length, [[read_start, read_end, ref_start, ref_end,
    chrom id, is_reverse], ...])] in organize_split_signal's order (primary first when present)."""

    def __init__(self, reads):
        off, cols = [0], [[] for _ in range(6)]
        for _, _, segs in reads:
            for s in segs:
                for k in range(6):
                    cols[k].append(int(s[k]))
            off.append(len(cols[0]))
        self.read_rec = np.array([r[0] for r in reads], np.uint32)
        self.read_len = np.array([r[1] for r in reads], np.int32)
        self.seg_off = np.array(off, np.uint64)
        self.q_start, self.q_end = np.array(cols[0], np.int32), np.array(cols[1], np.int32)
        self.ref_start, self.ref_end = np.array(cols[2], np.int32), np.array(cols[3], np.int32)
        self.ref_id, self.is_reverse = np.array(cols[4], np.int32), np.array(cols[5], np.uint8)

    def as_struct(self):
        s = Segments()
        s.n_reads, s.n_segs = len(self.read_rec), int(self.q_start.shape[0])
        for k in ("seg_off", "q_start", "q_end", "ref_id", "ref_start", "ref_end", "is_reverse"):
            setattr(s, k, getattr(self, k).ctypes.data_as(_C.c_void_p))
        s.n_tids, s.on_device = 0, 0
        return s


def split_reads(soa, sam_flags, query_lengths, sa_tags, chrom_id, min_mapq=20):
    """The reads parse_read sends to organize_split_signal (SE:479-492): SAM flag exactly 0 or 16 and an SA tag. Returns the
    `reads` list for SplitSegments. chrom_id: name -> integer id (grown on demand for names outside the BAM header)."""
    reads = []
    for i in range(soa.n_records):
        fl = int(sam_flags[i])
        if fl not in (0, 16) or not sa_tags[i]:
            continue
        if int(query_lengths[i]) < 0:
            continue
        a, b = int(soa.cigar_off[i]), int(soa.cigar_off[i + 1])
        ops = soa.cigar[a:b]
        mq = int(soa.mapq[i])
        segs = []
        local_min = min_mapq
        if mq >= min_mapq:                                                      # SE:481-488
            first, last = int(ops[0]), int(ops[-1])
            cl = (first >> 4) if (first & 15) in (4, 5) else 0                  # SE:447-451, 473-478: S, else H overrides
            cr = (last >> 4) if (last & 15) in (4, 5) else 0
            codes, lens = ops & 15, (ops >> 4).astype(np.int64)
            end = int(soa.pos[i]) + int(lens[(codes == 0) | (codes == 2) | (codes == 3) | (codes == 7) | (codes == 8)].sum())
            ql = int(query_lengths[i])
            tid_name = soa.tid_names[int(soa.tid[i])]
            if fl == 0:
                segs.append([cl, ql - cr, int(soa.pos[i]), end, chrom_id(tid_name), 0])
            else:
                segs.append([cr, ql - cl, int(soa.pos[i]), end, chrom_id(tid_name), 1])
            local_min = 0                                                       # SE:347-348
        for ent in sa_tags[i].split(';')[:-1]:                                  # SE:490
            seq = ent.split(',')
            if int(seq[4]) >= local_min:
                first, last, bias = acquire_clip_pos(seq[3])
                start, ql = int(seq[1]), int(query_lengths[i])
                if seq[2] == '+':
                    segs.append([first, ql - last, start, start + bias, chrom_id(seq[0]), 0])
                else:
                    segs.append([last, ql - first, start, start + bias, chrom_id(seq[0]), 1])
        reads.append((i, int(query_lengths[i]), segs))
    return reads


def split_candidates(soa, rows, seq_of, chrom_name):
    """Rows of vsv_cutesv_split -> {record: [candidates]} in the reference's layout: INS [int(pos), len, name, seq, 'INS', chr],
    DEL [pos, len, name, 'DEL', chr]. chrom_name: id -> name."""
    out = {}
    for row in rows:
        rec = int(row["rec"])
        name = soa.qname(rec)
        if int(row["meta"]) & M_DEL:
            c = [int(row["pos"]), int(row["svlen"]), name, "DEL", chrom_name(int(row["tid"]))]
        else:
            q = seq_of(rec)
            if int(row["meta"]) & M_QREV:
                q = q[::-1]
            c = [int(row["pos"]), int(row["svlen"]), name, str(q[int(row["q_start"]):int(row["q_end"])]), "INS", chrom_name(int(row["tid"]))]
        out.setdefault(rec, []).append(c)
    return out
