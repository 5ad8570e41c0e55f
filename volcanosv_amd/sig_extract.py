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
