"""Host side of Large_INDEL/sig_extract.py (the cuteSV-derived read-signature collector behind filter_GT_correction.py:119-127).

The per-read work of parse_read (SE:438-493) runs on the GPU: the CIGAR scan with the script's op table
(VSV_DTYPE_CUTESV) and the in-read merging of generate_combine_sigs (SE:373-435) come back as two tables — raw (one row per
>= min_siglength I/D op) and combined (one row per merged signal: summed length, number of pieces, raw index of the first).
What stays on the host is text: read names, the inserted sequence (sliced out of the read per piece and concatenated,
SE:468-469, 397), the `sort -u | sort -k2,2 -k3,3n` of the final files (SE:637-638)."""
import ctypes as _C
import re as _re

import numpy as np

from .abi import DTYPE_CUTESV, F_REVERSE, F_SECONDARY, F_SKIP, F_SUPP, F_UNMAPPED, M_DEL, M_QREV, Segments
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


class DeferredSeq:
    """Slices of the reads' sequences, asked for while the candidates are built and fetched together afterwards: with the device
    reader the sequences stay on the GPU and one call decodes every slice (DeviceRecordView.seq_slices); with the host reader
    they come out of the packed host copy. ask() returns a token, text(tokens) the concatenated slices after resolve()."""

    def __init__(self, fetch):
        self.fetch, self.req, self.out = fetch, [], None

    def ask(self, rec, start, stop, reverse=False):
        self.req.append((int(rec), int(start), int(stop), bool(reverse)))
        return len(self.req) - 1

    def resolve(self):
        self.out = self.fetch(self.req)

    def text(self, tokens):
        return "".join(self.out[t] for t in tokens)


def host_seq_fetch(seq_of):
    """fetch function of DeferredSeq over sequences on the host: seq_of(rec) -> the read's query_sequence (or an indexable of them,
    e.g. the PackedSeq of BamFile.fetch_soa(keep_seq=True))."""
    get = seq_of if callable(seq_of) else seq_of.__getitem__

    def fetch(reqs):
        cache, out = {}, []
        for rec, a, b, rv in reqs:
            if rec not in cache:
                cache[rec] = get(rec)
            q = cache[rec]
            out.append(str((q[::-1] if rv else q)[a:b]))
        return out
    return fetch


def _own_seqs(seqs):
    """(DeferredSeq, created here?) — the candidate builders also take a plain seq_of(rec) callable and then resolve at once."""
    return (seqs, False) if isinstance(seqs, DeferredSeq) else (DeferredSeq(host_seq_fetch(seqs)), True)


def cigar_candidates(soa, raw, combined, seqs, chr_name):
    """Per-record candidate lists in the reference's layout and order (SE:476-477: the read's INS signals, then its DEL
    signals): INS [pos, len, read_name, seq, 'INS', chr], DEL [pos, len, read_name, 'DEL', chr].
    `seqs` = DeferredSeq over the reads' stored sequences (query_sequence); the INS entries carry slice tokens until
    resolve_sequences()."""
    seqs, own = _own_seqs(seqs)
    out = {}
    for row in combined:
        rec = int(row["rec"])
        ins, dele = out.setdefault(rec, ([], []))
        name = soa.qname(rec)
        if int(row["meta"]) & M_DEL:
            dele.append([int(row["pos"]), int(row["svlen"]), name, "DEL", chr_name])
        else:
            hc, left, k = leading_hardclip(soa, rec), int(row["q_end"]), int(row["rec2"])
            parts = []
            while left:                                     # pieces = the next `q_end` INS rows of this record in raw order
                r = raw[k]
                if int(r["rec"]) == rec and not (int(r["meta"]) & M_DEL):
                    a = int(r["q_start"]) - hc              # SE:468-469: [shift_ins_read - len - hardclip_left : shift_ins_read - hardclip_left]
                    parts.append(seqs.ask(rec, a, a + int(r["svlen"])))       # same Python slice as the reference, same numbers
                    left -= 1
                k += 1
            ins.append([int(row["pos"]), int(row["svlen"]), name, parts, "INS", chr_name])
    res = {rec: a + b for rec, (a, b) in out.items()}
    if own:
        resolve_sequences(seqs, res)
    return res


# ---- split-read branch (organize_split_signal SE:341-371 on the host, analysis_split_read SE:193-319 on the GPU) ---------
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


def resolve_sequences(seqs, *candidate_maps):
    """Fetches every slice asked for and replaces the token lists of the INS candidates by their text."""
    seqs.resolve()
    for cands in candidate_maps:
        for lst in cands.values():
            for c in lst:
                if len(c) == 6 and isinstance(c[3], list):
                    c[3] = seqs.text(c[3])


def split_candidates(soa, rows, seqs, chrom_name):
    """Rows of vsv_cutesv_split -> {record: [candidates]} in the reference's layout: INS [int(pos), len, name, seq, 'INS', chr],
    DEL [pos, len, name, 'DEL', chr]. chrom_name: id -> name. `seqs`: DeferredSeq (tokens until resolve_sequences) or seq_of(rec)."""
    seqs, own = _own_seqs(seqs)
    out = {}
    for row in rows:
        rec = int(row["rec"])
        name = soa.qname(rec)
        if int(row["meta"]) & M_DEL:
            c = [int(row["pos"]), int(row["svlen"]), name, "DEL", chrom_name(int(row["tid"]))]
        else:
            tok = seqs.ask(rec, int(row["q_start"]), int(row["q_end"]), bool(int(row["meta"]) & M_QREV))
            c = [int(row["pos"]), int(row["svlen"]), name, [tok], "INS", chrom_name(int(row["tid"]))]
        out.setdefault(rec, []).append(c)
    if own:
        resolve_sequences(seqs, out)
    return out


# ---- the script body: main_ctrl / single_pipe (SE:496-645) --------------------------------------------------------------
def make_tasks(contigs, batches=10000000):
    """SE:598-612: [[chrom, start, end], ...] in BAM header order; contigs = [(name, length)]."""
    tasks = []
    for name, ln in contigs:
        if ln < batches:
            tasks.append([name, 0, ln])
        else:
            pos = 0
            for _ in range(int(ln / batches)):
                tasks.append([name, pos, pos + batches])
                pos += batches
            if pos < ln:
                tasks.append([name, pos, ln])
    return tasks


def load_bed(bed_file, task_list):
    """SE:59-82: per task, the BED regions (widened by 1000 bp) that start inside it or cover its start; None without a BED."""
    if bed_file is None:
        return None
    bed = {}
    with open(bed_file) as f:
        for line in f:
            seq = line.strip().split('\t')
            bed.setdefault(seq[0], []).append((int(seq[1]) - 1000, int(seq[2]) + 1000))
    region_list = [[] for _ in task_list]
    for chrom in bed:
        bed[chrom].sort(key=lambda x: (x[0], x[1]))
        for item in bed[chrom]:
            for i, t in enumerate(task_list):
                if chrom == t[0] and ((t[1] <= item[0] and t[2] > item[0]) or item[0] <= t[1] < item[1]):
                    region_list[i].append(item)
    return region_list


def _pad_empty_cigars(soa, n_ops):
    """Gives every record without a CIGAR (placed-but-unmapped reads; they are F_SKIP) a single zero-length M op, in place."""
    new_n = np.maximum(n_ops, 1)
    new_off = np.concatenate(([0], np.cumsum(new_n))).astype(np.uint64)
    cig = np.zeros(int(new_off[-1]), dtype=np.uint32)
    rec_of = np.repeat(np.arange(len(n_ops)), n_ops)
    within = np.arange(len(soa.cigar), dtype=np.int64) - soa.cigar_off[:-1].astype(np.int64)[rec_of]
    cig[new_off[:-1].astype(np.int64)[rec_of] + within] = soa.cigar
    soa.cigar, soa.cigar_off = cig, new_off


def sort_sigs(lines):
    """`sort -u | sort -k 2,2 -k 3,3n` (SE:637-638) with LC_ALL=C: unique lines ordered by chromosome field (bytes), then
    numeric position, then the whole line (sort's last-resort comparison)."""
    uniq = set(lines)

    def key(line):
        f = line.rstrip("\n").split("\t")
        return (f[1].encode(), int(f[2]), line.encode())

    return sorted(uniq, key=key)


def run(input_bam, reference, work_dir, batches=10000000, max_split_parts=7, min_mapq=20, min_read_len=500, merge_del_threshold=0,
        merge_ins_threshold=100, include_bed=None, min_size=30, max_size=100000, min_siglength=10, device=0, engine=None, log=print,
        device_ingest=True):
    """Writes <work_dir>/INS.sigs, DEL.sigs and reads.sigs like sig_extract.py (SE:575-645). The per-task files under
    signatures/ are not kept; a task without any candidate — INS, DEL or translocation — contributes no reads (single_pipe
    returns early, SE:533-535)."""
    import os

    from .bam import BamFile, DeviceRecordView
    from .engine import Engine
    if not os.path.isfile(reference):
        raise FileNotFoundError("[Errno 2] No such file: '%s'" % reference)
    os.makedirs(work_dir, exist_ok=True)
    out_dir = work_dir if work_dir.endswith('/') else work_dir + '/'
    eng = engine or Engine(device, grow=True)
    p = params(min_siglength, min_mapq, merge_del_threshold, merge_ins_threshold)
    ins_lines, del_lines, task_reads = [], [], {}
    try:
        with BamFile(input_bam) as bam:
            contigs = list(zip(bam.references, bam.lengths))
            tasks = make_tasks(contigs, batches)
            beds = load_bed(include_bed, tasks)
            ids = {n: i for i, n in enumerate(bam.references)}
            names = list(bam.references)

            def chrom_id(n):
                if n not in ids:
                    ids[n] = len(names)
                    names.append(n)
                return ids[n]

            for chrom, ln in contigs:
                # BGZF inflate + record parse on the GPU, the packed sequences stay there (the host reader only if the device
                # reader gives up on the file)
                view = bam.fetch_device(eng, chrom, sa=True, seq=True) if device_ingest else bam.fetch_soa(chrom, keep_seq=True)
                if view.n_records == 0:
                    continue
                if isinstance(view, DeviceRecordView):
                    soa = view.to_host()
                    soa.sa_tags = view.sa_tags
                    seqs = DeferredSeq(view.seq_slices)
                else:
                    soa = view
                    seqs = DeferredSeq(host_seq_fetch(soa.seq))
                n = soa.n_records
                if n == 0:
                    continue
                starts = np.array([t[1] for t in tasks if t[0] == chrom], dtype=np.int64)
                tix = [i for i, t in enumerate(tasks) if t[0] == chrom]
                n_ops = np.diff(soa.cigar_off.astype(np.int64))
                pos = soa.pos.astype(np.int64)
                task_of = np.searchsorted(starts, pos, side="right") - 1           # the task a read starts in (SE:521)
                codes, lens = soa.cigar & 15, (soa.cigar >> 4).astype(np.int64)
                refspan = np.where((codes == 0) | (codes == 2) | (codes == 3) | (codes == 7) | (codes == 8), lens, 0)
                csum = np.concatenate(([0], np.cumsum(refspan)))
                end = pos + csum[soa.cigar_off[1:].astype(np.int64)] - csum[soa.cigar_off[:-1].astype(np.int64)]
                live = (n_ops > 0) & ((soa.sam_flags & 4) == 0)                    # placed-but-unmapped reads carry no alignment
                if beds is not None:                                               # SE:510-520
                    in_bed = np.zeros(n, dtype=bool)
                    for k, ti in enumerate(tix):
                        m = task_of == k
                        for r0, r1 in beds[ti]:
                            in_bed |= m & ~((end <= r0) | (pos >= r1))
                    live &= in_bed
                flags = np.array([flag_bits(int(f), int(q), min_read_len) for f, q in zip(soa.sam_flags, soa.l_seq)], dtype=np.uint8)
                flags[~live] |= F_SKIP
                soa.flag = flags
                if (n_ops == 0).any():                                             # the scan wants >= 1 op per record: pad with one 0M
                    _pad_empty_cigars(soa, n_ops)
                eng.run(soa, p)
                raw, comb = eng.table("raw"), eng.table("cigar")
                cig = cigar_candidates(soa, raw, comb, seqs, chrom)
                ok = live & (soa.l_seq.astype(np.int64) >= min_read_len)
                sflags = np.where(ok, soa.sam_flags, 0xFFFF)                       # skipped reads never reach the split branch
                sreads = split_reads(soa, sflags, soa.l_seq, soa.sa_tags, chrom_id, min_mapq)
                spl = {}
                if sreads:
                    seg = SplitSegments(sreads)
                    rows = eng.cutesv_split(seg, seg.read_len, seg.read_rec, min_size, max_size, max_split_parts)
                    spl = split_candidates(soa, rows, seqs, lambda t: names[t])
                resolve_sequences(seqs, cig, spl)
                has_cand = np.zeros(len(tix), dtype=bool)
                if sreads:
                    # a translocation candidate (analysis_bnd) never reaches INS.sigs / DEL.sigs, but its task is not empty:
                    # single_pipe writes the task's reads (SE:533-535, 556-560)
                    tra = eng.cutesv_split_tra()
                    has_cand[task_of[np.asarray(seg.read_rec)[tra != 0]]] = True
                for rec in sorted(set(cig) | set(spl)):
                    for c in cig.get(rec, []) + spl.get(rec, []):
                        has_cand[task_of[rec]] = True
                        if len(c) == 6:                                            # SE:546-553 (the %d of a float position truncates)
                            ins_lines.append("%s\t%s\t%d\t%d\t%s\t%s\n" % (c[4], c[5], c[0], c[1], c[2], c[3]))
                        else:
                            del_lines.append("%s\t%s\t%d\t%d\t%s\n" % (c[3], c[4], c[0], c[1], c[2]))
                for k, ti in enumerate(tix):                                       # SE:524-529, 556-560
                    if has_cand[k]:
                        t = tasks[ti]
                        sel = np.flatnonzero(live & (task_of == k) & (soa.mapq >= min_mapq))
                        task_reads["_%s_%d_%d" % (t[0], t[1], t[2])] = [
                            "%s\t%d\t%d\t%d\t%s\n" % (chrom, pos[i], end[i], 1 if int(soa.sam_flags[i]) in (0, 16) else 0, soa.qname(i)) for i in sel]
                log("%s: %d records -> %d INS / %d DEL candidate lines so far" % (chrom, n, len(ins_lines), len(del_lines)))
    finally:
        if engine is None:
            eng.close()
    with open(out_dir + "INS.sigs", "w") as f:
        f.writelines(sort_sigs(ins_lines))
    with open(out_dir + "DEL.sigs", "w") as f:
        f.writelines(sort_sigs(del_lines))
    with open(out_dir + "reads.sigs", "w") as f:                                   # cat signatures/*.reads: glob order of the file names
        for name in sorted(task_reads, key=lambda s: s.encode()):
            f.writelines(task_reads[name])
    return out_dir
