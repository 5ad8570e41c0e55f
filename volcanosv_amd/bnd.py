"""Host side of the Complex_SV breakend (BND) branch: segment SoA construction, VCF text, filter_tra merge.

Mirrors, on top of the C-ABI rows (include/volcanosv.h `vsv_bnd`):
  retrieve_other_alignments      svim_asm/SVIM_COLLECT.py:8-54   (SA tag -> supplementary alignments)
  analyze_read_segments prologue svim_asm/SVIM_inter.py:62-81    (query coordinates flipped for reverse strands)
  CandidateBreakend.get_vcf_entry / _reverse   svim_asm/SVCandidate.py:389-443
  BND part of write_final_vcf + sorted_nicely  svim_asm/SVIM_COMBINE.py:369-376, 461-477
  load_raw_vcf / cluster_bnd / merge_bnd       Complex_SV/filter_tra.py:32-116
"""
import ctypes as C
import re
from collections import defaultdict

import numpy as np

from .abi import B_DST_FWD, B_GT_SHIFT, B_SRC_FWD, Segments

_CIG = re.compile(r"(\d+)([MIDNSHP=X])")
_OPS = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8}


def cigar_stats(cig):
    """cig: [(op, len)]. Returns (reference span, query_alignment_start, query_alignment_end, infer_read_length) with pysam's
    conventions for a record WITHOUT a stored sequence (the SA-tag alignments of SVIM_COLLECT.py:33-34): start = leading
    soft clips, end = sum(M,I,S,=,X) - trailing soft clips, read length includes hard clips."""
    ref = sum(l for op, l in cig if op in (0, 2, 3, 7, 8))
    qlen = sum(l for op, l in cig if op in (0, 1, 4, 7, 8))
    rl = qlen + sum(l for op, l in cig if op == 5)
    lead = 0
    for op, l in cig:
        if op == 5:
            continue
        if op == 4:
            lead += l
        else:
            break
    trail = 0
    for op, l in reversed(cig):
        if op == 5:
            continue
        if op == 4:
            trail += l
        else:
            break
    return ref, lead, qlen - trail, rl


def parse_sa(sa_text, tid_of):
    """SA tag -> [(tid, pos0, reverse, cigar, mapq)] (SVIM_COLLECT.py:17-52; malformed entries skipped, mapq
    OverflowError -> 0 is pysam's uint8 setter: values outside 0..255)."""
    out = []
    for el in sa_text.split(";"):
        f = el.split(",")
        if len(f) != 6:
            continue
        cig = [(_OPS[o], int(n)) for n, o in _CIG.findall(f[3])]
        mq = int(f[4])
        if mq < 0 or mq > 255:
            mq = 0
        int(f[5])
        out.append((tid_of(f[0]), int(f[1]) - 1, f[2] != "+", cig, mq))
    return out


class _NoNames:
    """len()-only stand-in for the read names of a synthetic segment table."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return "read%d" % i


class SegmentSoA:
    """vsv_segments on the host. reads: [{'hap': 1|2, 'name': str, 'segs': [[ref_id, ref_start, ref_end,
    query_alignment_start, query_alignment_end, read_length, is_reverse], ...]}] with the primary alignment first."""

    def __init__(self, reads, contigs):
        self.names = [r["name"] for r in reads]
        self.contigs = list(contigs)
        off, qs, qe, rid, rs, re_, rev, hap = [0], [], [], [], [], [], [], []
        for r in reads:
            for s in r["segs"]:
                if s[6]:   # SVIM_inter.py:68-73
                    qs.append(s[5] - s[4]); qe.append(s[5] - s[3])
                else:
                    qs.append(s[3]); qe.append(s[4])
                rid.append(s[0]); rs.append(s[1]); re_.append(s[2]); rev.append(1 if s[6] else 0)
            off.append(len(qs))
            hap.append(r["hap"])
        self.seg_off = np.array(off, np.uint64)
        self.q_start, self.q_end = np.array(qs, np.int32), np.array(qe, np.int32)
        self.ref_id, self.ref_start, self.ref_end = np.array(rid, np.int32), np.array(rs, np.int32), np.array(re_, np.int32)
        self.is_reverse, self.hap = np.array(rev, np.uint8), np.array(hap, np.uint8)
        self.contig_len = np.array([c[1] for c in contigs], np.int32)
        order = sorted(range(len(contigs)), key=lambda i: contigs[i][0])      # Python string order (SVCandidate.py:352)
        rank = np.zeros(len(contigs), np.int32)
        rank[order] = np.arange(len(contigs), dtype=np.int32)
        self.contig_rank = rank

    @classmethod
    def from_arrays(cls, contigs, seg_off, q_start, q_end, ref_id, ref_start, ref_end, is_reverse, hap, names=None):
        """Segment tables built elsewhere (synthetic streams): the arrays of `as_struct`, query coordinates already flipped
        for reverse strands."""
        self = cls([], contigs)
        self.seg_off = np.ascontiguousarray(seg_off, np.uint64)
        self.q_start, self.q_end = np.ascontiguousarray(q_start, np.int32), np.ascontiguousarray(q_end, np.int32)
        self.ref_id, self.ref_start, self.ref_end = (np.ascontiguousarray(a, np.int32) for a in (ref_id, ref_start, ref_end))
        self.is_reverse, self.hap = np.ascontiguousarray(is_reverse, np.uint8), np.ascontiguousarray(hap, np.uint8)
        self.n_reads = len(self.hap)
        self.names = names if names is not None else _NoNames(self.n_reads)
        return self

    def as_struct(self):
        s = Segments()
        s.n_reads, s.n_segs = len(self.names), int(self.q_start.shape[0])
        for k in ("seg_off", "q_start", "q_end", "ref_id", "ref_start", "ref_end", "is_reverse", "hap", "contig_len", "contig_rank"):
            setattr(s, k, getattr(self, k).ctypes.data_as(C.c_void_p))
        s.n_tids, s.on_device = len(self.contigs), 0
        return s


class DeviceSegments:
    """A SegmentSoA copied to the GPU once (torch tensors; `as_struct` hands device pointers, on_device = 1): the form a rank keeps
    its split-contig table in when the breakend branch runs step after step (bench config 5)."""

    def __init__(self, seg, device):
        import torch
        self.host = seg
        self.n_reads, self.n_segs = len(seg.hap), int(seg.q_start.shape[0])
        self.t = {k: torch.from_numpy(np.ascontiguousarray(getattr(seg, k)).view(np.int64 if k == "seg_off" else getattr(seg, k).dtype)).to(device)
                  for k in ("seg_off", "q_start", "q_end", "ref_id", "ref_start", "ref_end", "is_reverse", "hap", "contig_len", "contig_rank")}
        torch.cuda.current_stream(device).synchronize()

    def as_struct(self):
        s = Segments()
        s.n_reads, s.n_segs = self.n_reads, self.n_segs
        for k, v in self.t.items():
            setattr(s, k, C.c_void_p(v.data_ptr()))
        s.n_tids, s.on_device = len(self.host.contigs), 1
        return s


def segments_from_soa(soa, hap, min_mapq=20):
    """Record SoA of one haplotype BAM (volcanosv_amd.bam, with SA tags) -> reads list for SegmentSoA:
    analyze_alignment_file_coordsorted (SVIM_COLLECT.py:57-79) restricted to what the BND branch needs."""
    from .abi import F_SECONDARY, F_SUPP, F_UNMAPPED
    names = soa.tid_names
    reads = []
    for i in range(soa.n_records):
        fl = int(soa.flag[i])
        if fl & (F_UNMAPPED | F_SECONDARY) or int(soa.mapq[i]) < min_mapq or fl & F_SUPP:
            continue
        sa = soa.sa_tags[i] if hasattr(soa, "sa_tags") else ""
        if not sa:
            continue
        cig = [(int(w) & 15, int(w) >> 4) for w in soa.cigar[int(soa.cigar_off[i]):int(soa.cigar_off[i + 1])]]
        if any(op == 5 and l > 0 for op, l in cig):       # SVIM_COLLECT.py:11: hard-clipped primary -> no reconstruction
            continue
        others = [a for a in parse_sa(sa, lambda n: names.index(n)) if a[4] >= min_mapq]
        if not others:
            continue
        ref, qa, qb, rl = cigar_stats(cig)
        # the primary HAS a sequence: pysam's query_alignment_end = l_qseq - trailing soft clips = same formula here
        segs = [[int(soa.tid[i]), int(soa.pos[i]), int(soa.pos[i]) + ref, qa, qb, rl, 1 if fl & 1 else 0]]
        for tid, pos0, rev, c2, mq in others:
            r2, a2, b2, l2 = cigar_stats(c2)
            segs.append([tid, pos0, pos0 + r2, a2, b2, l2, 1 if rev else 0])
        reads.append({"hap": hap, "name": soa.qname(i), "segs": segs})
    return reads


# ---- VCF text -------------------------------------------------------------------------------------------------------
_GT = {1: "1/0", 2: "0/1", 3: "1/1"}


def _alt(src_fwd, dst_fwd, contig, pos1, reverse):
    if not reverse:      # get_vcf_entry, SVCandidate.py:392-399
        if src_fwd and dst_fwd:
            return "N[%s:%d[" % (contig, pos1)
        if src_fwd and not dst_fwd:
            return "N]%s:%d]" % (contig, pos1)
        if not src_fwd and not dst_fwd:
            return "]%s:%d]N" % (contig, pos1)
        return "[%s:%d[N" % (contig, pos1)
    if not src_fwd and not dst_fwd:   # get_vcf_entry_reverse, SVCandidate.py:420-427
        return "N[%s:%d[" % (contig, pos1)
    if src_fwd and not dst_fwd:
        return "N]%s:%d]" % (contig, pos1)
    if src_fwd and dst_fwd:
        return "]%s:%d]N" % (contig, pos1)
    return "[%s:%d[N" % (contig, pos1)


def call_fields(seg, c):
    names = [x[0] for x in seg.contigs]
    reads = [seg.names[int(c["read"])]] + ([seg.names[int(c["read2"])]] if int(c["read2"]) != 0xFFFFFFFF else [])
    m = int(c["meta"])
    return [names[int(c["src_tid"])], int(c["src_pos"]), "fwd" if m & B_SRC_FWD else "rev", names[int(c["dst_tid"])], int(c["dst_pos"]),
            "fwd" if m & B_DST_FWD else "rev", _GT[(m >> B_GT_SHIFT) & 3], reads]


def natural_key(entry):
    conv = lambda t: int(t) if t.isdigit() else t
    return ([conv(c) for c in re.split("([0-9]+)", str(entry[0][0]))], entry[0][1], entry[0][2])


def vcf_lines(seg, calls, query_names=True, id_prefix="svim_asm"):
    """Two lines per breakend, natural-sorted, ids <prefix>.BND.<n> (SVIM_COMBINE.py:461-477)."""
    entries = []
    for c in calls:
        f = call_fields(seg, c)
        info = "SVTYPE=BND" + ((";READS=" + ",".join(f[7])) if query_names else "")
        for reverse in (False, True):
            chrom, pos, ochrom, opos = (f[0], f[1], f[3], f[4]) if not reverse else (f[3], f[4], f[0], f[1])
            alt = _alt(f[2] == "fwd", f[5] == "fwd", ochrom, opos + 1, reverse)
            line = "%s\t%d\tPLACEHOLDERFORID\tN\t%s\t.\tPASS\t%s\tGT\t%s" % (chrom, pos + 1, alt, info, f[6])
            entries.append(((chrom, pos + 1, pos + 2), line))
    out = []
    for n, (_, line) in enumerate(sorted(entries, key=natural_key)):
        out.append(line.replace("PLACEHOLDERFORID", "%s.BND.%d" % (id_prefix, n + 1), 1))
    return out


# ---- filter_tra.py --------------------------------------------------------------------------------------------------
def merge_bnd_lines(lines, max_dist=100):
    """filter_tra.py load_raw_vcf + cluster_bnd + merge_bnd (:32-116) on VCF text lines. Returns (header, body)."""
    dc, header, l1, l2 = defaultdict(list), [], [], []
    for line in lines:
        if line[0] != "#" and "SVTYPE=BND" in line:
            d = line.split()
            chrom2, pos2 = d[4].replace("]", "[").split("[")[1].split(":")
            key = (d[0], int(d[1]), chrom2, int(pos2), "[" if "[" in d[4] else "]")
            (l1 if key[4] == "[" else l2).append(key)
            dc[key].append(line)
        elif line[0] == "#":
            header.append(line)

    def cluster(lst):
        out = {}
        if not lst:
            return out
        clusters = [[lst[0]]]
        for new in lst[1:]:
            old = clusters[-1][-1]
            if (new[0], new[2]) == (old[0], old[2]) and new[1] - old[1] <= max_dist and new[3] - old[3] <= max_dist and new[4] == old[4]:
                clusters[-1].append(new)
            else:
                clusters.append([new])
        for cl in clusters:
            p1 = [b[1] for b in cl]
            center = (cl[0][0], int(sum(p1) / len(p1)), cl[0][2], int(sum(p1) / len(cl)), cl[0][4])   # sic: avg_pos2 uses pos1 (:64)
            for b in cl:
                out[b] = center
        return out

    m = cluster(l1)
    m.update(cluster(l2))
    merged = defaultdict(list)
    for key, val in dc.items():
        merged[m[key]].extend(val)
    body = []
    for val in merged.values():
        if len(val) == 1:
            body.append(val[0])
        else:
            d = val[0].split()
            d[-1] = "1/1"
            body.append("\t".join(d) + "\n")
    return header, body


def vcf_header(contigs, query_names=True, sample="Sample"):
    """Header lines of variants.vcf exactly as `svim-asm diploid` writes them with its default type list
    (DEL,INS,INV,DUP:TANDEM,DUP:INT,BND; SVIM_COMBINE.py:394-425) — the breakend branch only emits BND records, but the caller
    (volcanosv-vc-complex-sv.py phase_vcf, :78) indexes the header from its end, so the line set is kept whole."""
    import time
    h = ["##fileformat=VCFv4.2", "##fileDate=%s" % time.strftime("%Y-%m-%d|%I:%M:%S%p|%Z|%z"), "##source=SVIM-asm-v1.0.2"]
    h += ["##contig=<ID=%s,length=%d>" % (n, l) for n, l in contigs]
    h += ['##ALT=<ID=DEL,Description="Deletion">', '##ALT=<ID=INV,Description="Inversion">', '##ALT=<ID=DUP,Description="Duplication">',
          '##ALT=<ID=DUP:TANDEM,Description="Tandem Duplication">', '##ALT=<ID=DUP:INT,Description="Interspersed Duplication">',
          '##ALT=<ID=INS,Description="Insertion">', '##ALT=<ID=BND,Description="Breakend">',
          '##INFO=<ID=SVTYPE,Number=1,Type=String,Description="Type of structural variant">',
          '##INFO=<ID=CUTPASTE,Number=0,Type=Flag,Description="Genomic origin of interspersed duplication seems to be deleted">',
          '##INFO=<ID=END,Number=1,Type=Integer,Description="End position of the variant described in this record">',
          '##INFO=<ID=SVLEN,Number=1,Type=Integer,Description="Difference in length between REF and ALT alleles">']
    if query_names:
        h.append('##INFO=<ID=READS,Number=.,Type=String,Description="Names of all supporting reads">')
    h += ['##FILTER=<ID=not_fully_covered,Description="Tandem duplication is not fully covered by a contig">',
          '##FILTER=<ID=incomplete_inversion,Description="Only one inversion breakpoint is supported">',
          '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
          '##FORMAT=<ID=CN,Number=1,Type=Integer,Description="Copy number of tandem duplication (e.g. 2 for one additional copy)">',
          "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample]
    return [x + "\n" for x in h]
