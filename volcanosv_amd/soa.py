"""Caller-owned SoA of alignment records (include/volcanosv.h `vsv_records`) on the host.

This is the layout the reference's `pysam.AlignmentFile.fetch()` loop (extract_contig_signature_Hifi.py:391,
extract_reads_signature.py:113) is replaced by: one row per BAM record in (tid, pos) order, CIGAR ops in
BAM packing (len<<4|op).
"""
import ctypes as C

import numpy as np

from .abi import F_HP1, F_HP2, F_REVERSE, Records


class RecordSoA:
    """Host-side record arrays. `qnames[qid]` is the query-name table (strings never cross the C-ABI)."""

    def __init__(self, pos, tid, qid, cigar_off, mapq, flag, cigar, qnames=None, tid_names=None):
        self.pos = np.ascontiguousarray(pos, dtype=np.int32)
        self.tid = np.ascontiguousarray(tid, dtype=np.int32)
        self.qid = np.ascontiguousarray(qid, dtype=np.uint32)
        self.cigar_off = np.ascontiguousarray(cigar_off, dtype=np.uint64)
        self.mapq = np.ascontiguousarray(mapq, dtype=np.uint8)
        self.flag = np.ascontiguousarray(flag, dtype=np.uint8)
        self.cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        n = self.pos.shape[0]
        assert self.tid.shape == (n,) and self.qid.shape == (n,) and self.mapq.shape == (n,) and self.flag.shape == (n,)
        assert self.cigar_off.shape == (n + 1,), "cigar_off must have n+1 entries"
        assert int(self.cigar_off[-1]) == self.cigar.shape[0]
        self.qnames = qnames
        self.tid_names = tid_names
        self.n_qids = int(self.qid.max()) + 1 if n else 0
        self.n_tids = int(self.tid.max()) + 1 if n else 0

    @property
    def n_records(self):
        return int(self.pos.shape[0])

    @property
    def n_ops(self):
        return int(self.cigar.shape[0])

    def algorithmic_bytes(self, n_sigs=0):
        """24 B record header + 4 B per CIGAR op read, 32 B per signature written."""
        return 24 * self.n_records + 4 * self.n_ops + 32 * int(n_sigs)

    def as_struct(self):
        r = Records()
        r.n_records, r.n_ops = self.n_records, self.n_ops
        for name in ("pos", "tid", "qid", "cigar_off", "mapq", "flag", "cigar"):
            setattr(r, name, getattr(self, name).ctypes.data_as(C.c_void_p))
        r.on_device, r.n_qids, r.n_tids = 0, self.n_qids, self.n_tids
        r.max_pos = int(getattr(self, "max_pos", 0))
        if getattr(self, "only_tid", None) is not None:      # single-chromosome table: keys need one tid bit
            r.tid_lo, r.n_tids = int(self.only_tid), int(self.only_tid) + 1
        return r

    def host_light(self):
        """Same interface as bam.DeviceRecordView.host_light(): a host SoA already is what the text writers need."""
        return self

    def qname(self, rec):
        q = int(self.qid[rec])
        return self.qnames[q] if self.qnames is not None else "q%d" % q

    def strand(self, rec):
        return "-" if self.flag[rec] & F_REVERSE else "+"

    def slice_records(self, lo, hi):
        """Sub-SoA of records [lo, hi) (re-based cigar offsets, same qid space)."""
        a, b = int(self.cigar_off[lo]), int(self.cigar_off[hi])
        return RecordSoA(self.pos[lo:hi], self.tid[lo:hi], self.qid[lo:hi], self.cigar_off[lo:hi + 1] - np.uint64(a),
                         self.mapq[lo:hi], self.flag[lo:hi], self.cigar[a:b], self.qnames, self.tid_names)

    @staticmethod
    def from_tuples(records, qnames=None, tid_names=None):
        """records: iterable of dicts/tuples (tid, pos, qname_or_qid, mapq, reverse, cigar[(op,len)...], hp_flags=None).
        qname strings get dense ids in first-appearance order; hp flags come from the 'hp1'/'hp2' substring
        test of extract_contig_signature_Hifi.py:392 unless given."""
        pos, tid, qid, mapq, flag, off, ops = [], [], [], [], [], [0], []
        table, names = {}, []
        for rec in records:
            t, p, qn, mq, rev, cig = rec[:6]
            if isinstance(qn, str):
                if qn not in table:
                    table[qn] = len(names)
                    names.append(qn)
                q = table[qn]
                f = (F_HP1 if "hp1" in qn else 0) | (F_HP2 if "hp2" in qn else 0)
            else:
                q, f = int(qn), 0
            if len(rec) > 6 and rec[6] is not None:
                f = int(rec[6])
            f |= F_REVERSE if rev else 0
            pos.append(p); tid.append(t); qid.append(q); mapq.append(mq); flag.append(f)
            for op, ln in cig:
                ops.append((int(ln) << 4) | int(op))
            off.append(len(ops))
        return RecordSoA(pos, tid, qid, off, mapq, flag, np.array(ops, dtype=np.uint32),
                         names if names else qnames, tid_names)
