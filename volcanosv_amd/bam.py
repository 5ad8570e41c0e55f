"""BAM <-> record SoA on the host.

`read_bam` is the replacement of `pysam.AlignmentFile(path).fetch(chrom)` in the reference
(extract_contig_signature_Hifi.py:387-391, extract_reads_signature.py:108-113): it goes through the C-ABI ingest
(vsv_bam_open / vsv_bam_load, BGZF inflate + record parse in native code) and returns a RecordSoA plus the name tables.
`write_bam` is a small pure-Python BAM writer used to build test inputs (no samtools / pysam in this image).
"""
import ctypes as C
import struct
import zlib

import numpy as np

from . import _lib
from .abi import Records, VsvError
from .soa import RecordSoA


class LazyLines:
    """'\\n'-joined text blob viewed as a read-only list of strings; lines are decoded on access (a 10^7-record BAM has
    10^7 names: splitting them eagerly costs more than the ingest itself)."""

    def __init__(self, blob, n):
        self.blob = blob
        self.n = n
        self._start = self._end = None

    def _index(self):
        """Line boundaries, found on the first access (a run that emits no signature never needs them)."""
        if self._start is None:
            if self.n:
                nl = np.flatnonzero(np.frombuffer(self.blob, dtype=np.uint8) == 10)
                self._start = np.concatenate(([0], nl + 1)).astype(np.int64)
                self._end = np.concatenate((nl, [len(self.blob)])).astype(np.int64)
            else:
                self._start = self._end = np.zeros(0, np.int64)
        return self._start, self._end

    @property
    def start(self):
        return self._index()[0]

    @property
    def end(self):
        return self._index()[1]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self.n))]
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        return self.blob[int(self.start[i]):int(self.end[i])].decode()

    def __iter__(self):
        return (self[i] for i in range(self.n))

    def __eq__(self, other):
        return list(self) == list(other)


class PackedSeq:
    """The 4-bit SEQ fields of the loaded records (BAM nibble alphabet); `seq[i]` decodes record i like pysam's
    query_sequence."""
    ALPHABET = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)

    def __init__(self, packed, l_seq):
        self.packed = packed
        self.l_seq = l_seq
        nbytes = (l_seq.astype(np.int64) + 1) // 2
        self.off = np.concatenate(([0], np.cumsum(nbytes)))

    def __getitem__(self, i):
        n = int(self.l_seq[i])
        b = self.packed[int(self.off[i]):int(self.off[i + 1])]
        nib = np.empty(2 * len(b), dtype=np.uint8)
        nib[0::2] = b >> 4
        nib[1::2] = b & 15
        return self.ALPHABET[nib[:n]].tobytes().decode()


class LightRecords:
    """Host-side companion of a DeviceRecordView for the VCF / signature text: qname(rec), strand(rec), mapq[rec], tid_names."""

    def __init__(self, qid, flag, mapq, tid, qnames, tid_names):
        self.qid, self.flag, self.mapq, self.tid, self.qnames, self.tid_names = qid, flag, mapq, tid, qnames, tid_names
        self.n_records = len(qid)

    def qname(self, rec):
        return self.qnames[int(self.qid[rec])]

    def strand(self, rec):
        return "-" if self.flag[rec] & 1 else "+"


class DeviceRecordView:
    """Record SoA resident on the GPU, produced by BamFile.fetch_device (raw device pointers owned by the engine's handle).
    Quacks like RecordSoA / DeviceRecords where the engine needs it (`as_struct`); `to_host()` copies the arrays back."""

    def __init__(self, rec, qnames, tid_names, l_seq_ptr, sam_flags_ptr, engine):
        self._rec = rec
        self.qnames, self.tid_names = qnames, tid_names
        self.n_records, self.n_ops = int(rec.n_records), int(rec.n_ops)
        self.n_qids, self.n_tids = int(rec.n_qids), int(rec.n_tids)
        self.max_pos = 0
        self.only_tid = None        # set when every record has this tid: trims the sort keys to one tid bit
        self._l_seq_ptr, self._sam_flags_ptr = l_seq_ptr, sam_flags_ptr
        self._engine = engine

    def as_struct(self):
        self._rec.max_pos = int(self.max_pos)
        if self.only_tid is not None:
            self._rec.tid_lo, self._rec.n_tids = int(self.only_tid), int(self.only_tid) + 1
        return self._rec

    def qname(self, rec):
        raise NotImplementedError("qid lives on the device: use to_host().qname(rec)")

    def _pull(self, ptr, count, dt, first=0):
        out = np.zeros(count, dtype=dt)
        if count:
            eng = self._engine
            addr = (ptr if isinstance(ptr, int) else int(ptr or 0)) + first * np.dtype(dt).itemsize
            eng._check(eng.lib.vsv_copy_to_host(eng.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(addr), out.nbytes))
        return out

    def host_light(self):
        """What the text writers need on the host (query ids, flags, mapq, tids; 10 bytes per record) — the CIGARs stay on the GPU."""
        r, n = self._rec, self.n_records
        return LightRecords(self._pull(r.qid, n, np.uint32), self._pull(r.flag, n, np.uint8), self._pull(r.mapq, n, np.uint8),
                            self._pull(r.tid, n, np.int32), self.qnames, self.tid_names)

    def slice_records(self, lo, hi):
        """View of records [lo, hi) (a chromosome of a coordinate-sorted file): same device arrays, offset pointers. The CIGAR
        array and cigar_off values stay absolute."""
        r = Records()
        for name, size in (("pos", 4), ("tid", 4), ("qid", 4), ("cigar_off", 8), ("mapq", 1), ("flag", 1)):
            setattr(r, name, C.c_void_p(int(getattr(self._rec, name) or 0) + lo * size))
        r.cigar = self._rec.cigar
        r.n_records, r.n_ops, r.on_device, r.n_qids, r.n_tids = hi - lo, self._rec.n_ops, 1, self._rec.n_qids, self._rec.n_tids
        v = DeviceRecordView(r, self.qnames, self.tid_names, int(self._l_seq_ptr or 0) + 4 * lo, int(self._sam_flags_ptr or 0) + 4 * lo, self._engine)
        v._parent = self
        v.has_seq, v._first_record = getattr(self, "has_seq", False), getattr(self, "_first_record", 0) + lo
        return v

    def seq_slices(self, requests):
        """requests: [(record, start, stop, reversed)] in Python slice terms (record's query_sequence[start:stop], of the reversed
        read if `reversed`; start / stop may be negative or run past the end exactly like a Python slice). Returns the strings. The
        sequences stay on the GPU (fetch_device(seq=True)); one call decodes all slices."""
        if not getattr(self, "has_seq", False):
            raise ValueError("fetch_device(seq=True) keeps the sequences on the device")
        n = len(requests)
        if n == 0:
            return []
        l_seq = self.l_seq_host()
        first = getattr(self, "_first_record", 0)
        rec = np.zeros(n, np.uint32); start = np.zeros(n, np.uint32); ln = np.zeros(n, np.uint32); rev = np.zeros(n, np.uint8)
        for i, (r, a, b, rv) in enumerate(requests):
            lo, hi, _ = slice(a, b).indices(int(l_seq[r]))
            rec[i], start[i], ln[i], rev[i] = first + r, lo, max(0, hi - lo), 1 if rv else 0
        off = np.zeros(n + 1, np.uint64)
        np.cumsum(ln, out=off[1:])
        out = np.zeros(int(off[n]) + 1, np.uint8)
        eng = self._engine
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        eng._check(eng.lib.vsv_bam_device_seq_slices(eng.h, vp(rec), vp(start), vp(ln), vp(rev), n, vp(off), vp(out), int(off[n])))
        text = out[:int(off[n])].tobytes().decode()
        return [text[int(off[i]):int(off[i + 1])] for i in range(n)]

    def l_seq_host(self):
        if getattr(self, "_l_seq_host", None) is None:
            self._l_seq_host = self._pull(self._l_seq_ptr, self.n_records, np.uint32)
        return self._l_seq_host

    def to_host(self):
        """RecordSoA copy (plus l_seq / sam_flags) of the device arrays."""
        r, n, eng = self._rec, self.n_records, self._engine

        def pull(ptr, count, dt):
            out = np.zeros(count, dtype=dt)
            if count:
                eng._check(eng.lib.vsv_copy_to_host(eng.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr if isinstance(ptr, int) else ptr), out.nbytes))
            return out

        soa = RecordSoA(pull(r.pos, n, np.int32), pull(r.tid, n, np.int32), pull(r.qid, n, np.uint32), pull(r.cigar_off, n + 1, np.uint64),
                        pull(r.mapq, n, np.uint8), pull(r.flag, n, np.uint8), pull(r.cigar, self.n_ops, np.uint32), self.qnames, self.tid_names)
        soa.n_tids = self.n_tids
        soa.l_seq = pull(self._l_seq_ptr, n, np.uint32)
        soa.sam_flags = pull(self._sam_flags_ptr, n, np.uint32)
        return soa


class BamFile:
    def __init__(self, path, threads=0):
        self.lib = _lib.load()
        self.h = C.c_void_p()
        st = self.lib.vsv_bam_open(str(path).encode(), C.byref(self.h))
        if st:
            raise VsvError(st, "cannot open BAM %s" % path)
        self.lib.vsv_bam_set_threads(self.h, int(threads))
        n = self.lib.vsv_bam_n_refs(self.h)
        self.references = [self.lib.vsv_bam_ref_name(self.h, i).decode() for i in range(n)]
        self.lengths = [int(self.lib.vsv_bam_ref_len(self.h, i)) for i in range(n)]

    def close(self):
        if self.h:
            self.lib.vsv_bam_close(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def get_tid(self, name):
        return self.references.index(name) if name in self.references else -1

    def fetch_device(self, engine, chrom=None, sa=False, seq=False):
        """All records of `chrom` (or of every reference) inflated and parsed on the GPU: returns a DeviceRecordView whose arrays
        live in `engine`'s handle (valid until its next fetch_device). Query names come back to the host (lazily split); with
        sa=True the SA:Z tag texts too (`view.sa_tags`, '' where absent); with seq=True the packed SEQ fields stay on the device and
        `view.seq_slices` decodes slices of them."""
        tid = -1 if chrom is None else self.get_tid(chrom)
        if chrom is not None and tid < 0:
            raise KeyError("reference %r not in BAM header" % chrom)
        r = Records()
        self.lib.vsv_bam_device_want_sa(engine.h, 1 if sa else 0)
        self.lib.vsv_bam_device_want_seq(engine.h, 1 if seq else 0)
        st = self.lib.vsv_bam_load_device(self.h, engine.h, tid, C.byref(r))
        if st:
            msg = self.lib.vsv_bam_error(self.h).decode()
            if "use the host reader" in msg:          # hash collision / speculation that does not settle: correct, just not on the GPU
                import warnings
                warnings.warn("device BAM reader gave up (%s); falling back to the host reader" % msg)
                return self.fetch_soa(chrom, keep_seq=seq)
            raise VsvError(st, msg)
        ln = C.c_int64()
        p = self.lib.vsv_bam_qnames(self.h, C.byref(ln))
        names = LazyLines(C.string_at(p, ln.value), int(r.n_qids))
        view = DeviceRecordView(r, names, self.references, self.lib.vsv_bam_l_seq_device(self.h), self.lib.vsv_bam_sam_flags_device(self.h), engine)
        if sa:
            p = self.lib.vsv_bam_device_sa_tags(engine.h, C.byref(ln))
            view.sa_tags = LazyLines(C.string_at(p, ln.value), int(r.n_records))
        view.has_seq = bool(seq)
        return view

    def use_gpu_inflate(self, engine):
        """Inflate the BGZF windows of the following loads on the GPU (engine = volcanosv_amd.engine.Engine; None: host zlib)."""
        self.lib.vsv_bam_set_inflate_device(self.h, engine.h if engine is not None else None)
        self._inflate_engine = engine

    def fetch_soa(self, chrom=None, keep_seq=False):
        """All records of `chrom` (or of every reference) in file order as a RecordSoA (arrays are copied out of the
        ingest object). soa.sa_tags[i] is the SA tag text of record i ('' if absent)."""
        tid = -1 if chrom is None else self.get_tid(chrom)
        if chrom is not None and tid < 0:
            raise KeyError("reference %r not in BAM header" % chrom)
        r = Records()
        self.lib.vsv_bam_set_keep_seq(self.h, 1 if keep_seq else 0)
        st = self.lib.vsv_bam_load(self.h, tid, C.byref(r))
        if st:
            raise VsvError(st, self.lib.vsv_bam_error(self.h).decode())
        n, nops = int(r.n_records), int(r.n_ops)

        def arr(ptr, count, dt):
            if count == 0:
                return np.zeros(0, dtype=dt)
            return np.frombuffer((C.c_char * (count * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt, count=count).copy()

        ln = C.c_int64()
        p = self.lib.vsv_bam_qnames(self.h, C.byref(ln))
        qnames = LazyLines(C.string_at(p, ln.value), int(r.n_qids))
        p = self.lib.vsv_bam_sa_tags(self.h, C.byref(ln))
        sa = LazyLines(C.string_at(p, ln.value), n)
        soa = RecordSoA(arr(r.pos, n, np.int32), arr(r.tid, n, np.int32), arr(r.qid, n, np.uint32),
                        arr(r.cigar_off, n + 1, np.uint64) if n else np.zeros(1, np.uint64), arr(r.mapq, n, np.uint8),
                        arr(r.flag, n, np.uint8), arr(r.cigar, nops, np.uint32), qnames, self.references)
        soa.n_tids = len(self.references)
        soa.sa_tags = sa
        soa.l_seq = arr(self.lib.vsv_bam_l_seq(self.h), n, np.uint32)
        soa.sam_flags = arr(self.lib.vsv_bam_sam_flags(self.h), n, np.uint32)
        if keep_seq:
            p = self.lib.vsv_bam_seq(self.h, C.byref(ln))
            soa.seq = PackedSeq(np.frombuffer(C.string_at(p, ln.value), dtype=np.uint8), soa.l_seq)
        return soa


def read_bam(path, chrom=None, threads=0):
    with BamFile(path, threads) as b:
        return b.fetch_soa(chrom)


# ---- minimal writer (tests / synthetic plumbing inputs) ------------------------------------------------------------
def _bgzf_block(data, level=6):
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    c = comp.compress(data) + comp.flush()
    bsize = len(c) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + c +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def _reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


_NIBBLE = np.full(256, 15, dtype=np.uint8)      # ASCII base -> BAM 4-bit code ('N' for anything outside the alphabet)
for _i, _ch in enumerate("=ACMGRSVTWYHKDBN"):
    _NIBBLE[ord(_ch)] = _i


def write_bam(path, references, records, header_text=None, block_bytes=60000, level=6, mutate=None):
    """references: [(name, length)]; records: iterable of dicts with keys tid, pos, qname, mapq, flag (SAM flag),
    cigar [(op,len)], optional seq (bases) or seq_len (bases are written as 'N'; 0 => '*'), optional tags {b'SA': 'text'},
    optional aux (raw, already encoded aux fields written before the CG / Z tags). block_bytes: uncompressed bytes per BGZF
    member (an int, or a callable returning the size of the next member); level: zlib level (0 = stored blocks); mutate(stream,
    record_offsets): optional hook that edits the uncompressed BAM stream in place before it is framed (malformed-input tests)."""
    if header_text is None:
        header_text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in references)
    out = bytearray()
    ht = header_text.encode()
    out += b"BAM\x01" + struct.pack("<i", len(ht)) + ht + struct.pack("<i", len(references))
    for name, length in references:
        nm = name.encode() + b"\x00"
        out += struct.pack("<i", len(nm)) + nm + struct.pack("<i", length)
    rec_offsets = []
    for r in records:
        rec_offsets.append(len(out))
        qn = r["qname"].encode() + b"\x00"
        cig = r["cigar"]
        ref_len = sum(l for op, l in cig if op in (0, 2, 3, 7, 8))
        seq = r.get("seq")
        l_seq = len(seq) if seq is not None else int(r.get("seq_len", 0))
        long_cigar = len(cig) > 65535
        cig_words = [(l << 4) | op for op, l in cig]
        tags = bytes(r.get("aux") or b"")
        if long_cigar:  # htslib convention: placeholder CIGAR kSmN + real CIGAR in CG:B,I
            tags += b"CGBI" + struct.pack("<I", len(cig_words)) + struct.pack("<%dI" % len(cig_words), *cig_words)
            cig_words = [(l_seq << 4) | 4, (ref_len << 4) | 3]
        for k, v in (r.get("tags") or {}).items():
            tags += k + b"Z" + v.encode() + b"\x00"
        body = struct.pack("<iiBBHHHiiii", r["tid"], r["pos"], len(qn), r["mapq"], _reg2bin(r["pos"], r["pos"] + max(ref_len, 1)),
                           len(cig_words), r["flag"], l_seq, -1, -1, 0)
        body += qn + struct.pack("<%dI" % len(cig_words), *cig_words)
        if seq is not None:
            nib = np.append(_NIBBLE[np.frombuffer(seq.encode(), dtype=np.uint8)], np.uint8(0))
            packed = ((nib[0:2 * ((l_seq + 1) // 2):2] << 4) | nib[1:2 * ((l_seq + 1) // 2):2]).tobytes()
        else:
            packed = b"\xff" * ((l_seq + 1) // 2)
        body += packed + b"\xff" * l_seq + tags
        out += struct.pack("<i", len(body)) + body
    if mutate is not None:
        mutate(out, rec_offsets)
    with open(path, "wb") as f:
        i = 0
        while i < len(out):
            step = int(block_bytes() if callable(block_bytes) else block_bytes)
            step = max(1, min(step, 65280))
            f.write(_bgzf_block(bytes(out[i:i + step]), level))
            i += step
        f.write(_bgzf_block(b""))  # EOF marker
