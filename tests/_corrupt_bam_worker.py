"""Worker of tests/test_vcf_bam.py::test_host_reader_survives_corrupted_files: runs in its own process so that a crash of the
native reader shows up as an exit code. Usage: _corrupt_bam_worker.py SEED N"""
import os
import struct
import sys
import tempfile
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from volcanosv_amd import bam  # noqa: E402
from volcanosv_amd.abi import VsvError  # noqa: E402


def inflate_all(path):
    data, out, o = open(path, "rb").read(), bytearray(), 0
    while o < len(data):
        xlen = struct.unpack_from("<H", data, o + 10)[0]
        bsize = struct.unpack_from("<H", data, o + 16)[0] + 1
        out += zlib.decompress(data[o + 12 + xlen:o + bsize - 8], -15)
        o += bsize
    return bytes(out)


def write_raw(path, raw, block=3000):
    with open(path, "wb") as f:
        for i in range(0, len(raw), block):
            f.write(bam._bgzf_block(raw[i:i + block]))
        f.write(bam._bgzf_block(b""))


def corrupt(raw, rng):
    """1-3 semantic corruptions of the uncompressed stream (the BGZF framing stays valid): a random byte, a length-like
    32-bit field, a deleted stretch."""
    b = bytearray(raw)
    for _ in range(int(rng.integers(1, 4))):
        k = int(rng.integers(0, len(b)))
        mode = int(rng.integers(0, 3))
        if mode == 0:
            b[k] = int(rng.integers(0, 256))
        elif mode == 1:
            v = struct.pack("<i", int(rng.choice([-1, 0, 1 << 30, 65535, 70000, -(1 << 31)])))
            b[k:k + 4] = v[: max(0, min(4, len(b) - k))]
        else:
            del b[k:k + int(rng.integers(1, 200))]
    return bytes(b)


def source_bam(d):
    recs = []
    for i in range(60):
        recs.append(dict(tid=0, pos=100 * i, qname="PS%d_hp%d_x" % (i % 7, 1 + i % 2), mapq=60, flag=0, cigar=[(4, 5), (0, 50), (1, 40), (0, 30)], seq_len=125,
                         tags={b"SA": "chr1,%d,+,50S75M,60,1;" % (i + 1)} if i % 3 == 0 else None))
    recs.append(dict(tid=0, pos=9000, qname="longcigar", mapq=60, flag=0, cigar=[(0, 2), (1, 1)] * 33000 + [(0, 1)], seq_len=10))
    src = os.path.join(d, "a.bam")
    bam.write_bam(src, [("chr1", 100000)], recs)
    return src


if __name__ == "__main__":
    rng = np.random.default_rng(int(sys.argv[1]))
    d = tempfile.mkdtemp()
    raw = inflate_all(source_bam(d))
    ok = err = 0
    for it in range(int(sys.argv[2])):
        p = os.path.join(d, "c.bam")
        write_raw(p, corrupt(raw, rng))
        try:
            with bam.BamFile(p, threads=int(rng.integers(1, 4))) as bf:
                s = bf.fetch_soa(None, keep_seq=bool(rng.integers(0, 2)))
                _ = [s.sa_tags[i] for i in range(min(3, s.n_records))]
            ok += 1
        except (VsvError, KeyError, ValueError, OSError):
            err += 1
    print("ok %d errors %d" % (ok, err))
