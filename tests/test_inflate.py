"""SURVEY §8f-1, GPU half: BGZF members inflated on the device (inflate.hip, one lane per member), bit-exact against zlib.

The checker is zlib itself (the reference's htslib links it): every member is produced by zlib's raw deflate at several
levels / strategies so that stored, fixed-Huffman and dynamic-Huffman blocks, long matches, distance-1 runs and empty members
all occur; the BAM-level test loads the same file through the host (zlib threads) and the GPU inflate and compares the SoA."""
import os
import zlib

import numpy as np
import pytest


def raw_deflate(data, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    return c.compress(data) + c.flush()


def make_members(seed=3):
    rng = np.random.default_rng(seed)
    blobs = [b"", b"A", b"ACGT" * 10, bytes(65280), bytes([7]) * 65535]
    blobs.append(rng.integers(0, 256, 65280, dtype=np.uint8).tobytes())                    # incompressible -> stored or near-stored
    blobs.append(rng.integers(0, 4, 65280, dtype=np.uint8).tobytes())                      # 2-bit entropy
    text = (b"chr10\t12345\tvolcano_INS_77\tA\tACGTTTGACCA\t.\tPASS\tSVTYPE=INS;SVLEN=10\tGT\t0/1\n" * 900)[:65000]
    blobs.append(text)
    for _ in range(40):                                                                     # BAM-like: structured records with repeats
        n = int(rng.integers(1, 65280))
        base = rng.integers(0, 16, 300, dtype=np.uint8)
        reps = np.tile(base, n // 300 + 1)[:n].copy()
        idx = rng.integers(0, n, n // 20 + 1)
        reps[idx] = rng.integers(0, 256, len(idx))
        blobs.append(reps.tobytes())
    # what the window decoder treats specially: matches from 8-32 KiB back (source outside the LDS ring), codes longer than the
    # fast tables resolve (a geometric byte distribution gives 12-15-bit literal codes), overlapping matches of short periods
    # (distance 1-5), literal / match alternation at every position, and their mixtures
    far = rng.integers(0, 256, 21000, dtype=np.uint8).tobytes()
    blobs.append((far * 4)[:65280])
    blobs.append((rng.integers(0, 256, 9000, dtype=np.uint8).tobytes() * 8)[:65280])
    geo = np.minimum(rng.geometric(0.06, 65280) - 1, 255).astype(np.uint8)
    blobs.append(geo.tobytes())
    blobs.append(b"".join(bytes([int(rng.integers(0, 256))] * int(rng.integers(1, 6))) * int(rng.integers(1, 40)) for _ in range(3000))[:65280])
    alt = bytearray()
    while len(alt) < 65000:
        alt += rng.integers(0, 256, int(rng.integers(1, 4)), dtype=np.uint8).tobytes()
        if len(alt) > 40:
            a = int(rng.integers(0, len(alt) - 12))
            alt += alt[a:a + int(rng.integers(3, 12))]
    blobs.append(bytes(alt[:65280]))
    mix = bytearray(geo[:30000].tobytes()) + bytearray(far[:15000]) + bytearray(geo[:5000].tobytes()) + bytearray(far[:15000])
    blobs.append(bytes(mix[:65280]))
    members = []
    for i, d in enumerate(blobs):
        for level, strat in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                             (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            if (i + level) % 2 == 0 or i < 8 or i >= len(blobs) - 6:
                members.append((raw_deflate(d, level, strat), d))
    return members


@pytest.mark.gpu
def test_gpu_inflate_equals_zlib():
    from volcanosv_amd.abi import VsvError
    from volcanosv_amd.engine import Engine
    members = make_members()
    assert len(members) > 150
    with Engine(0) as eng:
        got = eng.bgzf_inflate([m[0] for m in members], [len(m[1]) for m in members])
        for k, (g, (_, want)) in enumerate(zip(got, members)):
            assert g == want, k
        assert eng.bgzf_inflate([], []) == []
        # corrupt streams are reported, not decoded: truncated payload, wrong ISIZE, garbage
        bad = members[20][0]
        for payload, size in ((bad[: len(bad) // 2], len(members[20][1])), (bad, len(members[20][1]) + 1), (b"\\xff" * 40, 100)):
            with pytest.raises(VsvError) as e:
                eng.bgzf_inflate([members[3][0], payload], [len(members[3][1]), size])
            assert e.value.status == -1 and eng.last_count() == 1


@pytest.mark.gpu
def test_bam_load_with_gpu_inflate_equals_host_inflate(tmp_path):
    from volcanosv_amd import bam, synth
    from volcanosv_amd.engine import Engine
    t, nq, _ = synth.generate(60000, "hifi", seed=12)
    soa = synth.to_soa(t, nq)
    recs = []
    for i in range(soa.n_records):
        a, b = int(soa.cigar_off[i]), int(soa.cigar_off[i + 1])
        recs.append(dict(tid=0, pos=int(soa.pos[i]), qname="PS%d_hp%d_r" % (int(soa.qid[i]), 1 + i % 2), mapq=int(soa.mapq[i]), flag=16 * (i % 2),
                         cigar=[(int(w) & 15, int(w) >> 4) for w in soa.cigar[a:b]], seq="ACGTN"[i % 5] * (20 + i % 7)))
    path = str(tmp_path / "r.bam")
    bam.write_bam(path, [("chr10", synth.CHR10_LEN)], recs)
    with bam.BamFile(path) as bf:
        host = bf.fetch_soa("chr10", keep_seq=True)
    with Engine(0) as eng, bam.BamFile(path) as bf:
        bf.use_gpu_inflate(eng)
        dev = bf.fetch_soa("chr10", keep_seq=True)
    for name in ("pos", "tid", "qid", "cigar_off", "mapq", "flag", "cigar", "l_seq", "sam_flags"):
        assert np.array_equal(getattr(host, name), getattr(dev, name)), name
    assert list(host.qnames) == list(dev.qnames) and host.seq[123] == dev.seq[123] and host.n_records == 60000


@pytest.mark.gpu
def test_bam_parsed_on_the_device_equals_host_reader(tmp_path, monkeypatch):
    """vsv_bam_load_device: BGZF inflate + record-start chain + field parse + CIGAR copy + first-appearance query ids, all on the
    GPU; every array and the name table equal the host reader's, for one chromosome and for the whole file, with repeated
    names (split reads), records larger than a BGZF member, a long CIGAR in CG:B,I and unmapped-placed records."""
    from volcanosv_amd import bam, synth
    from volcanosv_amd.abi import DTYPE_HIFI
    from volcanosv_amd.engine import Engine, default_params
    recs = []
    rng = np.random.default_rng(8)
    for tid in (0, 1):
        t, nq, _ = synth.generate(40000, "hifi", seed=30 + tid, chrom_len=30_000_000)
        soa = synth.to_soa(t, nq)
        for i in range(soa.n_records):
            a, b = int(soa.cigar_off[i]), int(soa.cigar_off[i + 1])
            recs.append(dict(tid=tid, pos=int(soa.pos[i]), qname="c%d_PS%d_hp%d" % (tid, int(soa.qid[i]), 1 + int(soa.qid[i]) % 2), mapq=int(soa.mapq[i]),
                             flag=16 * (i % 2), cigar=[(int(w) & 15, int(w) >> 4) for w in soa.cigar[a:b]], seq_len=int(rng.integers(0, 400))))
        # a record far larger than a 64 KiB member (sequence bytes) and one with > 65535 CIGAR ops (CG tag)
        recs.append(dict(tid=tid, pos=29_000_000, qname="giant%d" % tid, mapq=60, flag=0, cigar=[(0, 100)], seq_len=300_000))
        recs.append(dict(tid=tid, pos=29_100_000, qname="longcigar%d" % tid, mapq=60, flag=0, cigar=[(0, 3), (1, 1)] * 40000 + [(0, 5)], seq_len=10))
        recs.append(dict(tid=tid, pos=29_200_000, qname="unmapped_placed%d" % tid, mapq=0, flag=4, cigar=[], seq_len=50))
    path = str(tmp_path / "two.bam")
    bam.write_bam(path, [("chr1", 30_000_000), ("chr2", 30_000_000)], recs)
    with Engine(0) as eng, bam.BamFile(path) as bf:
        for chrom in ("chr2", None, "chr1"):
            host = bf.fetch_soa(chrom)
            view = bf.fetch_device(eng, chrom)
            assert isinstance(view, bam.DeviceRecordView)
            dev = view.to_host()
            assert dev.n_records == host.n_records and dev.n_ops == host.n_ops and view.n_qids == host.n_qids
            for name in ("pos", "tid", "qid", "cigar_off", "mapq", "flag", "cigar", "l_seq", "sam_flags"):
                assert np.array_equal(getattr(host, name), getattr(dev, name)), (chrom, name)
            assert list(host.qnames) == list(dev.qnames)
        # the file streamed through the device in small member windows (WGS-size files do not fit at once): records that
        # straddle a window are completed by the next one; identical output. A window no record of the file fits in is an
        # error the caller answers with the host reader.
        host = bf.fetch_soa(None)
        for window in (16, 37, 300):
            monkeypatch.setenv("VSV_BAM_WINDOW", str(window))
            view = bf.fetch_device(eng, None)
            assert isinstance(view, bam.DeviceRecordView)
            dev = view.to_host()
            for name in ("pos", "tid", "qid", "cigar_off", "mapq", "flag", "cigar", "l_seq", "sam_flags"):
                assert np.array_equal(getattr(host, name), getattr(dev, name)), (window, name)
            assert list(host.qnames) == list(dev.qnames)
        monkeypatch.setenv("VSV_BAM_WINDOW", "2")
        with pytest.warns(UserWarning, match="larger than the device reader's window"):
            assert not isinstance(bf.fetch_device(eng, None), bam.DeviceRecordView)
        monkeypatch.delenv("VSV_BAM_WINDOW")
        # the device-resident view feeds the hot path directly
        view = bf.fetch_device(eng, "chr1")
        view.max_pos = 30_100_000
        keep = view.to_host()
        ok = np.flatnonzero(np.diff(keep.cigar_off.astype(np.int64)) > 0)      # the scan rejects empty CIGARs: same input for both
        p = default_params(DTYPE_HIFI)
        try:
            eng.run(view, p)
            got = eng.table("raw")
            raised = None
        except Exception as e:                                                 # noqa: BLE001
            raised = e
        assert raised is not None and "CIGAR" in str(raised) and len(ok) == keep.n_records - 1


@pytest.mark.gpu
def test_device_reader_rejects_damaged_files(tmp_path):
    """Truncated files, a corrupted deflate payload and a stream that ends inside a record fail with an error (as the host
    reader does), they never hang or return partial tables."""
    from volcanosv_amd import bam, synth
    from volcanosv_amd.abi import VsvError
    from volcanosv_amd.engine import Engine
    t, nq, _ = synth.generate(20000, "hifi", seed=2)
    soa = synth.to_soa(t, nq)
    recs = []
    for i in range(soa.n_records):
        a, b = int(soa.cigar_off[i]), int(soa.cigar_off[i + 1])
        recs.append(dict(tid=0, pos=int(soa.pos[i]), qname="r%d" % i, mapq=60, flag=0, cigar=[(int(w) & 15, int(w) >> 4) for w in soa.cigar[a:b]]))
    good = str(tmp_path / "good.bam")
    bam.write_bam(good, [("chr10", synth.CHR10_LEN)], recs)
    raw = open(good, "rb").read()
    cases = {"cut_in_member.bam": raw[: len(raw) // 2], "bad_payload.bam": raw[: len(raw) // 2] + bytes(64) + raw[len(raw) // 2 + 64:]}
    # a well-formed BGZF file whose record stream stops inside a record: drop the tail members but keep the EOF marker
    with bam.BamFile(good) as bf:
        pass
    offs, o = [], 0
    while o < len(raw):
        bsize = raw[o + 16] | (raw[o + 17] << 8)
        offs.append(o)
        o += bsize + 1
    cases["ends_in_record.bam"] = raw[: offs[len(offs) // 2]] + raw[offs[-1]:]
    with Engine(0) as eng:
        with bam.BamFile(good) as bf:
            assert bf.fetch_device(eng, "chr10").n_records == soa.n_records
        for name, data in cases.items():
            p = str(tmp_path / name)
            open(p, "wb").write(data)
            with bam.BamFile(p) as bf:
                with pytest.raises(VsvError):
                    bf.fetch_device(eng, "chr10")
                with pytest.raises(VsvError):
                    bf.fetch_soa("chr10")


@pytest.mark.gpu
def test_device_reader_rejects_an_empty_read_name(tmp_path):
    """l_read_name counts the NUL, so 0 is malformed (htslib rejects it). The device reader used to accept it and then stored the
    name table's separator one byte in front of the blob (d[l - 1] with l = 0): now the record is reported and the caller
    falls back to the host reader."""
    from volcanosv_amd import bam
    from volcanosv_amd.engine import Engine
    recs = [dict(tid=0, pos=100 + 10 * i, qname="PS1_hp%d_%d" % (1 + i % 2, i), mapq=60, flag=0, cigar=[(0, 50), (1, 40), (0, 50)]) for i in range(200)]

    def zero_name_len(stream, offs):
        stream[offs[77] + 12] = 0          # block_size(4) refID(4) pos(4) l_read_name(1)

    path = str(tmp_path / "noname.bam")
    bam.write_bam(path, [("chr10", 1_000_000)], recs, mutate=zero_name_len)
    from volcanosv_amd.abi import VsvError
    with Engine(0) as eng, bam.BamFile(path) as bf:
        # the device reader reports the record and hands over to the host reader, which (the name's bytes now read as CIGAR and
        # tags) rejects the file as well: an error, never a store in front of the name table
        with pytest.warns(UserWarning, match="use the host reader"), pytest.raises(VsvError):
            bf.fetch_device(eng, "chr10")
        good = str(tmp_path / "named.bam")
        bam.write_bam(good, [("chr10", 1_000_000)], recs)
        with bam.BamFile(good) as bg:                          # the engine is still usable afterwards
            assert bg.fetch_device(eng, "chr10").n_records == 200


@pytest.mark.gpu
def test_device_reader_equals_host_reader_on_random_files(tmp_path, monkeypatch):
    """Random BAM files: 1-3 references, unmapped-placed and reference-less records, names up to 250 characters with 'hp1' / 'hp2'
    anywhere, repeated names, empty CIGARs, > 65535-op CIGARs in CG:B,I behind aux fields of every type, SA tags, members of
    17 bytes to 64 KiB, stored / fixed / dynamic deflate blocks, one to many reader windows: every array and the name table of
    the device reader equal the host reader's."""
    import struct
    from volcanosv_amd import bam
    from volcanosv_amd.engine import Engine
    rng = np.random.default_rng(606 + int(os.environ.get("VSV_FUZZ_SEED", "0")))          # soak runs: other seeds, more files

    def random_aux():
        out = b""
        for _ in range(int(rng.integers(0, 6))):
            tag = bytes(rng.choice(list(b"ABXYZnm"), 2).astype(np.uint8))
            if tag in (b"CG", b"SA"):
                continue
            ty = rng.choice(list("AcCsSiIfZHB"))
            if ty == "A":
                out += tag + b"A" + b"x"
            elif ty in "cC":
                out += tag + ty.encode() + struct.pack("<B", int(rng.integers(0, 128)))
            elif ty in "sS":
                out += tag + ty.encode() + struct.pack("<H", int(rng.integers(0, 30000)))
            elif ty in "iI":
                out += tag + ty.encode() + struct.pack("<I", int(rng.integers(0, 1 << 30)))
            elif ty == "f":
                out += tag + b"f" + struct.pack("<f", float(rng.random()))
            elif ty == "Z":
                out += tag + b"Z" + bytes(rng.integers(33, 127, int(rng.integers(0, 40))).astype(np.uint8)) + b"\0"
            elif ty == "H":
                out += tag + b"H" + b"1AE3" * int(rng.integers(0, 5)) + b"\0"
            else:
                sub, cnt = rng.choice(list("cCsSiIf")), int(rng.integers(0, 20))
                width = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
                out += tag + b"B" + sub.encode() + struct.pack("<I", cnt) + bytes(rng.integers(0, 256, cnt * width).astype(np.uint8))
        return out

    with Engine(0) as eng:
        for case in range(12 * int(os.environ.get("VSV_FUZZ_SCALE", "1"))):
            nref = int(rng.integers(1, 4))
            refs = [("chr%d" % (i + 1), 5_000_000) for i in range(nref)]
            recs = []
            for tid in range(nref):
                pos = np.sort(rng.integers(0, 4_000_000, int(rng.integers(1, 1500))))
                for p in pos:
                    kind = rng.random()
                    if kind < 0.02:
                        cig = []
                    elif kind < 0.03:
                        cig = [(0, 2), (1, 1)] * int(rng.integers(33000, 36000)) + [(0, 1)]
                    else:
                        cig = [(int(rng.choice([0, 1, 2, 4, 7, 8, 3])), int(rng.integers(1, 3000))) for _ in range(int(rng.integers(1, 60)))]
                    stem = "PS%d_%s_x" % (int(rng.integers(0, 300)), rng.choice(["hp1", "hp2", "hp3", "h", "hp1hp2", ""]))
                    name = (stem + "q" * int(rng.integers(0, 240)))[:250] if rng.random() < 0.1 else stem
                    r = dict(tid=tid, pos=int(p), qname=name, mapq=int(rng.integers(0, 61)),
                             flag=int(rng.choice([0, 16, 4, 256, 2048, 2064, 272])), cigar=cig, seq_len=int(rng.integers(0, 3000)), aux=random_aux())
                    if rng.random() < 0.2:
                        r["tags"] = {b"SA": "chr1,%d,+,100S200M,60,3;" % int(rng.integers(1, 10000))}
                    recs.append(r)
            for _ in range(int(rng.integers(0, 4))):                        # reads without a reference at the end of the file
                recs.append(dict(tid=-1, pos=-1, qname="unplaced%d" % len(recs), mapq=0, flag=4, cigar=[], seq_len=100))
            sizes = [17, 300, 5000, 60000, 65280]
            path = str(tmp_path / ("r%d.bam" % case))
            fixed = int(rng.choice(sizes[1:]))
            bam.write_bam(path, refs, recs, block_bytes=(lambda: int(rng.choice(sizes))) if case % 2 else fixed, level=int(rng.choice([0, 1, 6, 9])))
            need = 300_000 // (17 if case % 2 else fixed) + 20          # members the longest record (a 70 k-op CIGAR) can span
            monkeypatch.setenv("VSV_BAM_WINDOW", str(int(rng.choice([w for w in (32768, 2000, 300, 40) if w >= need]))))
            with bam.BamFile(path) as bf:
                for chrom in [None] + [refs[int(rng.integers(0, nref))][0]]:
                    host = bf.fetch_soa(chrom)
                    view = bf.fetch_device(eng, chrom, sa=True)
                    assert isinstance(view, bam.DeviceRecordView), (case, chrom)
                    dev = view.to_host()
                    assert dev.n_records == host.n_records and dev.n_ops == host.n_ops and view.n_qids == host.n_qids, (case, chrom)
                    assert list(view.sa_tags) == list(host.sa_tags), (case, chrom)
                    for name in ("pos", "tid", "qid", "cigar_off", "mapq", "flag", "cigar", "l_seq", "sam_flags"):
                        assert np.array_equal(getattr(host, name), getattr(dev, name)), (case, chrom, name)
                    assert list(host.qnames) == list(dev.qnames), (case, chrom)
        monkeypatch.delenv("VSV_BAM_WINDOW")


@pytest.mark.gpu
def test_device_reader_survives_corrupted_files():
    """The corrupted streams of tests/_corrupt_bam_worker.py through the device reader (own process: a GPU fault is an exit code):
    every file either parses to the host reader's arrays, or is handed to the host reader, or raises — no kernel reads outside a
    record (rec_fields checks that name, CIGAR, SEQ and a CG:B,I array fit their record before anything dereferences them)."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_corrupt_bam_device_worker.py")
    for seed in (5, 6):
        r = subprocess.run([sys.executable, worker, str(seed), "100"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (seed, r.returncode, r.stderr[-600:])
        f = r.stdout.split()
        same, raised = int(f[1]), int(f[5])
        assert same > 20 and raised > 20


@pytest.mark.gpu
def test_inflate_survives_corrupted_payloads():
    """Bit flips, truncations, garbage and lying sizes in raw-deflate payloads (tests/_corrupt_inflate_worker.py, own process): the
    GPU decoder equals zlib whenever zlib decodes to exactly the announced size and reports the member otherwise."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_corrupt_inflate_worker.py")
    for seed in (2, 3):
        r = subprocess.run([sys.executable, worker, str(seed), "150"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (seed, r.returncode, r.stderr[-600:])
        f = r.stdout.split()
        assert int(f[1]) > 30 and int(f[3]) > 30


@pytest.mark.gpu
def test_member_crc32_is_checked_on_the_gpu(tmp_path):
    """The gzip trailer's CRC-32 of every member computed on the GPU (bgzf_crc32: 64 slices per member, byte-wise table per lane,
    zero-byte shift operators to combine): equal to zlib.crc32 for member sizes 0..65536, a wrong value names its member, and a
    BAM whose payload was altered without touching the framing no longer loads through the device reader."""
    import zlib
    from volcanosv_amd import bam
    from volcanosv_amd.abi import VsvError
    from volcanosv_amd.engine import Engine
    rng = np.random.default_rng(12)
    sizes = [0, 1, 2, 63, 64, 65, 127, 128, 1000, 4095, 4096, 4097, 65535, 65536] + [int(x) for x in rng.integers(1, 65536, 40)]
    datas = [rng.integers(0, 256, n).astype(np.uint8).tobytes() if i % 2 else (b"ACGT" * (n // 4 + 1))[:n] for i, n in enumerate(sizes)]
    pay = []
    for d in datas:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        pay.append(c.compress(d) + c.flush())
    crcs = [zlib.crc32(d) & 0xFFFFFFFF for d in datas]
    with Engine(0) as eng:
        assert eng.bgzf_inflate(pay, sizes, crcs) == datas
        bad = list(crcs)
        bad[17] ^= 0x10
        with pytest.raises(VsvError, match="member 17 fails its CRC-32"):
            eng.bgzf_inflate(pay, sizes, bad)
        assert eng.bgzf_inflate(pay, sizes) == datas                        # without expectations nothing is checked
        # a BAM with one payload byte changed and the member re-deflated under the OLD trailer
        recs = [dict(tid=0, pos=100 * i, qname="r%d" % i, mapq=60, flag=0, cigar=[(0, 100)], seq_len=100) for i in range(2000)]
        path = str(tmp_path / "x.bam")
        bam.write_bam(path, [("chr1", 1_000_000)], recs, block_bytes=20000)
        raw = bytearray(open(path, "rb").read())
        import struct
        o, k = 0, 0
        while o < len(raw):
            xlen = struct.unpack_from("<H", raw, o + 10)[0]
            bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
            if k == 3:
                body = bytearray(zlib.decompress(bytes(raw[o + 12 + xlen:o + bsize - 8]), -15))
                body[len(body) // 2 + 8] ^= 1                                 # inside a record's name / sequence bytes
                c = zlib.compressobj(6, zlib.DEFLATED, -15)
                new = c.compress(bytes(body)) + c.flush()
                blk = bytes(raw[o:o + 16]) + struct.pack("<H", len(new) + 25) + new + bytes(raw[o + bsize - 8:o + bsize])   # old CRC, old size
                raw[o:o + bsize] = blk
                break
            o += bsize
            k += 1
        open(path, "wb").write(bytes(raw))
        with bam.BamFile(path) as bf:
            with pytest.raises(VsvError, match="CRC"):
                bf.fetch_device(eng, None)
        with bam.BamFile(path) as bf:
            with pytest.raises(VsvError):
                bf.fetch_soa(None)                                           # the host reader checks it too


@pytest.mark.gpu
def test_device_reader_on_the_reference_held_bam_fixtures():
    """svim-asm's own htslib-written fixtures (tests/golden/chimeric_read.bam, chimeric_read_errors.bam = the data files of
    svim-asm tests/test_satag.py) through BamFile.fetch_device(sa=True): files this repo's writer did not produce. Every array,
    the name table and the SA:Z texts equal the host reader's, and the SA texts satisfy test_satag.py:13-54 (the primary's tag
    reconstructs the three other alignments; an entry with too many fields is skipped; a negative MAPQ becomes 0)."""
    from volcanosv_amd import bam, bnd
    from volcanosv_amd.engine import Engine
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with Engine(0) as eng:
        for name in ("chimeric_read.bam", "chimeric_read_errors.bam"):
            with bam.BamFile(os.path.join(golden, name)) as bf:
                host = bf.fetch_soa(None)
                view = bf.fetch_device(eng, None, sa=True)
                assert isinstance(view, bam.DeviceRecordView)
                dev = view.to_host()
                assert dev.n_records == host.n_records > 0 and view.n_qids == host.n_qids
                for field in ("pos", "tid", "qid", "cigar_off", "mapq", "flag", "cigar", "l_seq", "sam_flags"):
                    assert np.array_equal(getattr(host, field), getattr(dev, field)), (name, field)
                assert list(host.qnames) == list(dev.qnames)
                assert list(view.sa_tags) == list(host.sa_tags), name
                tid_of = lambda n: host.tid_names.index(n)
                if name == "chimeric_read.bam":
                    assert host.n_records == 4
                    prim = [i for i in range(4) if not (dev.flag[i] & 2)]
                    sa = bnd.parse_sa(view.sa_tags[prim[0]], tid_of)
                    others = [i for i in range(4) if i != prim[0]]
                    assert len(sa) == 3
                    for (tid, pos0, rev, c2, mq), i in zip(sa, others):                 # test_satag.py:25-35
                        cig = [(int(w) & 15, int(w) >> 4) for w in dev.cigar[int(dev.cigar_off[i]):int(dev.cigar_off[i + 1])]]
                        assert int(dev.tid[i]) == tid and int(dev.pos[i]) == pos0 and c2 == cig
                        assert bool(dev.flag[i] & 1) == rev and int(dev.mapq[i]) == mq
                else:
                    assert len(bnd.parse_sa(view.sa_tags[0], tid_of)) == 2               # test_satag.py:36-45
                    neg = bnd.parse_sa(view.sa_tags[1], tid_of)
                    assert len(neg) == 1 and neg[0][4] == 0                              # test_satag.py:46-54


@pytest.mark.gpu
def test_sequences_kept_on_the_device_and_sliced_there(tmp_path, monkeypatch):
    """fetch_device(seq=True): the packed SEQ fields stay on the GPU and DeviceRecordView.seq_slices decodes Python-style slices of
    them (negative and overlong bounds, the reversed read of sig_extract's split INS, SE:215): equal to the slices of the host
    reader's query sequences, for whole files and for one chromosome of a file, odd and even lengths, reads without SEQ."""
    from volcanosv_amd import bam
    from volcanosv_amd.abi import VsvError
    from volcanosv_amd.engine import Engine
    rng = np.random.default_rng(4242)
    bases = np.frombuffer(b"=ACMGRSVTWYHKDBN", dtype=np.uint8)
    refs = [("chr1", 3_000_000), ("chr2", 2_000_000)]
    recs = []
    for tid in range(2):
        for p in np.sort(rng.integers(0, 1_500_000, 700)):
            l = int(rng.choice([0, 1, 2, 3, 63, 64, 65, int(rng.integers(100, 9000))]))
            seq = bases[rng.integers(0, 16, l)].tobytes().decode() if l else None
            recs.append(dict(tid=tid, pos=int(p), qname="r%d_hp%d" % (len(recs), 1 + len(recs) % 2), mapq=60, flag=0,
                             cigar=[(0, max(1, l))] if l else [(0, 10)], seq=seq))
    path = str(tmp_path / "seq.bam")
    bam.write_bam(path, refs, recs)
    with Engine(0) as eng, bam.BamFile(path) as bf:
        for chrom, window in ((None, None), ("chr2", None), (None, 40), ("chr2", 17)):      # one reader window, and many small ones
            if window:
                monkeypatch.setenv("VSV_BAM_WINDOW", str(window))
            host = bf.fetch_soa(chrom, keep_seq=True)
            view = bf.fetch_device(eng, chrom, seq=True)
            monkeypatch.delenv("VSV_BAM_WINDOW", raising=False)
            assert isinstance(view, bam.DeviceRecordView) and view.n_records == host.n_records
            reqs, want = [], []
            for _ in range(3000):
                r = int(rng.integers(0, host.n_records))
                l = int(host.l_seq[r])
                a, b = int(rng.integers(-l - 5, l + 6)), int(rng.integers(-l - 5, l + 6))
                rv = bool(rng.random() < 0.3)
                reqs.append((r, a, b, rv))
                q = host.seq[r]
                want.append((q[::-1] if rv else q)[a:b])
            for r in range(0, host.n_records, 97):                          # whole reads
                reqs.append((r, 0, int(host.l_seq[r]), False))
                want.append(host.seq[r])
            assert view.seq_slices(reqs) == want
            assert view.seq_slices([]) == []
        plain = bf.fetch_device(eng, "chr1")                                # without seq=True nothing is kept
        with pytest.raises(ValueError):
            plain.seq_slices([(0, 0, 1, False)])
        # the C-ABI refuses slices outside their record / of records that were not loaded
        view = bf.fetch_device(eng, "chr1", seq=True)
        import ctypes as C
        one = lambda v, dt: np.array([v], dtype=dt)
        out, off = np.zeros(64, np.uint8), np.zeros(1, np.uint64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        l0 = int(view.l_seq_host()[0])
        for rec, start, ln in ((0, l0, 1), (view.n_records, 0, 0), (0, 0, l0 + 1)):
            st = eng.lib.vsv_bam_device_seq_slices(eng.h, vp(one(rec, np.uint32)), vp(one(start, np.uint32)), vp(one(ln, np.uint32)), vp(one(0, np.uint8)), 1,
                                                   vp(off), vp(out), 64)
            assert st != 0


@pytest.mark.gpu
def test_staged_file_whose_size_is_not_slice_aligned(tmp_path):
    """The device reader stages a file of 8 MiB and more into a page-locked buffer of the handle with up to 16 reader threads, one
    slice each. A file of 16 * 4096 * k + r bytes (r = 1..15) used to leave its last r bytes to no reader (slice size rounded from
    floor(size / threads)), i.e. to whatever an earlier, larger file had left in the reused buffer: the trailer of the last
    member then failed its checks. Same handle, larger file first; every array equals the host reader's."""
    import struct
    from volcanosv_amd import bam
    from volcanosv_amd.engine import Engine

    def recs(n):
        return [dict(tid=0, pos=1000 + 50 * i, qname="PS%d_hp%d" % (i, 1 + i % 2), mapq=60, flag=0, cigar=[(0, 500), (1, 40 + i % 9), (0, 300)],
                     seq_len=30000) for i in range(n)]

    def padded_empty_member(total):
        """An empty BGZF member of exactly `total` bytes: the BC subfield plus a second extra subfield as ballast."""
        ballast = total - 28 - 4
        assert 0 <= ballast < 60000
        xlen = 6 + 4 + ballast
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", xlen) + b"BC\x02\x00" + struct.pack("<H", total - 1) +
                b"XX" + struct.pack("<H", ballast) + b"\x00" * ballast + b"\x03\x00" + struct.pack("<II", 0, 0))

    big, small = str(tmp_path / "big.bam"), str(tmp_path / "small.bam")
    bam.write_bam(big, [("chr1", 50_000_000)], recs(330), level=0)
    bam.write_bam(small, [("chr1", 50_000_000)], recs(200), level=0)
    assert os.path.getsize(big) > os.path.getsize(small) + (1 << 20) and os.path.getsize(small) > (8 << 20)
    with Engine(0) as eng:
        for r in (1, 7, 15):
            body = open(small, "rb").read()[:-28]                       # without the EOF marker
            want = -(-(len(body) + 32 + 28) // 65536) * 65536 + r       # 16 * 4096 * k + r
            path = str(tmp_path / ("small_%d.bam" % r))
            with open(path, "wb") as f:
                f.write(body + padded_empty_member(want - len(body) - 28) + bam._bgzf_block(b""))
            assert os.path.getsize(path) % 65536 == r
            with bam.BamFile(big) as bf:
                assert isinstance(bf.fetch_device(eng, "chr1"), bam.DeviceRecordView)
            with bam.BamFile(path) as bf:
                host = bf.fetch_soa("chr1")
                view = bf.fetch_device(eng, "chr1")
                assert isinstance(view, bam.DeviceRecordView)
                dev = view.to_host()
                for name in ("pos", "tid", "qid", "cigar_off", "mapq", "flag", "cigar", "l_seq"):
                    assert np.array_equal(getattr(host, name), getattr(dev, name)), (r, name)
                assert host.n_records == 200
