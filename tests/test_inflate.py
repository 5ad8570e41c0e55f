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
    members = []
    for i, d in enumerate(blobs):
        for level, strat in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                             (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            if (i + level) % 2 == 0 or i < 8:
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
