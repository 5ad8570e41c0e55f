"""Host driver of the contig path: the per-chromosome body of extract_contig_signature_<dtype>.py
(reference Large_INDEL/extract_contig_signature_Hifi.py:725-772) on top of the HIP engine."""
import os

import numpy as np

from . import sigtable, vcf
from .abi import DTYPE_BY_NAME
from .bam import BamFile
from .engine import Engine, default_params


def run(dtype_name, bam_path, contig_path, ref_path, output_dir, chr_number=None, header_path=None, params=None, device=0,
        engine=None, log=print, device_ingest=True):
    """Writes <output_dir>/volcano_variant_chr<N>.vcf for chr_number (or chr1..22, H:729-742). Returns {chrom: lines}.
    device_ingest: the BAM is inflated and parsed on the GPU (the record SoA never visits the host; only names, flags and
    mapq come back for the VCF text); False uses the host reader."""
    dtype = DTYPE_BY_NAME[dtype_name]
    os.makedirs(os.path.join(output_dir, "signature"), exist_ok=True)          # H:717-718
    dc_contig = vcf.load_contigs(contig_path)
    dc_ref = vcf.load_contigs(ref_path)
    if header_path:
        with open(header_path) as f:
            header = f.readlines()
    else:
        header = vcf.default_header()
    chroms = [chr_number] if chr_number is not None else list(range(1, 23))
    # two engines (handle + HIP stream each) alternate over the chromosomes: the GPU works on chromosome i+1 while the host
    # formats the VCF of chromosome i, and the latency-bound stages of one overlap the CIGAR scan of the other
    engs = [engine] if engine is not None else [Engine(device, grow=True), Engine(device, stream=_side_stream(device), grow=True)]
    p = params or default_params(dtype)
    out = {}

    def drain(job):
        i, name, soa, eng = job
        eng.finish()
        calls, merged = eng.table("calls"), eng.table("merged")
        ref_seq = dc_ref[name] if name in dc_ref else next(iter(dc_ref.values()))   # ONT/CLR: single-chromosome FASTA (O:653-662)
        lines = vcf.vcf_lines(soa, calls, merged, ref_seq, dc_contig)
        # the per-source signature lists the reference leaves next to the VCF (H:402-405, 459-462); nothing in the pipeline reads them
        for fname, text in sigtable.signature_dump_texts(soa, eng.table("cigar"), eng.table("split"), name).items():
            with open(os.path.join(output_dir, "signature", fname), "w") as f:
                f.write(text)
        vcf.write_vcf(os.path.join(output_dir, "volcano_variant_chr%d.vcf" % i), header, lines)
        log("%s: %d records -> %d calls (%d written)" % (name, soa.n_records, len(calls), len(lines)))
        out[name] = lines

    try:
        with BamFile(bam_path) as bam:
            pending = []
            whole, ranges = None, {}
            if device_ingest and len(chroms) > 1:
                # several chromosomes: inflate and parse the file ONCE, then hand each chromosome's record range to an engine
                # (a coordinate-sorted BAM keeps a reference's records together); both engines read the same device arrays
                whole = bam.fetch_device(engs[0], None)
                tids = whole.host_light().tid
                if len(tids) and (np.diff(tids) < 0).any():
                    raise ValueError("BAM is not sorted by reference: use device_ingest=False")
                for t in np.unique(tids):
                    lo, hi = np.searchsorted(tids, t, side="left"), np.searchsorted(tids, t, side="right")
                    ranges[int(t)] = (int(lo), int(hi))
            for k, i in enumerate(chroms):
                name = "chr%d" % i
                if bam.get_tid(name) < 0:
                    raise KeyError("%s not in the BAM header" % name)
                eng = engs[k % len(engs)]
                while len(pending) >= len(engs):
                    drain(pending.pop(0))
                if device_ingest:
                    if whole is not None:
                        lo, hi = ranges.get(bam.get_tid(name), (0, 0))
                        view = whole.slice_records(lo, hi)
                    else:
                        view = bam.fetch_device(eng, name)
                    view.max_pos = bam.lengths[bam.get_tid(name)] + 100000
                    view.only_tid = bam.get_tid(name)
                    soa = view.host_light()
                    if view.n_records:
                        eng.run_async(view, p)
                    else:
                        eng.run_async(_EMPTY, p)
                    soa._view = view                                     # keeps the device arrays' owner alive
                else:
                    soa = bam.fetch_soa(name)
                    soa.max_pos = bam.lengths[bam.get_tid(name)] + 100000   # sort-key hints: positions never exceed the contig,
                    soa.only_tid = bam.get_tid(name)                        # and every record has this tid
                    eng.run_async(soa, p)
                pending.append((i, name, soa, eng))
            while pending:
                drain(pending.pop(0))
    finally:
        if engine is None:
            for e in engs:
                e.close()
    return out


def _empty_soa():
    import numpy as np
    from .soa import RecordSoA
    return RecordSoA(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.uint32), np.zeros(1, np.uint64), np.zeros(0, np.uint8),
                     np.zeros(0, np.uint8), np.zeros(0, np.uint32))


_EMPTY = _empty_soa()


def _side_stream(device):
    """A second HIP stream on `device`, created through torch (kept alive for the life of the process)."""
    import torch
    s = torch.cuda.Stream(device=device)
    _STREAMS.append(s)
    return s.cuda_stream


_STREAMS = []
