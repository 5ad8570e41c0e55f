"""Host driver of the reads path: extract_reads_signature.py (reference Large_INDEL/extract_reads_signature.py:268-286)."""
import os

from . import sigtable
from .abi import DTYPE_READS
from .bam import BamFile
from .engine import Engine, default_params


def run(input_path, output_dir, chr_number, device=0, engine=None, params=None, device_ingest=True):
    """Writes <output_dir>/reads_signature/chr<N>_reads_sig.txt (RS:251-265): tab-joined str() fields, sorted by pos; returns
    (path, number of lines).
    device_ingest: BAM inflated and parsed on the GPU (volcanosv_amd.bam.BamFile.fetch_device); False = host reader."""
    chrom = "chr%d" % chr_number
    out_dir = os.path.join(output_dir, "reads_signature")
    os.makedirs(out_dir, exist_ok=True)
    eng = engine or Engine(device, grow=True)
    p = params or default_params(DTYPE_READS)
    with BamFile(input_path) as bam:
        if device_ingest:
            view = bam.fetch_device(eng, chrom)
            soa = view.host_light()
            if view.n_records:
                eng.run(view, p)
            else:
                device_ingest = False
                soa = bam.fetch_soa(chrom)
        else:
            soa = bam.fetch_soa(chrom)
    if not device_ingest:
        eng.run(soa, p)
    lines = sigtable.reads_sig_lines(soa, eng.table("reads"))
    for fname, text in sigtable.reads_dump_texts(soa, eng.table("cigar"), eng.table("split"), chrom).items():   # RS:130-131, 242-243
        with open(os.path.join(out_dir, fname), "w") as f:
            f.write(text)
    path = os.path.join(out_dir, chrom + "_reads_sig.txt")
    with open(path, "w") as f:
        f.writelines(lines)
    if engine is None:
        eng.close()
    return path, len(lines)
