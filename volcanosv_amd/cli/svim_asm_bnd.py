#!/usr/bin/env python3
"""Drop-in for the `svim-asm diploid <dir> <hp1.bam> <hp2.bam> <ref> --query_names` call of
Complex_SV/volcanosv-vc-complex-sv.py:124-126, restricted to the breakend (BND) branch this build accelerates:
writes <dir>/variants.vcf with the svim header shape and the BND records (two lines per breakend, natural sort,
ids svim_asm.BND.<n>). INS/DEL/INV/DUP haplotype pairing (edlib) is out of scope."""
import os
import sys
import time
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import bnd  # noqa: E402
from volcanosv_amd.bam import BamFile  # noqa: E402
from volcanosv_amd.engine import Engine  # noqa: E402

ap = ArgumentParser()
ap.add_argument("sub", choices=["diploid"])
ap.add_argument("working_dir")
ap.add_argument("bam_file1")
ap.add_argument("bam_file2")
ap.add_argument("genome")
ap.add_argument("--query_names", action="store_true")
ap.add_argument("--min_mapq", type=int, default=20)
ap.add_argument("--sample", default="Sample")
args = ap.parse_args()

reads = []
contigs = None
eng = Engine(0)
for hap, path in ((1, args.bam_file1), (2, args.bam_file2)):
    with BamFile(path) as b:
        if contigs is None:
            contigs = list(zip(b.references, b.lengths))
        view = b.fetch_device(eng, None, sa=True)          # inflated and parsed on the GPU, SA:Z texts included
        if hasattr(view, "to_host"):
            soa = view.to_host()
            soa.sa_tags = view.sa_tags
        else:
            soa = view                                       # the device reader gave up: this is the host reader's table
    reads += bnd.segments_from_soa(soa, hap, args.min_mapq)
seg = bnd.SegmentSoA(reads, contigs)
cand, calls = eng.bnd(seg)
eng.close()
os.makedirs(args.working_dir, exist_ok=True)
with open(os.path.join(args.working_dir, "variants.vcf"), "w") as f:   # header: SVIM_COMBINE.py:394-425
    f.writelines(bnd.vcf_header(contigs, args.query_names, args.sample))
    for line in bnd.vcf_lines(seg, calls, args.query_names):
        f.write(line + "\n")
print("%d split contigs -> %d breakend candidates -> %d calls" % (len(reads), len(cand), len(calls)))
