#!/usr/bin/env python3
"""Drop-in for Large_INDEL/Raw_variant_call.py (same flags). Only the aligner stays external: the contig and reads signature
extraction, the false-positive filter (FP_filter_v1.py) and the redundancy removal (remove_redundancy.py) run on the GPU in
this process."""
import os
import shutil
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import contig_signature, fp_filter, pipeline, reads_signature, remove_redundancy, vcf  # noqa: E402

parser = ArgumentParser(description="", usage='use "python3 %(prog)s --help" for more information')
parser.add_argument('--contig_path', '-contig')
parser.add_argument('--reference_path', '-ref')
parser.add_argument('--signature_dir', '-sigd')
parser.add_argument('--rbam_file', '-rbam')
parser.add_argument('--output_dir', '-o')
parser.add_argument('--data_type', '-dtype', help="Hifi;CLR;ONT")
parser.add_argument('--chr_num', '-chr')
parser.add_argument('--header_file', '-header')
parser.add_argument('--n_thread', '-t', type=int, default=10)
parser.add_argument('--mem_per_thread', '-mempt', default='1G')
a = parser.parse_args()
os.makedirs(a.output_dir, exist_ok=True)
prefix = a.contig_path.split('/')[-1].split('.')[0]
bam = pipeline.align_contigs(a.reference_path, a.contig_path, a.output_dir + '/' + prefix + '.sorted.bam', "asm5", a.n_thread, a.mem_per_thread)
chr_num = int(a.chr_num) if a.chr_num not in (None, "None") else None
per_chr = contig_signature.run(a.data_type, bam, a.contig_path, a.reference_path, a.output_dir, chr_num, a.header_file)
header = open(a.header_file).readlines() if a.header_file else vcf.default_header()
raw = a.output_dir + "/volcano_raw_variant.vcf"
vcf.write_vcf(raw, header, [l for lines in per_chr.values() for l in lines])          # Raw_variant_call.py:77-80
sigd = a.signature_dir
if sigd is None and a.rbam_file and chr_num is not None:
    reads_signature.run(a.rbam_file, a.output_dir, chr_num)                            # :83-88
    sigd = a.output_dir + "/reads_signature/"
filtered = a.output_dir + "/volcano_variant_filtered.vcf"
final_dir = a.output_dir + '/final_vcf/'
if sigd is not None:
    fp_filter.run(raw, sigd, filtered)                                                 # :91-96
else:
    shutil.copy(raw, filtered)
    print("note: no reads signature directory (-sigd / -rbam with -chr): FP filter skipped")
remove_redundancy.run(filtered, final_dir)                                            # :99-104
