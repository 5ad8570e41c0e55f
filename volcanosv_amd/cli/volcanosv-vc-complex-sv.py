#!/usr/bin/env python3
"""Drop-in for Complex_SV/volcanosv-vc-complex-sv.py (same flags), breakend (TRA) branch on the GPU. DUP recovery
(align_ins2ref.py) and INV filtering (filter_inv.py) are outside this build and are spawned from VOLCANOSV_COMPLEX_CODE_DIR
when configured."""
import glob
import os
import subprocess
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import pipeline  # noqa: E402

p = ArgumentParser()
p.add_argument('--input_dir', '-i')
p.add_argument('--indelvcf', '-vcf')
p.add_argument('--bam_file', '-bam')
p.add_argument('--reference', '-ref')
p.add_argument('--data_type', '-dtype', choices=['Hifi', 'CLR', 'ONT'])
p.add_argument('--output_dir', '-o')
p.add_argument('--n_thread', '-t', type=int, default=22)
p.add_argument('--prefix', '-px', default="Sample")
a = p.parse_args()
here = os.path.dirname(os.path.abspath(__file__))
out, raw = a.output_dir, a.output_dir + "/Raw_Detection/"
os.makedirs(raw, exist_ok=True)
# merge_fasta (volcanosv-vc-complex-sv.py:41-65): contigs whose name carries hp1 / hp2 go to hp1.fa / hp2.fa
with open(out + "/hp1.fa", "w") as f1, open(out + "/hp2.fa", "w") as f2:
    for path in sorted(glob.glob(a.input_dir + "/chr*/assembly/final_contigs/*_final_contigs.fa")):
        fw = None
        for line in open(path):
            if line[0] == '>':
                fw = f1 if 'hp1' in line else f2
            fw.write(line)
for hp in (1, 2):
    pipeline.align_contigs(a.reference, out + "/hp%d.fa" % hp, raw + "assembly_hp%d.bam" % hp, "asm10", a.n_thread, "4G")
subprocess.check_call([sys.executable, here + "/svim_asm_bnd.py", "diploid", raw, raw + "assembly_hp1.bam", raw + "assembly_hp2.bam", a.reference, "--query_names"])
subprocess.check_call([sys.executable, here + "/filter_tra.py", "-vcf", raw + "variants.vcf", "-o", out + "/TRA/", "-bam", str(a.bam_file)])
code = os.environ.get("VOLCANOSV_COMPLEX_CODE_DIR")
pipeline.spawn_reference_script(code, "align_ins2ref.py", ["-i", a.indelvcf, "-o", out + "/DUP", "-d", a.data_type, "-ref", a.reference, "-t", a.n_thread])
pipeline.spawn_reference_script(code, "filter_inv.py", ["-vcf", raw + "/variants.vcf", "-o", out + "/INV/", "-bam", a.bam_file])
lines = [l for l in open(out + "/TRA/TRA_final.vcf") if l[0] == '#']
for part in ("DUP/DUP_final.vcf", "TRA/TRA_final.vcf", "INV/INV_final.vcf"):
    if os.path.exists(out + "/" + part):
        lines += [l for l in open(out + "/" + part) if l[0] != '#']
lines = [l.replace("svim_asm", "volcanosv").replace("SVIM-asm-v1.0.2", "VolcanoSV") for l in lines]   # the sed of :158-159
open(out + "/variants.vcf", "w").writelines(lines)
outfile = out + "/%s_volcanosv_complex_SV.vcf" % a.prefix
open(outfile, "w").writelines(pipeline.phase_complex(lines))
print("wrote", outfile)
