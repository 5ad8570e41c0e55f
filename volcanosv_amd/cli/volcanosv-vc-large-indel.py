#!/usr/bin/env python3
"""Drop-in for Large_INDEL/volcanosv-vc-large-indel.py (same flags, plus --gpus). Chromosomes fan out over GPUs (one
Raw_variant_call.py process per chromosome, LPT over the visible devices) instead of joblib over CPU cores: at most
--n_thread of them run at a time (the reference's joblib.Parallel(n_jobs=n_thread), volcanosv-vc-large-indel.py:268), each
GPU working down its own queue; a chromosome that fails stops the run with its command. The signature filter and the
genotype correction that follow (filter_GT_correction.py) run in this process on the GPU."""
import argparse
import os
import subprocess
import sys
from argparse import ArgumentParser

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from volcanosv_amd import pipeline, shard  # noqa: E402

p = ArgumentParser(description="", usage='use "python3 %(prog)s --help" for more information', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
p.add_argument('--input_dir', '-i')
p.add_argument('--output_dir', '-o')
p.add_argument('--data_type', '-dtype', help='Hifi;CLR;ONT')
p.add_argument('--bam_file', '-bam')
p.add_argument('--reference', '-ref')
p.add_argument('--read_signature_dir', '-rdsig')
p.add_argument('--pre_cutesig', '-presig')
p.add_argument('--chr_num', '-chr', type=int, default=None)
p.add_argument('--n_thread', '-t', type=int, default=11)
p.add_argument('--n_thread_align', '-ta', type=int, default=10)
p.add_argument('--mem_per_thread', '-mempt', default='768M')
p.add_argument('--prefix', '-px', default="Sample")
p.add_argument('--gpus', type=int, default=1, help="GPUs to spread the chromosomes over")
a = p.parse_args()
here = os.path.dirname(os.path.abspath(__file__))
out = a.output_dir
out_chr = out + "/chr%d/" % a.chr_num if a.chr_num is not None else out
os.makedirs(out_chr, exist_ok=True)
header_file = out_chr + "/VCF_header"
pipeline.generate_vcf_header(a.reference, header_file, a.chr_num, a.prefix)
ref_dir = out_chr + "/ref_by_chr/"
pipeline.split_reference(a.reference, ref_dir, a.chr_num)
chroms = [a.chr_num] if a.chr_num is not None else list(range(1, 23))
fai = {l.split("\t")[0]: int(l.split("\t")[1]) for l in open(a.reference + ".fai")}
owner = shard.lpt_assign([fai.get("chr%d" % c, 1) for c in chroms], max(1, a.gpus))
import shlex  # noqa: E402
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

n_gpus = max(1, a.gpus)
# every process runs minimap2 -t n_thread_align + samtools sort and holds GPU contexts: n_thread bounds how many run at once
# (spread evenly over the GPUs, at least one per GPU)
per_gpu = max(1, max(1, a.n_thread) // n_gpus)


def run_chromosome(c, g):
    fasta = a.input_dir + "/chr%d/assembly/final_contigs/%s_final_contigs.fa" % (c, a.prefix)
    cmd = [sys.executable, os.path.join(here, "Raw_variant_call.py"), "-contig", fasta, "-ref", ref_dir + "/chr%d.fa" % c]
    cmd += ["-sigd", a.read_signature_dir] if a.read_signature_dir else ["-rbam", a.bam_file]
    cmd += ["-o", out + "/chr%d/" % c, "-dtype", a.data_type, "-t", str(a.n_thread_align), "-chr", str(c), "-header", header_file]
    rc = subprocess.call(cmd, env=dict(os.environ, HIP_VISIBLE_DEVICES=str(g)))
    return c, rc, " ".join(shlex.quote(x) for x in cmd)


failed = []
# one queue per GPU (its chromosomes by shard.lpt_assign, largest first) worked down by per_gpu threads: at most per_gpu processes
# (each with its aligner and its GPU buffers) share a device, and no GPU idles while another one holds several jobs
queues = [sorted([c for c, o in zip(chroms, owner) if o == g], key=lambda c: -fai.get("chr%d" % c, 1)) for g in range(n_gpus)]
with ThreadPoolExecutor(max_workers=n_gpus * per_gpu) as pool:
    futs = []
    for g, q in enumerate(queues):
        it = iter(q)
        lock = __import__("threading").Lock()

        def worker(g=g, it=it, lock=lock):
            res = []
            while True:
                with lock:
                    c = next(it, None)
                if c is None:
                    return res
                res.append(run_chromosome(c, g))
        futs += [pool.submit(worker) for _ in range(min(per_gpu, max(1, len(q))))]
    for f in futs:
        for c, rc, cmd in f.result():
            if rc != 0:
                failed.append((c, rc, cmd))
if failed:
    for c, rc, cmd in failed:
        print("chr%d: Raw_variant_call.py exited with status %d: %s" % (c, rc, cmd), file=sys.stderr)
    sys.exit(1)
body, header = [], open(header_file).readlines()
for c in chroms:
    path = out + '/chr%d/final_vcf/volcano_variant_no_redundancy.vcf' % c
    body += [l for l in open(path) if l[0] != '#']
raw_vcf = out + ("/raw_variants_wgs.vcf" if a.chr_num is None else "/chr%d/final_vcf/volcano_variant_no_redundancy.vcf" % a.chr_num)
if a.chr_num is None:
    open(raw_vcf, 'w').writelines(header + body)
# filter_GT_correction.py (DRV: the step after the per-chromosome calls): signatures, coverage band, genotype correction
from volcanosv_amd import gt_correction  # noqa: E402
infile = gt_correction.filter_gt_correction(raw_vcf, a.bam_file, a.reference, a.pre_cutesig, a.data_type, a.chr_num)
outfile = (out_chr + "/%s_volcanosv_large_indel_chr%d.vcf" % (a.prefix, a.chr_num)) if a.chr_num is not None else out + "/%s_volcanosv_large_indel.vcf" % a.prefix
open(outfile, "w").writelines(pipeline.phase_large_indel(open(infile).readlines(), header))
print("wrote", outfile)
