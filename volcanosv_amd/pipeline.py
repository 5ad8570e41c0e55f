"""Host glue around the hot path: the parts of the reference's L4/L5 drivers that the CLI contract needs
(Large_INDEL/volcanosv-vc-large-indel.py, Raw_variant_call.py; Complex_SV/volcanosv-vc-complex-sv.py).
The aligner stays external (minimap2 | samtools sort, spawned where the reference spawns it). The post-filters of the
Large_INDEL calling stage (FP_filter_v1.py, remove_redundancy.py, calculate_signature_support.py, filter_GT_correction.py)
run in-process on the GPU (fp_filter.py, remove_redundancy.py, signature_support.py, gt_correction.py); only the
Complex_SV extras that are out of this build's scope (align_ins2ref.py, filter_inv.py) can still be spawned from a
reference checkout, and only when VOLCANOSV_CODE_DIR names one (off by default)."""
import os
import shlex
import shutil
import subprocess

from . import vcf

PS_INFO = '##INFO=<ID=PS,Number=.,Type=Integer,Description="phase block name">\n'


def have(tool):
    return shutil.which(tool) is not None


def write_fai(fasta):
    """samtools-faidx compatible index (name, length, offset, linebases, linewidth) — volcanosv-vc-large-indel.py:77-91
    shells out to `samtools faidx`; this keeps the driver usable where samtools is absent."""
    fai = fasta + ".fai"
    if os.path.exists(fai):
        return fai
    rows, name, length, offset, lb, lw, pos = [], None, 0, 0, 0, 0, 0
    with open(fasta, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if name is not None:
                    rows.append((name, length, offset, lb, lw))
                name, length, offset, lb, lw = line[1:].split()[0].decode(), 0, pos + len(line), 0, 0
            else:
                if lw == 0:
                    lw, lb = len(line), len(line.rstrip(b"\r\n"))
                length += len(line.rstrip(b"\r\n"))
            pos += len(line)
    if name is not None:
        rows.append((name, length, offset, lb, lw))
    with open(fai, "w") as f:
        for r in rows:
            f.write("%s\t%d\t%d\t%d\t%d\n" % r)
    return fai


def generate_vcf_header(reference_fa, header_file, chr_num, prefix):
    """volcanosv-vc-large-indel.py:104-131: ##fileformat + ##contig lines from the .fai + header_info with the sample name."""
    contigs = [l.split("\t")[:2] for l in open(write_fai(reference_fa))]
    head = "##fileformat=VCFv4.2\n"
    for name, length in contigs:
        if chr_num is None or name == "chr" + str(chr_num):
            head += "##contig=<ID=%s,length=%s>\n" % (name, length)
    info = "".join('##INFO=<ID=%s,Number=%s,Type=%s,Description="%s">\n' % t for t in vcf._INFO) + PS_INFO
    info += '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n' % prefix
    with open(header_file, "w") as f:
        f.write(head + info)
    return head


def split_reference(input_path, output_dir, chr_num):
    """volcanosv-vc-large-indel.py:137-151: chr<i>.fa for the first 22 records of the reference FASTA."""
    os.makedirs(output_dir, exist_ok=True)
    chunks = open(input_path).read().split(">")[1:]
    for i in range(min(22, len(chunks))):
        if chr_num is None or i + 1 == chr_num:
            path = os.path.join(output_dir, "chr%d.fa" % (i + 1))
            with open(path, "w") as f:
                f.write(">" + chunks[i])
            write_fai(path)


def phase_large_indel(lines, header):
    """phase_vcf of volcanosv-vc-large-indel.py:202-231: PS from the first TIG_REGION contig name (PS<id>_hp<k>_...),
    0/1 -> 1|0 (hp1) or 0|1 (hp2), 1/1 -> 1|1."""
    out = list(header)
    for line in lines:
        if line[0] == "#":
            continue
        d = line.split()
        hps = d[7].split("TIG_REGION=")[1].split(";")[0].split(",")
        ps = hps[0].split("_")[0][2:]
        gt = ("1|0" if "hp1" in hps[0] else "0|1") if d[-1] == "0/1" else "1|1"
        d[7] += ";PS=" + ps
        d[-1] = gt
        out.append("\t".join(d) + "\n")
    return out


def phase_complex(lines):
    """phase_vcf of volcanosv-vc-complex-sv.py:67-97: PS from the first READS name; '/' -> '|' for 0/1, 1/0, 1/1."""
    header = [l for l in lines if l[0] == "#"]
    body = [l for l in lines if l[0] != "#"]
    header[-6] = header[-6] + PS_INFO
    out = list(header)
    for line in body:
        d = line.split()
        ps = d[7].split("READS=")[1].split(";")[0].split(",")[0].split("_")[0][2:]
        gt = d[-1].split(":")[0]
        d[7] += ";PS=" + ps
        d[-2] = "GT"
        d[-1] = gt.replace("/", "|") if gt in ("0/1", "1/0", "1/1") else gt
        out.append("\t".join(d) + "\n")
    return out


def align_contigs(reference, contigs, bam_out, preset, threads, mem="1G"):
    """minimap2 -a -x <preset> --cs -r2k | samtools sort ; samtools index (Raw_variant_call.py:49-58). Uses an existing
    BAM if the tools are missing. A failing aligner or sort raises (the reference ignores the status and calls variants on
    whatever the redirection left behind): the pipe runs under pipefail into a temporary file that only a clean run renames."""
    if have("minimap2") and have("samtools"):
        tmp = bam_out + ".tmp.%d" % os.getpid()
        q = shlex.quote
        cmd = "set -o pipefail; minimap2 -a -x %s --cs -r2k -t %d %s %s | samtools sort -@ %d -m %s > %s" % (
            q(preset), threads, q(reference), q(contigs), threads, q(mem), q(tmp))
        try:
            rc = subprocess.call(["bash", "-c", cmd])
            if rc != 0:
                raise RuntimeError("minimap2 | samtools sort exited with status %d: %s" % (rc, cmd))
            os.replace(tmp, bam_out)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        rc = subprocess.call(["samtools", "index", bam_out])
        if rc != 0:
            raise RuntimeError("samtools index %s exited with status %d" % (bam_out, rc))
    if not os.path.exists(bam_out):
        raise FileNotFoundError("%s not found and minimap2/samtools are not on PATH (the aligner is external to this build)" % bam_out)
    return bam_out


def spawn_reference_script(code_dir, script, args):
    """Runs one of the reference's own scripts from a checkout named by VOLCANOSV_CODE_DIR (off unless it is set): only the
    Complex_SV extras outside this build's scope use it. `args`: list of arguments. Returns False when no checkout is
    configured; a script that fails raises."""
    path = os.path.join(code_dir or "", script)
    if code_dir and os.path.exists(path):
        argv = ["python3", path] + (shlex.split(args) if isinstance(args, str) else [str(x) for x in args])
        rc = subprocess.call(argv)
        if rc != 0:
            raise RuntimeError("%s exited with status %d" % (" ".join(shlex.quote(x) for x in argv), rc))
        return True
    return False
