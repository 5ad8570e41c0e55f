"""Host mirror of Large_INDEL/FP_filter_v1.py — the false-positive filter Raw_variant_call.py:91-96 runs on the raw
variant VCF. Same function names and argument meaning as the reference; `eval_sig` runs on the GPU through
vsv_support_join (one wave per call over the position-sorted read signatures) instead of the Python double loop
(FP_filter_v1.py:106-123)."""
import numpy as np

from .engine import Engine


def vcflines_to_sig(vcf_lines, target_chr_name):
    """FP_filter_v1.py:59-75: [chr, 'DEL'|'INS', pos, |len(ALT)-len(REF)|] per non-header line of the chromosome."""
    sig_list = []
    for line in vcf_lines:
        if line[0] != '#':
            data = line.split()
            svlen = len(data[4]) - len(data[3])
            if data[0] == target_chr_name:
                sig_list.append([data[0], 'DEL' if svlen < 0 else 'INS', int(data[1]), abs(svlen)])
    return sig_list


def vcf_to_sig(vcf_path, target_chr_name):
    """FP_filter_v1.py:39-57."""
    with open(vcf_path, 'r') as f:
        return vcflines_to_sig(f.readlines(), target_chr_name)


def load_sig(sig_path):
    """FP_filter_v1.py:77-86: whitespace-split lines of <chr>_reads_sig.txt with fields 2, 3 (pos, svlen) as ints."""
    sig_list = []
    with open(sig_path, 'r') as f:
        for line in f:
            data = line.split()
            data[2], data[3] = int(data[2]), int(data[3])
            sig_list.append(data)
    return sig_list


def eval_sig(sig_list, reads_sig_list, max_dist, max_comp_svlen=300, max_shift=500, min_size_sim=0.3, engine=None):
    """FP_filter_v1.py:106-123 (defaults as there). Returns the support list (60 for calls above max_comp_svlen)."""
    eng = engine or Engine(0)
    try:
        p = eng.support_params(max_comp_svlen=max_comp_svlen, max_dist=max_dist, max_shift=max_shift, min_size_sim=float(min_size_sim))
        sup = eng.support_join(np.array([s[2] for s in sig_list], dtype=np.int32), np.array([s[3] for s in sig_list], dtype=np.int32),
                               np.array([s[2] for s in reads_sig_list], dtype=np.int32),
                               np.array([s[3] for s in reads_sig_list], dtype=np.int32), p)
    finally:
        if engine is None:
            eng.close()
    return [int(x) for x in sup]


def filter_vcf(vcf_lines, chr_name, sig_path, max_dist, max_comp_svlen, max_shift, min_size_sim, engine=None):
    """FP_filter_v1.py:135-147: the lines of `vcf_lines` whose call has support > 0."""
    sig_chr = vcflines_to_sig(vcf_lines, chr_name)
    reads_sig_list = load_sig(sig_path)
    support_list = eval_sig(sig_chr, reads_sig_list, max_dist, max_comp_svlen, max_shift, min_size_sim, engine=engine)
    assert len(support_list) == len(vcf_lines)
    support_list = np.array(support_list)
    print("reduced %d lines" % ((support_list == 0).sum()))
    return [vcf_lines[i] for i in np.where(support_list > 0)[0]]


def load_wgs_vcf(vcf_path):
    """FP_filter_v1.py:149-163: (header lines, {chrom: [lines]})."""
    header, dc = [], {}
    with open(vcf_path, 'r') as f:
        for line in f:
            if line[0] == '#':
                header.append(line)
            else:
                dc.setdefault(line.split()[0], []).append(line)
    return header, dc


def run(input_path, signature_dir, output_path, max_comp_svlen=250, max_dist=1000, max_shift=500, min_size_sim=0.5, device=0, engine=None):
    """The script body (FP_filter_v1.py:169-202): chr1..chr22 only, in that order; header copied through."""
    header, dc = load_wgs_vcf(input_path)
    eng = engine or Engine(device)
    final_lines = []
    try:
        for i in range(1, 23):
            chr_name = 'chr%d' % i
            if chr_name in dc:
                sig_path = signature_dir + "/%s_reads_sig.txt" % chr_name
                final_lines.append(filter_vcf(dc[chr_name], chr_name, sig_path, max_dist, max_comp_svlen, max_shift, min_size_sim, engine=eng))
            else:
                final_lines.append([])
    finally:
        if engine is None:
            eng.close()
    with open(output_path, 'w') as fw:
        fw.writelines(header)
        for i in range(22):
            chr_name = 'chr%d' % (i + 1)
            if chr_name in dc:
                print(chr_name, len(dc[chr_name]), len(final_lines[i]))
                fw.writelines(final_lines[i])
    return final_lines
