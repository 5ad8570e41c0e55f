"""Host mirror of Large_INDEL/correct_gt_del_real_data.py (DG) and correct_gt_ins_real_data.py (IG), the genotype-correction
half of filter_GT_correction.py (FGT:150-170): for every call, the reads that carry a matching signature (n_support) over the
reads that span the locus (n_cov) decide between 0/1 and 1/1 against per-platform thresholds.

The two joins run on the GPU: vsv_gt_support (window sums over the distinct read signatures) and vsv_span_count (reads with
start < a and end > b over the record SoA of the reads BAM, which the device reader leaves in HBM). The reference's
"resume at the last match" index — visited by both its forward and its backward scan, hence counted twice when it lies inside
the window — is replayed on the host from the window bounds. Text (VCF, .sigs, pandas TSVs) stays on the host."""
import os
from collections import OrderedDict

import numpy as np

# Large_INDEL/para/GT_correction_para_<dtype>_<vtype>.txt: t_large_11, t_small_11, t_large_01, t_small_01
PARA = {
    ("CLR", "DEL"): (0.55, 0.59, 0.65, 0.75), ("CLR", "INS"): (np.nan, np.nan, 0.64, 0.75),
    ("Hifi", "DEL"): (0.6, 0.69, 0.71, 0.91), ("Hifi", "INS"): (np.nan, 0.59, 0.63, 0.79),
    ("ONT", "DEL"): (0.61, 0.61, 0.68, 0.79), ("ONT", "INS"): (np.nan, np.nan, 0.67, 0.72),
}


# ---- loaders ---------------------------------------------------------------------------------------------------------------
def load_vcf_del(vcffile):
    """DG:39-53: [(gt, |SVLEN|, line)] of the SVTYPE=DEL lines."""
    out = []
    with open(vcffile) as f:
        for line in f:
            if line[0] != '#' and 'SVTYPE=DEL' in line:
                out.append((line.split()[-1].split(':')[0], abs(int(line.split('SVLEN=')[1].split(';')[0])), line))
    return out


def load_sig_del(sigfile):
    """DG:65-86: distinct (chrom, pos, svlen) in first-appearance order with the number of reads."""
    dc = OrderedDict()
    with open(sigfile) as f:
        for line in f:
            _, chrom, pos, svlen, rname = line.split()
            key = (chrom, int(pos), int(svlen))
            dc[key] = dc.get(key, 0) + 1
    return [k + (n,) for k, n in dc.items()]


def load_vcf_ins(vcffile):
    """IG:36-57: [[chrom number, pos, SVLEN, gt, svid]] of the SVTYPE=INS lines."""
    out = []
    with open(vcffile) as f:
        for line in f:
            if line[0] != '#' and 'SVTYPE=INS' in line:
                data = line.split()
                out.append([int(data[0][3:]), int(data[1]), int(line.split('SVLEN=')[1].split(';')[0]), data[-1].split(':')[0], data[2]])
    return out


def load_sig_file_ins(sig_file):
    """IG:66-99: {(chrom number, pos, svlen): reads} for svlen >= 30 on numbered chromosomes; also writes <sig_file>.gte30auto."""
    dc = OrderedDict()
    with open(sig_file) as f:
        for line in f:
            _, chrom, pos, svlen, rname = line.split()[:5]
            pos, svlen = int(pos), int(svlen)
            if svlen >= 30:
                try:
                    key = (int(chrom[3:]), pos, svlen)
                except ValueError:
                    continue
                dc[key] = dc.get(key, 0) + 1
    with open(sig_file + '.gte30auto', 'w') as f:
        for (chrom, pos, svlen), n in dc.items():
            f.write(f'{chrom}\t{pos}\t{svlen}\t{n}\n')
    return dc


# ---- the two joins ---------------------------------------------------------------------------------------------------------
def signature_support(eng, var_chrom, var_pos, var_svlen, sig_list, max_shift_ratio=2.3, min_size_sim=0.6):
    """match_varlist_siglist (DG:92-137) / extract_sig_support (IG:105-156): (support per variant, resume index per variant).
    sig_list = [(chrom, pos, svlen, reads)] in file order: one block per chromosome, positions ascending inside a block."""
    blocks, pos = {}, np.array([s[1] for s in sig_list], dtype=np.int64)
    for j, s in enumerate(sig_list):
        if s[0] not in blocks:
            blocks[s[0]] = [j, j + 1]
        elif blocks[s[0]][1] == j:
            blocks[s[0]][1] = j + 1
        else:
            raise ValueError("signature file is not grouped by chromosome (sort -k2,2 -k3,3n expected)")
    for lo, hi in blocks.values():
        if (np.diff(pos[lo:hi]) < 0).any():
            raise ValueError("signature positions do not ascend inside a chromosome block")
    blk = [blocks.get(c, (0, 0)) for c in var_chrom]
    svl = np.array([s[2] for s in sig_list], dtype=np.int32)
    cnt = np.array([s[3] for s in sig_list], dtype=np.int32)
    sums, lo, hi = eng.gt_support(var_pos, var_svlen, [b[0] for b in blk], [b[1] for b in blk], pos, svl, cnt, max_shift_ratio, min_size_sim)
    support, match, last = [], [], 0
    for v in range(len(var_pos)):
        match.append(last)
        s = int(sums[v])
        if lo[v] <= last < hi[v]:                                       # visited by the forward AND the backward scan
            svlen = var_svlen[v]
            if svlen * min_size_sim <= svl[last] <= svlen / min_size_sim:
                s += int(cnt[last])
        if hi[v] > lo[v]:
            last = int(lo[v])
        support.append(s)
    return support, match


def spanning_reads(eng, recs, tid_of, chroms, a, b):
    """count_reads_span_region (DG:140-147): reads with reference_start < a and reference_end > b, per query."""
    tids = [tid_of.get(c, -1) for c in chroms]
    return eng.span_count(recs, tids, a, b)


def depth_del(eng, recs, tid_of, chroms, pos, svlen):
    """check_full_cover_reads of the DEL script (DG:149-170): spanning reads of the deletion, or the mean of two 100-bp probes
    150 bp outside it when it is longer than 1000 bp."""
    qc, qa, qb, where = [], [], [], []
    for i, (c, p, l) in enumerate(zip(chroms, pos, svlen)):
        if l <= 1000:
            qc.append(c); qa.append(p); qb.append(p + abs(l)); where.append((i, 0))
        else:
            left = p - 150
            right = p + l + 150
            qc += [c, c]; qa += [left, right]; qb += [left + 100, right + 100]; where += [(i, 1), (i, 2)]
    cnt = spanning_reads(eng, recs, tid_of, qc, qa, qb)
    depth = [0] * len(pos)
    acc = {}
    for (i, kind), c in zip(where, cnt):
        if kind == 0:
            depth[i] = int(c)
        else:
            acc.setdefault(i, []).append(int(c))
    for i, (l_, r_) in acc.items():
        depth[i] = (l_ + r_) / 2
    return depth


def correct_gt_eval(df, t_large_11, t_small_11, t_large_01, t_small_01):
    """DG:291-319 / IG:256-283."""
    new_gt = df['call_gt'].values.copy()
    for large, gt, t in ((True, '1/1', t_large_11), (False, '1/1', t_small_11), (True, '0/1', t_large_01), (False, '0/1', t_small_01)):
        if not np.isnan(t):
            cond = ((df['svlen'] > 1000) if large else (df['svlen'] <= 1000)) & (df['call_gt'] == gt)
            new_gt[cond & (df['n_ratio'] > t)] = '1/1'
            new_gt[cond & (df['n_ratio'] <= t)] = '0/1'
    return new_gt


def write_new_gt_vcf(vcffile, outfile, df, vtype):
    """DG:322-337: the SVTYPE=<vtype> lines with the sample column replaced by the new genotype (no header)."""
    dc = dict(zip(df['svid'].values, df['new_gt'].values))
    with open(vcffile) as fin, open(outfile, 'w') as fout:
        for line in fin:
            if line[0] != '#' and f'SVTYPE={vtype}' in line:
                data = line.split()
                data[-1] = dc[data[2]]
                fout.write('\t'.join(data) + '\n')


def _reads_view(eng, bamfile):
    from .bam import BamFile
    with BamFile(bamfile) as bf:
        view = bf.fetch_device(eng, None)
        tid_of = {n: i for i, n in enumerate(bf.references)}
    return view, tid_of


def run_del(input_path, output_path, bamfile, sigfile, dtype, eng, reads=None):
    """The DEL script body (DG:347-385 without the evaluation branch): <output_path>, <output_path>.newgt, <input>.newgt.DEL."""
    import pandas as pd
    vars_comp = load_vcf_del(input_path)
    sig_list = load_sig_del(sigfile)
    chroms = [v[2].split()[0] for v in vars_comp]
    pos = [int(v[2].split()[1]) for v in vars_comp]
    svlen = [abs(v[1]) for v in vars_comp]
    support, _ = signature_support(eng, chroms, pos, svlen, sig_list, 2.3, 0.6)
    view, tid_of = reads if reads is not None else _reads_view(eng, bamfile)
    depth = np.array(depth_del(eng, view, tid_of, chroms, pos, svlen))
    support = np.array(support)
    ratio = [1 if depth[i] == 0 else support[i] / depth[i] for i in range(len(support))]
    df = pd.DataFrame({'svlen': svlen, 'svid': [v[2].split()[2] for v in vars_comp], 'call_gt': [v[0] for v in vars_comp], 'n_support': support,
                       'n_cov': depth, 'n_ratio': ratio})
    df.to_csv(output_path, sep='\t', index=False)
    df = pd.read_csv(output_path, sep='\t')
    df['new_gt'] = correct_gt_eval(df, *PARA[(dtype, 'DEL')])
    df.to_csv(output_path + '.newgt', sep='\t', index=False)
    write_new_gt_vcf(input_path, input_path + '.newgt.DEL', df, 'DEL')
    return df


def run_ins(input_path, output_path, bamfile, sig_file, dtype, eng, reads=None, flanking=100):
    """The INS script body (IG:303-349 without the evaluation branch)."""
    import pandas as pd
    sv_list = load_vcf_ins(input_path)
    sig_dc = load_sig_file_ins(sig_file)
    sig_list = [k + (n,) for k, n in sig_dc.items()]
    support, match = signature_support(eng, [v[0] for v in sv_list], [v[1] for v in sv_list], [v[2] for v in sv_list], sig_list, 2.3, 0.6)
    view, tid_of = reads if reads is not None else _reads_view(eng, bamfile)
    cov = spanning_reads(eng, view, tid_of, ['chr' + str(v[0]) for v in sv_list], [v[1] - flanking for v in sv_list], [v[1] + flanking for v in sv_list])
    df = pd.DataFrame(sv_list, columns=['chrom', 'pos', 'svlen', 'call_gt', 'svid'])
    df['match_id'] = match
    df['n_support'] = support
    df['n_cov'] = [int(c) for c in cov]
    df['n_ratio'] = df['n_support'] / df['n_cov']
    df.to_csv(output_path, index=False, sep='\t')
    df = pd.read_csv(output_path, sep='\t')
    df['new_gt'] = correct_gt_eval(df, *PARA[(dtype, 'INS')])
    df.to_csv(output_path + '.newgt', sep='\t', index=False)
    write_new_gt_vcf(input_path, input_path + '.newgt.INS', df, 'INS')
    return df


def vcf_sort(lines):
    """`vcf-sort` (vcftools): header first, then `sort -k1,1d -k2,2n` of the body (LC_ALL=C)."""
    header = [l for l in lines if l[0] == '#']
    body = [l for l in lines if l[0] != '#']
    body.sort(key=lambda l: (l.split('\t')[0].encode(), int(l.split('\t')[1]), l.encode()))
    return header + body


def filter_gt_correction(vcffile, bamfile, reference, pre_cutesig, dtype, chr_num=None, device=0, engine=None):
    """filter_GT_correction.py: read signatures (sig_extract.py unless given) -> calculate_signature_support -> coverage band on
    DEL (filter_vcf_by_sig_cov_insdel -v DEL) -> genotype correction of DEL and INS -> variants_filtered_GT_corrected.vcf."""
    from . import sig_cov_filter, sig_extract, signature_support
    from .engine import Engine
    eng = engine or Engine(device)
    try:
        workdir = os.path.dirname(vcffile)
        if pre_cutesig is None:
            sigdir = os.path.join(workdir, "cute_sig")
            os.makedirs(sigdir, exist_ok=True)
            bed = None
            if chr_num is not None:                                      # FGT:64-74
                with open(reference + ".fai") as f:
                    chr_len = int(f.readlines()[chr_num - 1].split()[1])
                bed = sigdir + "/sample.bed"
                with open(bed, 'w') as f:
                    f.write("chr" + str(chr_num) + "\t1\t" + str(chr_len) + '\n')
            sig_extract.run(bamfile, reference, sigdir, include_bed=bed, engine=eng)
        else:
            sigdir = pre_cutesig
        signature_support.run(vcffile, sigdir, chr_num=chr_num, engine=eng)
        filtered = sig_cov_filter.run(vcffile, dtype.lower(), "volcano", "DEL")
        gtdir = os.path.join(workdir, "GT_Correction")
        os.makedirs(gtdir, exist_ok=True)
        reads = _reads_view(eng, bamfile)
        run_del(filtered, gtdir + "/bnd_del_real.tsv", bamfile, sigdir + "/DEL.sigs", dtype, eng, reads)
        run_ins(filtered, gtdir + "/bnd_ins_real.tsv", bamfile, sigdir + "/INS.sigs", dtype, eng, reads)
    finally:
        if engine is None:
            eng.close()
    header = [l for l in open(filtered) if l[0] == '#']
    final_vcf = os.path.join(workdir, "variants_filtered_GT_corrected.vcf")
    with open(final_vcf, 'w') as f:
        f.writelines(vcf_sort(header + open(filtered + ".newgt.DEL").readlines() + open(filtered + ".newgt.INS").readlines()))
    return final_vcf
