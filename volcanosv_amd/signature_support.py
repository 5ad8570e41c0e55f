"""Host mirror of Large_INDEL/calculate_signature_support.py — signature coverage of every INS/DEL call, the first step of
the GT-correction filter chain (filter_GT_correction.py:134-137). Same function names and file formats as the reference;
the coverage sums run on the GPU (vsv_support_cov_ins / vsv_support_cov_del, one wave per call) instead of the Python
scans (CS:81-125, 138-280)."""
import os
from collections import defaultdict

import numpy as np

from .engine import Engine


def load_vcf(vcffile, svtype, min_size=30):
    """CS:32-56: {chrom: [(start, end, svlen, svid, gt, svtype)]} for lines holding 'SVTYPE=<svtype>' and |SVLEN| >= min_size."""
    dc = defaultdict(list)
    with open(vcffile, 'r') as f:
        for line in f:
            if line[0] != '#' and 'SVTYPE=%s' % svtype in line:
                data = line.split()
                svlen = int(data[7].split('SVLEN=')[1].split(';')[0])
                if abs(svlen) >= min_size:
                    start = int(data[1])
                    end = start + 1 if svtype == 'INS' else start - svlen
                    dc[data[0]].append((start, end, svlen, data[2], data[-1].split(':')[0], svtype))
    return dc


def load_sig(sig_path, svtype):
    """CS:58-79: {chrom: [(start, end, svlen)]} from INS.sigs / DEL.sigs (DEL lengths negated, end = start + length)."""
    info_dc = defaultdict(list)
    with open(sig_path, 'r') as f:
        for line in f:
            data = line.split()
            start, svlen = int(data[2]), int(data[3])
            if svtype == 'INS':
                end = start + 1
            else:
                svlen = -svlen
                end = start - svlen
            info_dc[data[1]].append((start, end, svlen))
    return info_dc


def _sorted_sig_arrays(sig_list):
    a = np.array(sig_list, dtype=np.int64).reshape(-1, 3)
    order = np.argsort(a[:, 0], kind="stable")
    return a[order].astype(np.int32)


def calc_ins_call_cov(call_list, sig_list, flanking=1000, engine=None):
    """CS:81-125: {call position: summed length (np.float64, as np.bincount's weights give) of the signatures within +-flanking}."""
    eng = engine or Engine(0)
    try:
        pos = np.array(sorted(set(c[0] for c in call_list)), dtype=np.int32)
        s = _sorted_sig_arrays(sig_list)
        cov = eng.support_cov_ins(pos, s[:, 0], s[:, 2], flanking)
    finally:
        if engine is None:
            eng.close()
    return dict(zip((int(p) for p in pos), (np.float64(c) for c in cov)))


def calc_del_call_cov(call_list, sig_list, flanking=1000, engine=None):
    """CS:138-280: {(start, end): summed (negative) svlen of the distinct signatures meeting [start-flanking, end+flanking]};
    calls without a supporting signature have no entry."""
    eng = engine or Engine(0)
    try:
        cs = np.array([c[0] for c in call_list], dtype=np.int32)
        ce = np.array([c[1] for c in call_list], dtype=np.int32)
        s = _sorted_sig_arrays(sig_list)
        cov = eng.support_cov_del(cs, ce, s[:, 0], s[:, 1], s[:, 2], flanking)
    finally:
        if engine is None:
            eng.close()
    return {(int(a), int(b)): int(c) for a, b, c in zip(cs, ce, cov) if c != 0}


def run(vcffile, cutesv_dir, flanking=1000, min_size=30, chr_num=None, device=0, engine=None):
    """The script body (CS:285-378): writes <vcf dir>/<vcf base>_cutesv_sig_support_mins<min_size>_fl<flanking>.csv."""
    import pandas as pd
    dc_sig_ins = load_sig(cutesv_dir + '/INS.sigs', 'INS')
    dc_sig_del = load_sig(cutesv_dir + '/DEL.sigs', 'DEL')
    dc_call_ins = load_vcf(vcffile, 'INS', min_size)
    dc_call_del = load_vcf(vcffile, 'DEL', min_size)
    chroms = ['chr' + str(i) for i in range(1, 23)] if chr_num is None else ['chr' + str(chr_num)]
    eng = engine or Engine(device)
    final_info = []
    try:
        for chrom in chroms:
            if chrom in dc_call_ins:
                cov = calc_ins_call_cov(dc_call_ins[chrom], dc_sig_ins[chrom], flanking, engine=eng)
                for start, end, svlen, svid, gt, svtype in dc_call_ins[chrom]:
                    final_info.append([start, end, svlen, svid, gt, svtype, cov[start] if start in cov else 0])
        for chrom in chroms:
            if chrom in dc_call_del:
                cov = calc_del_call_cov(dc_call_del[chrom], dc_sig_del[chrom], flanking, engine=eng)
                for start, end, svlen, svid, gt, svtype in dc_call_del[chrom]:
                    final_info.append([start, end, svlen, svid, gt, svtype, cov[(start, end)] if (start, end) in cov else 0])
    finally:
        if engine is None:
            eng.close()
    df = pd.DataFrame(final_info, columns=['start', 'end', 'svlen', 'svid', 'gt', 'svtype', 'cov'])
    df['rel_cov'] = df['cov'] / df['svlen']
    csv = os.path.dirname(vcffile) + '/' + os.path.basename(vcffile).split('.')[0] + '_cutesv_sig_support_mins%d_fl%d.csv' % (min_size, flanking)
    df.to_csv(csv, index=False)
    return csv
