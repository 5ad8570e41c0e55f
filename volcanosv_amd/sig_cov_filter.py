"""Host mirror of Large_INDEL/filter_vcf_by_sig_cov_insdel.py: keeps the calls whose relative signature coverage
(calculate_signature_support's CSV) lies inside an empirical band around the median (FV:38-68). Pure host logic on a
per-call table; the coverage itself comes from the GPU (volcanosv_amd/signature_support.py)."""
import numpy as np

# empirical band factors of the reference's filter_para.csv: (asm, dtype) -> (lb_ins, rb_ins, lb_del, rb_del)
FILTER_PARA = {
    ("other", "hifi"): (0.048, 2.61, 0.097, 2.59),
    ("other", "clr"): (0.0327, 2.476, 0.102, 2.638),
    ("other", "ont"): (0.191, 2.44, 0.123, 2.67),
    ("volcano", "hifi"): (0.097, 2.754, 0.2, 2.605),
    ("volcano", "clr"): (0.075, 2.383, 0.186, 3.018),
    ("volcano", "ont"): (0.206, 2.79, 0.242, 2.77),
}


def passing_ids(csvfile, dtype, asm, vtype='INSDEL'):
    """FV:31-68: ids of the calls that stay. vtype names the SV type the band is applied to (the other type passes whole)."""
    import pandas as pd
    assert vtype in ['INS', 'DEL', 'INSDEL']
    lb_ins_r, rb_ins_r, lb_del_r, rb_del_r = FILTER_PARA[(asm, dtype)]
    df = pd.read_csv(csvfile)
    df['re_cov'] = df['cov'] / df['svlen']
    ids = set()
    for svtype, lb_r, rb_r, skip in (('INS', lb_ins_r, rb_ins_r, 'DEL'), ('DEL', lb_del_r, rb_del_r, 'INS')):
        d = df[df['svtype'] == svtype]
        if d.shape[0]:
            if vtype != skip:
                med = np.quantile(d['re_cov'], 0.5)
                d = d[(d['re_cov'] >= med * lb_r) & (d['re_cov'] <= med * rb_r)]
            ids |= set(d['svid'].values)
    return ids


def run(input_path, dtype, asm, vtype='INSDEL'):
    """The script body (FV:31-36, 107-121): writes <input>_filter_<vtype>.vcf next to the input VCF."""
    csvfile = input_path.replace('.vcf', '_cutesv_sig_support_mins30_fl1000.csv')
    ids = passing_ids(csvfile, dtype, asm, vtype)
    outvcf = input_path.replace('.vcf', '_filter_%s.vcf' % vtype)
    with open(outvcf, 'w') as fw, open(input_path, 'r') as f:
        for line in f:
            if line[0] == '#' or line.split()[2] in ids:
                fw.write(line)
    return outvcf
