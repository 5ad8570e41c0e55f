"""VCF text of the Large_INDEL raw call set — the host step right of the hot path.

Mirrors add_seq_to_sig / write_vcf / load_contigs / reverse_compelement of the reference
(extract_contig_signature_Hifi.py:605-714): REF/ALT are sliced from the reference chromosome and contig strings,
SVLEN is len(ALT)-len(REF) of the sliced strings, POS is the 0-based signature offset written verbatim, ids are
volcano.<chr>.<type>.<n> with per-type counters in output order, signatures whose contig name is missing from the
contig FASTA are dropped (H:648). The integer columns come straight from the C-ABI call table.
"""
from . import sigtable

HG19_LENGTHS = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022, 141213431, 135534747,
                135006516, 133851895, 115169878, 107349540, 102531392, 90354753, 81195210, 78077248, 59128983, 63025520,
                48129895, 51304566]

_INFO = [
    ("SVTYPE", "1", "String", "Type of SV:DEL=Deletion, CON=Contraction, INS=Insertion, DUP=Duplication, INV=Inversion"),
    ("SVLEN", ".", "Integer", "Difference in length between REF and ALT alleles"),
    ("TIG_REGION", ".", "String", "Contig region where variant was found (one per alt with h1 before h2 for homozygous calls)"),
    ("QUERY_STRAND", ".", "String", "Strand of variant in the contig relative to the reference (order follows TIG_REGION)"),
    ("SIG_SOURCE", ".", "String", "Source of the variant call signature (order follows TIG_REGION)"),
    ("TIG_MAPQ", ".", "String", "Mapping quality of the contigs (order follows TIG_REGION)"),
    ("CollapseId", "1", "Integer", "collapse match ID"),
]


def default_header(sample="HG002", contigs=None):
    """The reference's default header (Large_INDEL/header:1-35: VCFv4.2, hg19 chr1-22 lengths, the seven INFO keys,
    GT) regenerated from its parts; `contigs` = [(name, length)] overrides the hg19 table
    (volcanosv-vc-large-indel.py:104-131 builds the same thing from a .fai)."""
    if contigs is None:
        contigs = [("chr%d" % (i + 1), n) for i, n in enumerate(HG19_LENGTHS)]
    lines = ["##fileformat=VCFv4.2", '##FILTER=<ID=PASS,Description="All filters passed">', "##fileDate=20180605", "##reference=GRCh37"]
    lines += ["##contig=<ID=%s,length=%s>" % (n, l) for n, l in contigs]
    lines += ['##INFO=<ID=%s,Number=%s,Type=%s,Description="%s">' % t for t in _INFO]
    lines += ['##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
              "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample]
    return [l + "\n" for l in lines]


def load_contigs(fasta_path):
    """FASTA -> {name: sequence}; name = header line without '>' and newline (H:605-624)."""
    dc, cur = {}, None
    with open(fasta_path) as f:
        for line in f:
            if ">" in line:
                cur = line[1:-1]
            else:
                dc.setdefault(cur, []).append(line[:-1])
    return {k: "".join(v) for k, v in dc.items()}


_COMP = {"N": "N", "A": "T", "T": "A", "G": "C", "C": "G"}


def reverse_complement(seq):
    return "".join(_COMP[c] for c in seq.upper()[::-1])   # KeyError on other letters, like H:626-632


def vcf_lines(soa, calls, merged, ref_seq, dc_contig):
    """Record lines for one chromosome's call table, in table order (already sorted by pos)."""
    out = []
    ins_cnt = del_cnt = 0
    for c in calls:
        sig = sigtable.call_fields(soa, c, merged)
        if sig[4] not in dc_contig:          # H:648
            continue
        if sig[1] == "INS":
            seq_contig = dc_contig[sig[4]]
            seq = reverse_complement(seq_contig[-sig[6]:-sig[5]]) if sig[7] == "-" else seq_contig[sig[5]:sig[6]]   # H:666-669
        else:
            seq = ref_seq[sig[2]:sig[2] + sig[3]]                                                                    # H:672
        pos = sig[2] - 1
        if sig[1] == "DEL":
            alt = ref_seq[pos]
            ref = alt + seq
            del_cnt += 1
            n = del_cnt
        else:
            ref = ref_seq[pos]
            alt = ref + seq
            ins_cnt += 1
            n = ins_cnt
        svlen = len(alt) - len(ref)
        info = "SVLEN=%d;SVTYPE=%s;TIG_REGION=%s;QUERY_STRAND=%s;SIG_SOURCE=%s;TIG_MAPQ=%s" % (svlen, sig[1], sig[11], sig[12], sig[13], sig[14])
        out.append(sig[0] + "\t" + str(sig[2]) + "\tvolcano.%s.%s.%d\t%s\t%s\t%d\tPASS\t%s\tGT\t%s\n" %
                   (sig[0], sig[1], n, ref.upper(), alt.upper(), 20, info, sig[10]))
    return out


def write_vcf(path, header_lines, lines):
    with open(path, "w") as f:
        f.writelines(header_lines)
        f.writelines(lines)


def integer_columns(lines):
    """(CHROM, POS, SVTYPE, SVLEN, END, GT) of VCF record lines; END is None where the record has no END key
    (Large_INDEL records never have one)."""
    out = []
    for l in lines:
        if l.startswith("#"):
            continue
        f = l.rstrip("\n").split("\t")
        info = dict(kv.split("=", 1) for kv in f[7].split(";") if "=" in kv)
        out.append((f[0], int(f[1]), info.get("SVTYPE"), int(info["SVLEN"]) if "SVLEN" in info else None,
                    int(info["END"]) if "END" in info else None, f[9] if len(f) > 9 else None))
    return out
