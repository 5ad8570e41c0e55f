"""Host glue of the CLI contract (volcanosv_amd/pipeline.py): header generation, .fai, reference split, phasing tags."""
import os

from volcanosv_amd import pipeline


def test_fai_header_split(tmp_path):
    fa = tmp_path / "ref.fa"
    fa.write_text(">chr1 desc\nACGTACGTAC\nACG\n>chr2\nTTTT\n")
    fai = pipeline.write_fai(str(fa))
    assert open(fai).read() == "chr1\t13\t11\t10\t11\nchr2\t4\t32\t4\t5\n"
    hdr = tmp_path / "VCF_header"
    head = pipeline.generate_vcf_header(str(fa), str(hdr), None, "HG002")
    assert head == "##fileformat=VCFv4.2\n##contig=<ID=chr1,length=13>\n##contig=<ID=chr2,length=4>\n"
    txt = open(hdr).read()
    assert txt.endswith("FORMAT\tHG002\n") and "ID=PS," in txt and txt.count("##INFO") == 8
    assert pipeline.generate_vcf_header(str(fa), str(hdr), 2, "S") == "##fileformat=VCFv4.2\n##contig=<ID=chr2,length=4>\n"
    pipeline.split_reference(str(fa), str(tmp_path / "by_chr"), None)
    assert open(tmp_path / "by_chr" / "chr2.fa").read() == ">chr2\nTTTT\n"


def test_phase_tags():
    line = "chr1\t100\tvolcano.chr1.DEL.1\tAC\tA\t20\tPASS\tSVLEN=-1;SVTYPE=DEL;TIG_REGION=PS1200_hp2_c:5-6;QUERY_STRAND=+\tGT\t0/1\n"
    out = pipeline.phase_large_indel([line], ["##h\n"])
    assert out[0] == "##h\n" and out[1].endswith(";PS=1200\tGT\t0|1\n")
    both = line.replace("PS1200_hp2_c:5-6", "PS7_hp1_a:1-2,PS7_hp2_b:3-4").replace("0/1", "1/1")
    assert pipeline.phase_large_indel([both], [])[0].endswith(";PS=7\tGT\t1|1\n")
    hdr = ["##a\n"] * 7 + ["#CHROM\n"]
    b = "chr1\t5\tsvim_asm.BND.1\tN\tN[chr2:9[\t.\tPASS\tSVTYPE=BND;READS=PS42_hp1_x,PS42_hp2_y\tGT\t1/0\n"
    ph = pipeline.phase_complex(hdr + [b])
    assert ph[2] == "##a\n" + pipeline.PS_INFO and ph[-1].endswith(";PS=42\tGT\t1|0\n")


def test_cli_scripts_parse(tmp_path):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for s in ("Raw_variant_call.py", "volcanosv-vc-large-indel.py", "volcanosv-vc-complex-sv.py", "extract_contig_signature_Hifi.py",
              "extract_reads_signature.py", "filter_tra.py", "FP_filter_v1.py", "calculate_signature_support.py",
              "filter_vcf_by_sig_cov_insdel.py", "sig_extract.py", "remove_redundancy.py", "correct_gt_del_real_data.py",
              "correct_gt_ins_real_data.py", "filter_GT_correction.py"):
        r = subprocess.run([sys.executable, os.path.join(root, "volcanosv_amd", "cli", s), "--help"], capture_output=True, text=True)
        assert r.returncode == 0, s + r.stderr[-500:]
