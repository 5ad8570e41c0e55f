set -e
timeout -k 10 600 python -m pytest tests/test_inflate.py tests/test_sig_extract.py tests/test_vcf_bam.py -x -q -m gpu 2>&1 | tail -2
VSV_BAM_TIMING=1 python tools/ingest_bench.py 3000000 0 --device-only 2>&1 | grep "upload+inflate\|GPU inflate + GPU parse\|member table" | tail -3
