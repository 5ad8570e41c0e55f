set -e
for S in 1 2 4; do echo "slices $S"; VSV_INFLATE_SLICES=$S VSV_BAM_TIMING=1 python tools/ingest_bench.py 3000000 0 --device-only 2>&1 | grep "upload+inflate\|GPU inflate + GPU parse" | tail -2; done
python tools/ingest_bench.py 600000 0 --device-only 2>&1 | tail -1
python tools/inflate_bench.py 9124 2>&1 | tail -1
