export VSV_DEBUG=1 PYTHONUNBUFFERED=1 VSV_SPLIT_STREAM=main
{
for k in sl_bucket_lds "sl_scatter<12" "sl_hist<12" sl_pair_lds sl_cluster; do
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
    echo "== $k : $c"
    bash tools/pmc_kernel.sh r4p "$k" "$c" --config 6 --streams 1 2>&1 | tail -4
  done
done
} > gpurun_out/r04_pmc_row2c_post_scan_kernels.txt 2>&1
tail -70 gpurun_out/r04_pmc_row2c_post_scan_kernels.txt
