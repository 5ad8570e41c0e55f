export VSV_DEBUG=1 PYTHONUNBUFFERED=1
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bucket_sort_overflow" --timeout=300 --timeout-method=thread > gpurun_out/r4i_tests.log 2>&1 || { grep -v "^  File\|^    " gpurun_out/r4i_tests.log | tail -40; exit 1; }
tail -3 gpurun_out/r4i_tests.log
VSV_BK_TIEMAX=0 timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or synthetic_vs_oracle or multi_tid or random_small or dense_runs or pairing_in_rounds or full_size_config2 or config4_shape" --timeout=300 --timeout-method=thread > gpurun_out/r4i_tests2.log 2>&1 || { grep -v "^  File\|^    " gpurun_out/r4i_tests2.log | tail -40; exit 1; }
tail -3 gpurun_out/r4i_tests2.log
