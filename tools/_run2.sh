export VSV_DEBUG=1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overflow_batches or capacity" > gpurun_out/r4u_tests.log 2>&1 || { tail -60 gpurun_out/r4u_tests.log; exit 1; }
tail -3 gpurun_out/r4u_tests.log
