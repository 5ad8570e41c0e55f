set -e
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/parity.log 2>&1 || { tail -30 gpurun_out/parity.log; exit 1; }
tail -2 gpurun_out/parity.log
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["roofline"]["avg_launch_ms"])'
B="python bench.py --extras none --reps 3 --cpu-sample 0 --cpu-procs 0"
VSV_SPLIT_STREAM=main $B --streams 1 2>/dev/null | python -c "$J" main1_self
VSV_SPLIT_STREAM=main VSV_BK_SCAN=launch $B --streams 1 2>/dev/null | python -c "$J" main1_launch
$B --streams 1 2>/dev/null | python -c "$J" fork1
$B --streams 2 2>/dev/null | python -c "$J" fork2
$B --streams 3 2>/dev/null | python -c "$J" nofork3
VSV_BK_SCAN=launch $B --streams 3 2>/dev/null | python -c "$J" nofork3_launch
