set -e
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["roofline"]["avg_launch_ms"])'
B="python bench.py --extras none --reps 3 --cpu-sample 0 --cpu-procs 0"
for k in 1 2; do
$B --streams 3 2>/dev/null | python -c "$J" compact_streams3
VSV_BK_COMPACT=0 $B --streams 3 2>/dev/null | python -c "$J" linear_streams3
done
$B --streams 1 2>/dev/null | python -c "$J" compact_streams1
VSV_BK_COMPACT=0 $B --streams 1 2>/dev/null | python -c "$J" linear_streams1
$B --config 6 --streams 3 2>/dev/null | python -c "$J" row2c_streams3
$B --config 4 --streams 3 2>/dev/null | python -c "$J" c4_compact
VSV_BK_COMPACT=0 $B --config 4 --streams 3 2>/dev/null | python -c "$J" c4_linear
