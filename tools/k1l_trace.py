#!/usr/bin/env python3
"""Phase breakdown of the in-order long scan from a VSV_K1L_TRACE dump (6 u64 per part: start, staged, streamed, resolved, end on the
100 MHz wall clock, rows)."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 6)
t = raw.astype(np.int64)
hw = (raw[:, 5] >> np.uint64(16)) & np.uint64(0xFFFFFFFF)
xcc = ((raw[:, 5] >> np.uint64(48)) & np.uint64(0xF)).astype(np.int64)
t[:, 5] = (raw[:, 5] & np.uint64(0xFFFF)).astype(np.int64)
cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(np.int64); se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64); sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64)
n = len(t)
us = lambda a: a / 100.0
ph = {"stage (bounds, record starts)": t[:, 1] - t[:, 0], "stream": t[:, 2] - t[:, 1], "look-back (deferred)": t[:, 3], "flush": t[:, 4], "start -> aggregate": t[:, 2] - t[:, 0]}
print("%d parts, kernel span %.1f us, rows/part mean %.1f max %d" % (n, us(t[:, 2].max() - t[:, 0].min()), t[:, 5].mean(), t[:, 5].max()))
for k, v in ph.items():
    v = us(v)
    print("%-32s mean %7.2f  p50 %7.2f  p90 %7.2f  p99 %7.2f  max %8.2f us" % (k, v.mean(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max()))
# how far in front of its predecessors does a part end its stream? (positive = it has to wait for them)
end = t[:, 2]
run_max = np.maximum.accumulate(end)
lag = us(np.concatenate(([0], run_max[:-1])) - end)
print("slowest lower part's stream end minus own: mean %.2f p50 %.2f p90 %.2f p99 %.2f us" % (lag.mean(), np.percentile(lag, 50), np.percentile(lag, 90), np.percentile(lag, 99)))
st = t[:, 0]
print("start order inversions (part starts before its predecessor): %.1f %%; start skew p99 %.2f us" % (100.0 * np.mean(st[1:] < st[:-1]), np.percentile(us(np.maximum.accumulate(st)[:-1] - st[1:]), 99)))
conc = (t[:, 2] - t[:, 0] + t[:, 3] + t[:, 4]).sum() / float(t[:, 2].max() - t[:, 0].min())
print("average parts in flight: %.0f" % conc)

strm = us(t[:, 2] - t[:, 1])
print("stream time by XCC:", " ".join("%d:%.1f(n=%d)" % (x, strm[xcc == x].mean(), (xcc == x).sum()) for x in np.unique(xcc)))
print("stream time by SE :", " ".join("%d:%.1f" % (x, strm[se == x].mean()) for x in np.unique(se)))
print("stream time by CU :", " ".join("%d:%.1f" % (x, strm[cu == x].mean()) for x in np.unique(cu)))
print("corr(stream, rows) = %.3f; corr(stream, start time) = %.3f" % (np.corrcoef(strm, t[:, 5])[0, 1], np.corrcoef(strm, t[:, 0])[0, 1]))
q = np.argsort(t[:, 0]); k = len(q) // 10
print("stream time by start decile:", " ".join("%.1f" % strm[q[i * k:(i + 1) * k]].mean() for i in range(10)))
print("parts by XCC in the first 20000 parts:", " ".join("%d:%d" % (x, (xcc[:20000] == x).sum()) for x in np.unique(xcc)))
