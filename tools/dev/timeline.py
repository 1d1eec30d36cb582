"""Timeline analysis of a rocprofv3 --kernel-trace database (rocpd sqlite): for the last `frames` render calls of a pipelined
run — how busy is the device (union of kernel intervals over wall time), how many kernels overlap on average, how long are the
gaps between consecutive kernels of one stream (launch latency + waiting for dependencies), which queues do the streams use.
usage: python tools/dev/timeline.py <results.db> [window_fraction_from_end=0.5]"""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
c = sqlite3.connect(db).cursor()
rows = c.execute("select name, stream_id, queue_id, start, end from kernels order by start").fetchall()
t0, t1 = rows[0][3], max(r[4] for r in rows)
lo = t1 - (t1 - t0) * frac
rows = [r for r in rows if r[3] >= lo]
wall = (max(r[4] for r in rows) - rows[0][3]) / 1e6
# union and integral of concurrency
ev = sorted([(r[3], 1) for r in rows] + [(r[4], -1) for r in rows])
busy = area = 0.0
depth, last = 0, ev[0][0]
hist = defaultdict(float)
for t, d in ev:
    if depth > 0:
        busy += t - last
    area += depth * (t - last)
    hist[depth] += t - last
    depth += d
    last = t
print("window %.2f ms, %d kernels; device busy %.1f %% of the time; mean kernels in flight while busy %.2f" % (wall, len(rows), 100 * busy / 1e6 / wall, area / max(busy, 1)))
print("time by number of kernels in flight: " + "  ".join("%d: %.0f %%" % (k, 100 * v / 1e6 / wall) for k, v in sorted(hist.items())))
by = defaultdict(list)
for r in rows:
    by[r[0].split("(")[0][-40:]].append((r[4] - r[3]) / 1e3)
print("kernel: calls, mean us")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("  %-42s %5d  %8.1f" % (k, len(v), sum(v) / len(v)))
gaps = defaultdict(list)
prev = {}
for r in rows:
    s = r[1]
    if s in prev:
        gaps[s].append((r[3] - prev[s]) / 1e3)
    prev[s] = r[4]
print("stream: queue, kernels, mean gap between consecutive kernels (us), share of wall spent in gaps")
qs = {}
for r in rows:
    qs.setdefault(r[1], set()).add(r[2])
for s, g in sorted(gaps.items()):
    pos = [x for x in g if x > 0]
    print("  stream %s queues %s: %d kernels, mean gap %.1f us, gaps %.0f %% of wall" % (s, sorted(qs[s]), len(g) + 1, sum(pos) / max(1, len(pos)), 100 * sum(pos) / 1e3 / wall))
