"""Busy / idle analysis of a rocprofv3 kernel trace (csv): union of kernel intervals against the wall span, time with
two or more kernels resident, and the gaps.  usage: analyze_trace.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_lo, t_hi = ev[0][0], ev[-1][1]
cut = t_lo + (t_hi - t_lo) * skip            # steady state only: drop set-up, capture and warm-up
ev = [e for e in ev if e[0] >= cut]
span = ev[-1][1] - ev[0][0]
points = []
for s, e, _ in ev:
    points.append((s, 1)); points.append((e, -1))
points.sort()
busy = overlap = 0
depth = 0
last = points[0][0]
gaps = []
for t, d in points:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        overlap += t - last
    if depth == 0 and t > last:
        gaps.append(t - last)
    depth += d
    last = t
tot = sum(e - s for s, e, _ in ev)
print("kernels %d  span %.2f ms  busy %.2f ms (%.1f%%)  >=2 resident %.2f ms (%.1f%%)  sum of durations %.2f ms" % (
    len(ev), span / 1e6, busy / 1e6, 100.0 * busy / span, overlap / 1e6, 100.0 * overlap / span, tot / 1e6))
gaps.sort()
if gaps:
    big = [g for g in gaps if g > 20000]
    print("idle gaps %d  total %.2f ms  median %.1f us  p90 %.1f us  max %.1f us  | gaps > 20 us: %d totalling %.2f ms" % (
        len(gaps), sum(gaps) / 1e6, gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3,
        len(big), sum(big) / 1e6))
