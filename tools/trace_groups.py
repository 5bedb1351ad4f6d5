"""Per (kernel, grid) durations from a rocprofv3 kernel_trace.csv: count, average, min, max in microseconds.
usage: trace_groups.py kernel_trace.csv [substring]"""
import collections, csv, re, sys
sub = sys.argv[2] if len(sys.argv) > 2 else ""
g = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
        if sub not in name:
            continue
        grid = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        g[(name[:60], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in g.values())
for (name, grid), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print("%6.2f%% %9.1f ms  n %5d  avg %8.1f  min %8.1f  max %8.1f  %s  grid %s" %
          (100 * sum(v) / tot, sum(v) / 1e3, len(v), sum(v) / len(v), min(v), max(v), name, grid))
