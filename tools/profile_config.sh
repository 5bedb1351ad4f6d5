#!/bin/bash
# rocprofv3 kernel statistics of tools/bench_config.py for a named configuration (run through gpurun from the repo root).
# usage: profile_config.sh <paper|causal3> [batch]
set -e
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$1
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o c -- \
    python3 $REPO/tools/bench_config.py $1 ${2:-16} > $OUT/line.txt 2> $OUT/err.txt
cd $REPO
cp $(find $OUT/raw -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/raw
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-72s calls %5d avg %8.1f us %5.1f%%" % (r["Name"].replace("(anonymous namespace)::", "")[:72], int(r["Calls"]),
                                                   float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
cat $OUT/line.txt
