#!/bin/bash
# rocprofv3 kernel statistics of any python script of this repository (run through gpurun from the repo root).
# usage: profile_script.sh <tag> <script.py> [args...]: prints the twenty kernels with the largest total time
set -e
REPO=$PWD
tag=$1; shift
OUT=$REPO/gpurun_out/prof_$tag
rm -rf $OUT && mkdir -p $OUT
script=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o c -- python3 $script "$@" > $OUT/line.txt 2> $OUT/err.txt
cd $REPO
cp $(find $OUT/raw -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/raw
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:20]:
    print("%-78s calls %6d avg %8.1f us %5.1f%%" % (r["Name"].replace("(anonymous namespace)::", "")[:78], int(r["Calls"]),
                                                   float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
tail -3 $OUT/line.txt
