#!/bin/bash
# Kernel trace of the default bench mode (graph replay, two sub-batch streams) and its busy / idle analysis.
set -e
REPO=$PWD
OUT=$REPO/gpurun_out/trace
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o t -- \
    python3 $REPO/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-split-probe --no-roofline > $OUT/line.json 2> $OUT/err.txt
cd $REPO
F=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python3 tools/analyze_trace.py $F 0.6 > $OUT/analysis.txt
cat $OUT/analysis.txt
rm -rf $OUT/raw
