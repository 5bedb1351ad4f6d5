#!/bin/bash
# rocprofv3 passes of bench.py whose summaries go to profiles/ (run on the GPU box through gpurun, from the repo root):
#   1. kernel trace + stats of the profiled command (eager, one stream: the mode in which a kernel's duration is its own)
#   2. kernel trace + stats of the default mode (graph replay, one stream)
#   3. / 4. PMC passes, one counter each: FETCH_SIZE, WRITE_SIZE
# Outputs land under gpurun_out/prof/.
set -e
REPO=$PWD
OUT=$REPO/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager -o eager -- \
    python3 $REPO/bench.py --steps 10 --warmup 2 --streams 1 --no-graph --no-cpu-baseline --no-split-probe --no-recipe-batch > $OUT/eager_line.json 2> $OUT/eager.err
echo "pass 1 done"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -o default -- \
    python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-split-probe --no-recipe-batch > $OUT/default_line.json 2> $OUT/default.err
echo "pass 2 done"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- \
    python3 $REPO/bench.py --steps 1 --warmup 1 --streams 1 --no-graph --no-cpu-baseline --no-split-probe --no-recipe-batch --no-roofline > $OUT/fetch_line.json 2> $OUT/fetch.err
echo "pass 3 done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- \
    python3 $REPO/bench.py --steps 1 --warmup 1 --streams 1 --no-graph --no-cpu-baseline --no-split-probe --no-recipe-batch --no-roofline > $OUT/write_line.json 2> $OUT/write.err
echo "pass 4 done"
cd $REPO
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
python3 tools/summarize_pmc.py $F $W $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
cp $(find $OUT/eager -name "*kernel_stats.csv" | head -1) $OUT/eager_kernel_stats.csv
cp $(find $OUT/default -name "*kernel_stats.csv" | head -1) $OUT/default_kernel_stats.csv
# the raw traces are large: keep the summaries only
rm -rf $OUT/eager $OUT/default $OUT/fetch $OUT/write
ls -la $OUT
