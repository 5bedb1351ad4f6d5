#!/bin/bash
# GPU test suite, then the default bench line; the bench runs only if pytest ended normally (passed or plain failures):
# after a fault, abort or time limit nothing else is started on the GPU.
mkdir -p gpurun_out
timeout -k 10 ${1:-900} python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/tests.log 2>&1
rc=$?
tail -5 gpurun_out/tests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest ended with $rc: stopping"; exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench.err
brc=$?
echo "bench rc=$brc"; tail -3 gpurun_out/bench.err; head -c 1500 gpurun_out/bench_line.json; echo
[ $rc -eq 0 ] && [ $brc -eq 0 ]
