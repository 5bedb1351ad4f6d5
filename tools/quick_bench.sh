#!/bin/bash
# Headline step and per-kernel rows only, with the in-kernel split and with prepared weights (usage: [extra bench args]).
cd "$(dirname "$0")/.."
for pw in 0 1; do
  CTN_PREPARED_WEIGHTS=$pw timeout -k 10 200 python bench.py --no-cpu-baseline --no-split-probe --no-recipe-batch "$@" \
    | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('prepared=$pw', d['value'],'utt/s', d['ms_per_step'],'ms', 'nn',d['roofline']['achieved'],d['roofline']['avg_launch_us'],'nt',d['roofline_wgrad']['achieved'], d['kernel_ms_per_step'])
" || exit 1
done
