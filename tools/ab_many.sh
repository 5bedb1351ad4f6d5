#!/bin/bash
# Headline step with the in-tree library and every diagnostic build under tools/variants, interleaved (usage: [rounds]).
cd "$(dirname "$0")/.."
for r in $(seq 1 ${1:-2}); do
  for lib in speech_amd/libctn_hip.so tools/variants/libctn_*.so; do
    CTN_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-split-probe --no-recipe-batch --no-roofline --steps 40 2>/dev/null \
      | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'],'utt/s', d['ms_per_step'],'ms')" || exit 1
  done
done
