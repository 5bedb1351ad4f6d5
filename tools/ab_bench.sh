#!/bin/bash
# A/B of the headline step on one box: the in-tree library against a diagnostic build (usage: ab_bench.sh path/to/lib.so [rounds]).
cd "$(dirname "$0")/.."
for r in $(seq 1 ${2:-2}); do
  for lib in "" "$1"; do
    CTN_LIB=$PWD/${lib:-speech_amd/libctn_hip.so} timeout -k 10 200 python bench.py --no-cpu-baseline --no-split-probe --no-recipe-batch --no-roofline --steps 40 2>/dev/null \
      | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-in-tree}', d['value'],'utt/s', d['ms_per_step'],'ms')" || exit 1
  done
done
