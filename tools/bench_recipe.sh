#!/bin/bash
# The reference recipe's batch of 2 per GPU: the bench line's recipe_batch_2 entry only.
cd "$(dirname "$0")/.."
timeout -k 10 200 python bench.py --no-cpu-baseline --no-split-probe --no-roofline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['recipe_batch_2']['value'],'utt/s at batch 2,', d['recipe_batch_2']['ms_per_step'],'ms')"
