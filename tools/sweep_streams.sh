cd /root/repo
for cfg in "32 1" "32 2" "32 4" "64 2" "64 4" "64 8" "16 2"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --batch $1 --streams $2 --no-cpu-baseline --no-split-probe --no-recipe-batch --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $1 streams $2:', d['value'],'utt/s', d['ms_per_step'],'ms')" || exit 1
done
