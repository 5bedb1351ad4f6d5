#!/bin/bash
# usage: pmc_kernel.sh <tag> <counters...> -- <program args>: one counter pass of a program, per-kernel averages printed
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace -d $out -o r --output-format csv -- "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"][:60]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (k, row["Dispatch_Id"])
    if key not in seen:
        seen.add(key); n[k] += 1
for k in acc:
    print(k, "launches", n[k], " ".join("%s=%.4g" % (c, v / n[k]) for c, v in sorted(acc[k].items())))
PY
