"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE counter_collection CSVs).
FETCH_SIZE (KB) is doubled: on gfx950 it tallies the 128-byte requests of wide streaming reads at 64 bytes.
usage: summarize_pmc.py fetch.csv write.csv [out.json [batch_per_gpu]]"""
import collections, csv, json, re, sys


def load(path, cname):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != cname:
                continue
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fe, key=lambda k: -fe[k][1]):
    n, v = fe[k]
    wn, wv = wr.get(k, [0, 0.0])
    rd = 2.0 * v / n * 1024.0
    wb = wv / max(1, wn) * 1024.0
    out[k] = dict(launches=n, read_bytes_per_launch=rd, write_bytes_per_launch=wb, hbm_bytes_per_launch=rd + wb)
for k, d in list(out.items())[:16]:
    print("%-46s launches %4d  read %8.1f MB  write %8.1f MB per launch" %
          (k[:46], d["launches"], d["read_bytes_per_launch"] / 1e6, d["write_bytes_per_launch"] / 1e6))
if len(sys.argv) > 3:
    doc = {
        "batch_per_gpu": int(sys.argv[4]) if len(sys.argv) > 4 else 32,
        "command": "bench.py --steps 1 --warmup 1 --streams 1 --no-graph --no-cpu-baseline --no-split-probe --no-recipe-batch "
                   "--no-roofline (default arithmetic: three bf16 pieces)",
        "collection": "two rocprofv3 --pmc passes of the same command (FETCH_SIZE, WRITE_SIZE), tools/profile_bench.sh; "
                      "FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of 16-byte-per-lane streaming reads at 64 "
                      "bytes); kernels that read 4 bytes per lane (maskdecode_*, encoder_bwd) are over-counted by that "
                      "correction",
        "kernels": out,
    }
    json.dump(doc, open(sys.argv[3], "w"), indent=1)
