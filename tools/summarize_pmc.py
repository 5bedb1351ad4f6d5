"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE counter_collection CSVs).
FETCH_SIZE (KB) is doubled: on gfx950 it tallies the 128-byte requests of wide streaming reads at 64 bytes (EXACT_FETCH lists the kernels whose requests are 64 bytes wide and need no correction).
Besides the per-kernel averages the summary carries `sequence`: for the GEMM and depthwise kernel families the bytes of
every launch of ONE pass in launch order (averaged over the passes of the run), so that bench.py can attribute traffic
to the launch shapes of a residual layer although several shapes share one kernel instance.
usage: summarize_pmc.py fetch.csv write.csv [out.json [batch_per_gpu [passes]]]"""
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


FAMILIES = {"gemm_nn": "gemm_nn", "gemm_nt": "gemm_nt", "dw_fwd": "dw_fwd", "dw_bwd": "dw_bwd"}
# kernels whose read requests are 64 bytes wide (four lanes of a row): FETCH_SIZE tallies them as they are -- for the
# heads' whole-output weight gradient TCC_EA0_RDREQ x 64 bytes equals the operands' 524 MB at batch 32 -- so no doubling
EXACT_FETCH = ("gemm_nt_whole_kernel<false, 16",)


def fetch_scale(name):
    return 1.0 if any(k in name for k in EXACT_FETCH) else 2.0


def load_sequence(path, cname, scale):
    """family -> [bytes of each dispatch, in dispatch order]"""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == cname:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"],
                             float(r["Counter_Value"]) * 1024.0 * (fetch_scale(r["Kernel_Name"]) if scale == 2.0 else scale)))
    rows.sort()
    seq = collections.defaultdict(list)
    for _, name, v in rows:
        if "gemm_nt_whole_reduce" in name and seq["gemm_nt"]:
            seq["gemm_nt"][-1] += v          # second launch of the same ctn_gemm_nt call: one entry per call
            continue
        for fam, key in FAMILIES.items():
            if key in name:
                seq[fam].append(v)
                break
    return seq


def per_pass(seq, passes):
    """Average over the passes, position by position; families whose launch count is not a multiple of `passes` are left out."""
    out = {}
    for fam, vals in seq.items():
        if not vals or len(vals) % passes:
            continue
        n = len(vals) // passes
        out[fam] = [sum(vals[p * n + i] for p in range(passes)) / passes for i in range(n)]
    return out


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fe, key=lambda k: -fe[k][1]):
    n, v = fe[k]
    wn, wv = wr.get(k, [0, 0.0])
    rd = fetch_scale(k) * v / n * 1024.0
    wb = wv / max(1, wn) * 1024.0
    out[k] = dict(launches=n, read_bytes_per_launch=rd, write_bytes_per_launch=wb, hbm_bytes_per_launch=rd + wb)
for k, d in list(out.items())[:16]:
    print("%-46s launches %4d  read %8.1f MB  write %8.1f MB per launch" %
          (k[:46], d["launches"], d["read_bytes_per_launch"] / 1e6, d["write_bytes_per_launch"] / 1e6))
if len(sys.argv) > 3:
    passes = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    rseq = per_pass(load_sequence(sys.argv[1], "FETCH_SIZE", 2.0), passes)
    wseq = per_pass(load_sequence(sys.argv[2], "WRITE_SIZE", 1.0), passes)
    sequence = {fam: [round(a + b) for a, b in zip(rseq[fam], wseq[fam])]
                for fam in rseq if fam in wseq and len(rseq[fam]) == len(wseq[fam])}
    doc = {
        "batch_per_gpu": int(sys.argv[4]) if len(sys.argv) > 4 else 32,
        "command": "bench.py --steps 1 --warmup 1 --streams 1 --no-graph --no-cpu-baseline --no-split-probe --no-recipe-batch "
                   "--no-roofline (default arithmetic: three bf16 pieces)",
        "collection": "two rocprofv3 --pmc passes of the same command (FETCH_SIZE, WRITE_SIZE), tools/profile_bench.sh; "
                      "FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of 16-byte-per-lane streaming reads at 64 "
                      "bytes), except for gemm_nt_whole_kernel<false, 16, ...>, whose requests are 64 bytes wide and "
                      "tallied as they are; kernels that read 4 bytes per lane (maskdecode_*, encoder_bwd) are "
                      "over-counted by the correction",
        "passes": passes,
        "kernels": out,
        "sequence": sequence,
    }
    json.dump(doc, open(sys.argv[3], "w"), indent=1)
