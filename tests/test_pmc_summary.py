"""tools/summarize_pmc.py on a synthetic pair of counter files: FETCH_SIZE is doubled except for the kernels listed as
exact, the reduction launch of a whole-output weight gradient is folded into its ctn_gemm_nt call's entry of the
per-launch sequence, and sequences are averaged over the passes."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write(path, counter, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for i, (name, v) in enumerate(rows):
            w.writerow([i + 1, name, counter, v])


def test_summary(tmp_path):
    heads = "void (anonymous namespace)::gemm_nt_whole_kernel<false, 16, 4, 2, 2, 4>(ctn_gemm_nt_args, bool)"
    red = "void (anonymous namespace)::gemm_nt_whole_reduce_kernel<4, 2, 2, 4>(ctn_gemm_nt_args, int, int, int)"
    bneck = "void (anonymous namespace)::gemm_nt_whole_kernel<true, 32, 2, 2, 4, 2>(ctn_gemm_nt_args, bool)"
    nn = "void (anonymous namespace)::gemm_nn_fast_kernel<256, 0, false, 6>(ctn_gemm_nn_args, int, int)"
    one_pass = [(nn, 100.0), (heads, 500.0), (red, 60.0), (bneck, 300.0), (red, 30.0)]
    fetch, wr = tmp_path / "f.csv", tmp_path / "w.csv"
    write(fetch, "FETCH_SIZE", one_pass * 2)                       # two passes
    write(wr, "WRITE_SIZE", [(n, v / 10) for n, v in one_pass] * 2)
    out = tmp_path / "o.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), str(fetch), str(wr), str(out), "32", "2"],
                   check=True, capture_output=True)
    d = json.load(open(out))
    k = d["kernels"]
    kb = 1024.0
    assert k["void gemm_nn_fast_kernel<256, 0, false, 6>"]["read_bytes_per_launch"] == 2 * 100.0 * kb
    assert k["void gemm_nt_whole_kernel<false, 16, 4, 2, 2, 4>"]["read_bytes_per_launch"] == 500.0 * kb     # 64-byte requests: as tallied
    assert k["void gemm_nt_whole_kernel<true, 32, 2, 2, 4, 2>"]["read_bytes_per_launch"] == 2 * 300.0 * kb
    seq = d["sequence"]
    assert seq["gemm_nn"] == [round(2 * 100.0 * kb + 10.0 * kb)]
    # one entry per ctn_gemm_nt call: the kernel plus its reduction launch
    assert seq["gemm_nt"] == [round((500.0 + 2 * 60.0) * kb + (50.0 + 6.0) * kb), round((2 * 300.0 + 2 * 30.0) * kb + (30.0 + 3.0) * kb)]
    assert d["passes"] == 2 and d["batch_per_gpu"] == 32
