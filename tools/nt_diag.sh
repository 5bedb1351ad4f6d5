set -e
for m in 0 2 3; do echo "SPLIT=$m base"; CTN_GEMM_SPLIT=$m timeout -k 10 100 python tools/bench_nt_time.py 32; done
for v in NT_NO_MFMA NT_NO_SPLIT; do for m in 2 3; do echo "SPLIT=$m $v"; CTN_LIB=$PWD/tools/variants/libctn_$v.so CTN_GEMM_SPLIT=$m timeout -k 10 100 python tools/bench_nt_time.py 32; done; done
