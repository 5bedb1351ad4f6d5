"""DPRNN-TasNet step time under the two BLAS back ends torch offers for its fp32 GEMMs.  usage: [batch]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in ("default", "hipblaslt", "cublas"):
    env = dict(os.environ, CTN_DPRNN_BLAS=lib)
    code = ("import os, sys, runpy, torch\n"
            "lib = os.environ['CTN_DPRNN_BLAS']\n"
            "if lib != 'default': torch.backends.cuda.preferred_blas_library(lib)\n"
            "print('blas', lib, torch.backends.cuda.preferred_blas_library())\n"
            "sys.argv = ['bench_dprnn.py'] + sys.argv[1:]\n"
            "runpy.run_path(os.path.join(%r, 'tools', 'bench_dprnn.py'), run_name='__main__')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code] + sys.argv[1:], env=env, capture_output=True, text=True, timeout=280)
    print(r.stdout[-2500:])
    if r.returncode:
        print(r.stderr[-1500:])
