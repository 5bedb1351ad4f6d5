#!/bin/bash
# Diagnostic builds with several CTN_DBG_* macros at once: each argument is NAME=MACRO1,MACRO2 (timing only).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
for spec in "$@"; do
  name=${spec%%=*}; macros=${spec#*=}
  defs=""; for m in ${macros//,/ }; do defs="$defs -DCTN_DBG_$m"; done
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics -shared $defs \
     -o tools/variants/libctn_$name.so speech_amd/csrc/*.hip &
done
wait
ls tools/variants
