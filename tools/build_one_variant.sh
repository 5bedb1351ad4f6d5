#!/bin/bash
# usage: build_variant.sh <out.so> <file.hip> <extra flags...>: builds the library with one source recompiled with extra flags
out=$1; src=$2; shift 2
cd /root/repo/speech_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics -Wall -Wno-unused-function -fno-gpu-rdc "$@" -c $src -o /tmp/variant.o || exit 1
objs=$(ls build/*.o | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs /tmp/variant.o
