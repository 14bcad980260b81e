#!/bin/bash
# Sanitizer pass over the kernel SOURCE on the CPU (GPU AddressSanitizer is not available on the pool):
# builds audiblelight_amd/csrc/*.hip for the host with -fsanitize=address,undefined through the emulation header
# and renders a small mixed batch (static, moving, zero-emitter events, chunked, mixdown).
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/libal_asan.so
g++ -std=c++17 -O1 -g -DHOSTEMU_THREADS -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -pthread -x c++ \
    -I "$ROOT/tests/hostemu" "$ROOT/audiblelight_amd/csrc/al_kernels.hip" "$ROOT/audiblelight_amd/csrc/al_transforms.hip" \
    "$ROOT/audiblelight_amd/csrc/al_plan.cpp" -o "$OUT"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python "$ROOT/tests/hostemu/asan_run.py" "$OUT"
