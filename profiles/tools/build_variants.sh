#!/bin/bash
# Build the library with different scheduler options for the FFT translation unit (experiment; output is git-ignored).
set -e
cd "$(dirname "$0")/../.."
C=audiblelight_amd/csrc; V=profiles/tools/variants; mkdir -p $V
HIPCC=/opt/rocm/bin/hipcc
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $C/al_kernels.hip -o $V/al_kernels.o
i=0
while IFS= read -r flags; do
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c $C/al_transforms.hip -o $V/tr_$i.o
  $HIPCC --offload-arch=gfx950 -shared -fPIC $V/al_kernels.o $V/tr_$i.o -o $V/lib_$i.so
  echo "$i: $flags" >> $V/index.txt
  i=$((i+1))
done <<'LIST'

-mllvm -amdgpu-sched-strategy=max-ilp
-mllvm -amdgpu-sched-strategy=max-memory-clause
-mllvm -amdgpu-schedule-metric-bias=0
-mllvm -amdgpu-sched-strategy=gcn-iterative-ilp
-mllvm -amdgpu-schedule-relaxed-occupancy
-mllvm -amdgpu-use-amdgpu-trackers
-mllvm -enable-post-misched=0
LIST
