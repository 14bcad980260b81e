#!/bin/bash
# Build one library per set of -D switches (experiments; output git-ignored under profiles/tools/variants/).
#   profiles/tools/build_defs.sh name1 "-DAL_NT=1" name2 "-DAL_NT=3" ...
set -e
cd "$(dirname "$0")/../.."
C=audiblelight_amd/csrc; V=profiles/tools/variants; mkdir -p $V; rm -f $V/lib_*.so $V/index.txt
HIPCC=/opt/rocm/bin/hipcc
while [ $# -gt 1 ]; do
  name=$1; defs=$2; shift 2
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC $defs -c $C/al_kernels.hip -o $V/k_$name.o &
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $defs -c $C/al_transforms.hip -o $V/t_$name.o &
  wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC $V/k_$name.o $V/t_$name.o -o $V/lib_$name.so
  echo "$name: $defs" >> $V/index.txt
done
