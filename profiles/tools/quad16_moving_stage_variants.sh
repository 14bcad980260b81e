# forward-transform time of moving batches (tools/quad16_moving_stages.py N Lir) per library variant under profiles/tools/mfvariants/
R=${GRAFT_REPO_ROOT:-.}; cd $R
for L in ${LIRS:-65536 81920 131072}; do
  for lib in profiles/tools/mfvariants/lib_*.so; do
    echo "== Lir=$L $(basename $lib)"; AUDIBLELIGHT_HIP_LIB=$PWD/$lib python3 profiles/tools/quad16_moving_stages.py ${N:-32} $L 2>&1 | grep -v amdgpu.ids | grep -A1 "^lb 14" | grep forward
  done
done
