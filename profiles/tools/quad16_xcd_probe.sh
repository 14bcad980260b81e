R=${GRAFT_REPO_ROOT:-.}; cd $R
for L in 81920 65536; do
  for lib in profiles/tools/mfvariants/lib_*.so; do
    echo "== Lir=$L $(basename $lib) AL_FLAG_IR_RUN(4)"; AL_EXTRA_FLAGS=$((4<<24)) AUDIBLELIGHT_HIP_LIB=$PWD/$lib python3 profiles/tools/quad16_moving_stages.py 32 $L 2>&1 | grep -v amdgpu.ids | grep -A1 "^lb 14" | grep forward
  done
done
