# capsule-loop accumulate of clips longer than 24 blocks: partition spectra staged through LDS (default) vs the register version
# (AL_EXTRA_FLAGS bit 13), and the headline shape for reference
for f in 8192 0 8192 0; do AL_EXTRA_FLAGS=$f python3 profiles/tools/lockstep_probe.py 2>&1 | grep La= | sed "s/^/flags=$f /"; done
