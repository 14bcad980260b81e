"""Defaults shared with the reference (audiblelight/config.py:7-11,22-25)."""
SAMPLE_RATE = 44100
BUFFER_SIZE = 8192
FFT_SIZE = 512
WIN_SIZE = 256
HOP_SIZE = 128
SCENE_DURATION = 60
DEFAULT_REF_DB = -65
MIN_REF_DB, MAX_REF_DB = -80, -50
SEED = 42  # audiblelight/utils.py:35
