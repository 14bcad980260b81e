/*
 * audiblelight_hip.h -- C ABI of the MI355X (gfx950) spatial-audio synthesis path.
 *
 * Drop-in boundary for the hot path of AudibleLight's audiblelight/synthesize.py (reference paths
 * below are relative to the AudibleLight repository).  The reference has no FFI of its own: its
 * boundary is a set of Python functions that mutate Scene/Event objects.  The
 * Python mirror of those functions lives in audiblelight_amd/synthesize.py and calls ONLY the entry
 * points declared here (ctypes).  Every pointer marked "device" is a HIP device pointer to
 * contiguous memory owned by the caller; the library never allocates, never frees, keeps no
 * global state and never synchronises: every call only enqueues kernels on `stream`.
 *
 * Error convention: 0 = success, negative = AL_E_*; al_last_error() returns a thread-local string.
 *
 * Algorithm: uniformly partitioned overlap-save convolution with block size
 * B = 2^log2_block.  Spectra of real 2B-sample windows are stored as B complex floats: bin 0
 * packs (DC, Nyquist) in (re, im), bins 1..B-1 are ordinary.
 */
#ifndef AUDIBLELIGHT_HIP_H
#define AUDIBLELIGHT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of THIS header's struct layouts and entry points.  al_abi_version() returns the value the library was built
 * with; a host must compare it with the AL_ABI_VERSION it was compiled against before passing any struct (al_batch and
 * al_mix grew fields in version 2: clip_scale, xspec/hspec_zero_block, ambience, ambience_scale; version 3 puts
 * struct_size + abi_version at the head of both, so a descriptor built against another header is refused, not misread;
 * version 4 appends al_batch.emitter_parts; version 5 adds the host-side planner (al_plan_*, al_workspace_bytes,
 * al_plan_mixdown) and the AL_FLAG_QUAD_SPECTRA / AL_FLAG_FUSED_MOVING path: no struct changed; version 6 adds
 * al_plan_batch_flags -- the dispatch policy (layout + accumulate flags per chunk) moves from the hosts into the library --
 * and al_scale_rows_f64; the opt-in fused kernels of versions 2-5 (AL_FLAG_FUSED_STATIC / _FUSED_MOVING / _FUSED_NJ5,
 * al_mac_synthesis, al_fused_supported, al_moving_fused_supported, the quad layout at B = 8192, the fused_moving argument of
 * al_plan_emitter_parts) are gone: measured 20-35 % slower than the stored-spectra path, profiles/r05c_fused_ab.txt). */
#define AL_ABI_VERSION 6

#define AL_OK 0
#define AL_E_BADARG (-1)
#define AL_E_HIP (-2)
#define AL_E_UNSUPPORTED (-3)

/* al_batch.flags */
#define AL_FLAG_NO_IR_NORM 1 /* IRs are already normalised: emitter_gain := 1 (time_invariant_convolution,
                                 time_variant_convolution called directly, synthesize.py:71,277) */

#define AL_FLAG_SPLIT_SPECTRA 32 /* spectra in the split layout of csrc/al_split.h (even bins | odd bins, every window transformed
                                    as two half-size FFTs); all of al_ir_spectra / al_signal_spectra / al_block_synthesis must
                                    see the same setting; the accumulate does not care.  B >= 2048 */
#define AL_FLAG_STATIC_MAC 64   /* al_spectral_mac: static (one-emitter) events of batches with at most 21 partitions through k_spectral_mac_static (one workgroup per
                                   (event, k-tile, bin tile) looping over the capsules); the tile kernels skip them */
#define AL_FLAG_ONLY_STATIC 128 /* with AL_FLAG_STATIC_MAC: the batch has no multi-emitter event, so the other accumulate
                                   kernels are not launched at all (the host knows the event table, the library does not) */
#define AL_FLAG_QUAD_SPECTRA 256 /* with AL_FLAG_SPLIT_SPECTRA at B = 16384: the four-tile layout of csrc/al_quad16.h (every window as
                                    four 4096-point transforms) -- set both there: the one- / two-transform kernels are 15-20 %
                                    slower per scene at that block size.  Ignored at other block sizes.  al_plan_batch_flags sets
                                    the layout flags for the plan's block size */
#define AL_FLAG_NARROW_FFT 4 /* A/B switch: 16 complex values per thread at every block size (default: 32 from
                               B = 8192 up, see csrc/al_fft.h) */
#define AL_FLAG_SYNTH_RUN(n) (((n) & 0xff) << 16) /* al_block_synthesis: n consecutive blocks per workgroup (0 = 1) */
#define AL_FLAG_IR_RUN(n) (((n) & 0x7f) << 24)    /* al_ir_spectra: n consecutive partitions per workgroup (0 = 1; the B = 16384 quad-tile
                                                     kernels: 0 = chosen per batch, equal runs, csrc/al_quad16.h) */

#define AL_SPARSE_MAX_NJ 6          /* longest stream (in blocks) the sliding-window accumulate accepts */
#define AL_SPARSE_MAX_PARTITIONS 24 /* most IR partitions it accepts */
#define AL_STATIC_MAC_MAX_PARTITIONS 21 /* most IR partitions the capsule-loop accumulate (AL_FLAG_STATIC_MAC) takes */

#define AL_MIN_LOG2_BLOCK 10
#define AL_MAX_LOG2_BLOCK 14

typedef void *al_stream_t; /* hipStream_t */

/* One event as render_event_audio sees it (synthesize.py:507-608; attribute list SURVEY 8a A15). */
typedef struct {
  int64_t audio_off;  /* first sample of the mono clip inside `audio` */
  int64_t out_off;    /* first element of this event's (C, len) block inside `spatial` */
  int32_t len;        /* clip samples La = output samples per capsule (synthesize.py:553,590) */
  int32_t valid_len;  /* convolution samples kept before zero padding: La for static events,
                         min(La, n_frames*hop - win) for moving ones (synthesize.py:274,590) */
  int32_t n_blocks;   /* K = ceil(len / B) output blocks */
  int32_t stream0;    /* first entry of this event in the stream table */
  int32_t n_streams;  /* emitters: 1 static, >1 moving (synthesize.py:564-587), 0 = tiled dry clip */
  int32_t yspec_base; /* block index of (c=0,k=0) inside yspec; layout [c][k] */
  int32_t part_base;  /* index of (c=0,k=0) inside the partial-statistics array; layout [c][k] */
  float snr;          /* Event.snr */
  float ref_db;       /* Scene.ref_db (synthesize.py:598) */
  int32_t reserved;   /* 1: moving event whose streams all have n_j <= AL_SPARSE_MAX_NJ (sliding-window accumulate) */
} al_event;

/* One (event, emitter) source stream: the clip weighted by that emitter's cross-fade envelope
 * (time_variant_convolution, synthesize.py:277-310, in the envelope form of SURVEY 8a A7). */
typedef struct {
  int32_t event;    /* owning event */
  int32_t emitter;  /* column of the IR tensor (synthesize.py:662) */
  int32_t j_lo;     /* first non-zero signal block */
  int32_t n_j;      /* number of non-zero signal blocks */
  int32_t xspec_base; /* block index of block j_lo inside xspec */
  int32_t w_off;    /* offset of this stream's frame weights W[:, l] in `wtab`, -1 = static (env == 1) */
  int32_t w_len;    /* number of frames n_frames = min(F_a, W.shape[0]) (synthesize.py:208-210) */
  float gain;       /* clip gain folded into the spectrum (peak normalisation event.py:535-536,
                       sample-wise FX gain/polarity, x fft_size for moving events) */
} al_stream;

/* Everything one launch sequence needs.  All pointers are device pointers unless noted. */
typedef struct {
  int32_t struct_size;  /* sizeof(al_batch) as the CALLER compiled it, and the AL_ABI_VERSION it was compiled against: every */
  int32_t abi_version;  /* entry point refuses a descriptor whose two fields differ from the library's own (AL_E_BADARG) */
  int32_t log2_block;   /* B = 1 << log2_block */
  int32_t n_capsules;   /* C */
  int32_t n_events;     /* E */
  int32_t n_streams;    /* S */
  int32_t n_emitters;   /* columns of the IR tensor used by this batch: [emitter0, emitter0 + n_emitters) */
  int32_t ir_len;       /* Lir samples per IR row */
  int64_t ir_stride_c;  /* elements between capsules  (multiple of 4) */
  int64_t ir_stride_n;  /* elements between emitters (multiple of 4) */
  int32_t n_partitions; /* P = ceil(ir_len / B) */
  int32_t max_blocks;   /* max over events of n_blocks */
  int32_t max_nj;       /* max over streams of n_j */
  int32_t hop;          /* STFT hop of the moving path (config.py:11), 128 */
  /* Chunking.  A long scene is run as several batches ("chunks") over ONE set of global tables: a chunk
   * covers events [event0, event0+n_events), streams [stream0, stream0+n_streams) and IR columns
   * [emitter0, emitter0+n_emitters).  hspec/xspec/yspec are chunk-local workspaces that every chunk reuses
   * (so they stay in the 256 MiB Infinity Cache instead of round-tripping through HBM): global block
   * indices from the tables are rebased by xspec_block0 / yspec_block0 / emitter0.  ir_energy,
   * emitter_gain, partials, spatial, event_stats and event_scale are indexed globally. */
  int32_t event0;
  int32_t stream0;
  int32_t emitter0;
  int32_t xspec_block0;
  int32_t yspec_block0;
  int32_t flags;        /* AL_FLAG_* */

  const float *twiddle;   /* al_twiddle_init output, B complex */
  const float *audio;     /* mono clips, float32 */
  const float *ir;        /* IR tensor (C, N, Lir) float32: WorldState.get_irs() layout (worldstate.py:2183-2255) */
  const float *wtab;      /* frame weights of moving streams */
  const al_event *events;   /* E entries */
  const al_stream *streams; /* S entries */

  float *ir_energy;  /* workspace: n_emitters * C * P partial sums of ir^2 */
  float *emitter_gain; /* workspace/out: n_emitters, 1 / mean_c ||ir||  (normalize_irs, synthesize.py:404-428); 0 for an emitter whose
                          IRs are all zeros (the reference divides zeros by tiny and keeps zeros) */
  float *hspec;      /* workspace: n_emitters * C * P blocks of B complex, [n - emitter0][c][p] */
  float *xspec;      /* workspace: sum(n_j) blocks of B complex */
  float *yspec;      /* workspace: sum(C * n_blocks) blocks of B complex */
  float *spatial;    /* out: per event (C, len) float32, UNSCALED convolution truncated/padded to len */
  float *partials;   /* workspace: 4 floats per (event, c, k): sum|x|, max|x|, non-finite count, pad */
  double *event_stats; /* out: 4 doubles per event: sum|x|, max|x|, non-finite count, total scale */
  float *event_scale;  /* out: per event multiplier = apply_snr o db_to_multiplier (synthesize.py:594-599), saturated at +-FLT_MAX:
                          a silent render (all-zero clip or IRs) has 10^(dB/20) / tiny here, which the reference multiplies its
                          zeros by in float64 -- silence stays silence, never inf * 0 */
  const float *clip_scale; /* optional (NULL = 1): per event scalar applied to the clip on top of al_stream.gain; written on
                              the device by al_clip_scales (peak normalisation + folded Gain/Invert, event.py:529-536), so
                              the clip's peak never travels to the host.  Indexed globally like event_scale. */
  int32_t xspec_zero_block; /* index of an all-zero block inside xspec / hspec (-1: none): the LDS-DMA accumulate reads the rows past */
  int32_t hspec_zero_block; /* an odd partition count from it instead of masking per lane (13..21 partitions need hspec_zero_block). */
  const int32_t *emitter_parts; /* optional (NULL = n_partitions everywhere), indexed globally by IR column: how many leading
                                   partitions of that IR can reach a block some event keeps.  pad_or_truncate_audio
                                   (synthesize.py:590) drops everything from block n_blocks on, and partition p of an IR whose
                                   signal starts at block j_lo only feeds blocks >= j_lo + p; al_forward_spectra / al_ir_spectra
                                   still read the later partitions (normalize_irs needs their energy) but neither transform nor
                                   store them, and the sliding-window accumulate never reads them.  Must be n_partitions for
                                   the IR of every event that is not a sliding-window (al_event.reserved == 1) event; ignored when n_partitions >
                                   AL_SPARSE_MAX_PARTITIONS (those events then go through the tile accumulate). */
} al_batch;

/* Mixdown of one microphone (generate_scene_audio_from_events, synthesize.py:314-401). */
typedef struct {
  int32_t struct_size;    /* sizeof(al_mix) and AL_ABI_VERSION as the caller compiled them (checked like al_batch's) */
  int32_t abi_version;
  int32_t n_capsules;     /* rows of the scene buffer */
  int32_t n_samples;      /* round(scene.duration * sample_rate) (synthesize.py:331) */
  int32_t tile;           /* samples per time tile: 4096 */
  int32_t n_tiles;        /* ceil(n_samples / tile) */
  int32_t accumulate;     /* 0: scene is overwritten, 1: scene already holds ambience (synthesize.py:335-356) */
  int32_t reserved;
  const int32_t *tile_ptr;  /* n_tiles + 1 */
  const int32_t *tile_events; /* indices into the slot arrays, insertion order inside a tile */
  const int64_t *slot_src;  /* per slot: offset of the event's (C, len) block inside `spatial` */
  const int32_t *slot_len;  /* per slot: event len (row stride) */
  const int32_t *slot_start; /* per slot: max(0, round(scene_start*sr)) (synthesize.py:361) */
  const int32_t *slot_count; /* per slot: samples added = min(end-start, len) (synthesize.py:372-378) */
  const int32_t *slot_rows;  /* per slot: capsules of that event */
  const int32_t *slot_event; /* per slot: index into event_scale */
  const float *spatial;
  const float *event_scale;
  float *scene;             /* (C, n_samples) float32 */
  const float *ambience;       /* optional (NULL = none): (C, n_samples) noise, added as ambience_scale[c] * noise[c] in the */
  const float *ambience_scale; /* same pass that mixes the events (synthesize.py:350-356 fused with :358-383): the scene is */
                               /* written once instead of zeroed and read-modify-written twice.  n_capsules floats: the */
                               /* noise-floor multiplier, times 1/peak of the channel when the noise is handed over */
                               /* un-normalised (al_ambience_scales) */
} al_mix;

const char *al_last_error(void);
int al_abi_version(void);

/* Bytes of the twiddle table for block 2^log2_block; al_twiddle_init fills it (device). */
int64_t al_twiddle_bytes(int log2_block);
int al_twiddle_init(float *twiddle, int log2_block, al_stream_t stream);

/* Stage entry points (each only enqueues).  al_render_batch = stages 1-6 in order. */
int al_ir_spectra(const al_batch *b, al_stream_t stream);      /* A1 energy partials + IR partition spectra */
int al_emitter_gains(const al_batch *b, al_stream_t stream);   /* A1 normalize_irs scalar per emitter */
int al_signal_spectra(const al_batch *b, al_stream_t stream);  /* A13 gain + A7 envelope + block spectra */
int al_forward_spectra(const al_batch *b, al_stream_t stream); /* al_ir_spectra + al_signal_spectra (independent of each other):
                                                                  one launch in the split layout, else the two in turn */
int al_spectral_mac(const al_batch *b, al_stream_t stream);    /* A2/A7 frequency-domain accumulate */
/* Which kernel instantiations al_spectral_mac launches for this batch (no launch; the launcher and this call read the
 * SAME descriptor, so the parity tests' assertion of the regime they cover cannot drift).  *static_code = the kernel that
 * takes one-emitter events: 1000000*KSPLIT + 10000*KT + 100*PT + VB (k-tile, partition tile, bins per thread of the tile
 * kernel k_spectral_mac), or 3120000 + 100*P + D for the capsule-loop kernels (AL_FLAG_STATIC_MAC, P <= 21 partitions):
 * D = 1: k_spectral_mac_static<12,P,1> (clips of at most 12 blocks), 2: <12,P,2> (13..24 blocks), 3: the partition spectra
 * staged through LDS by registers, k_spectral_mac_static_lds<12,P> (more than 24 blocks) or <12,ceil(P/2),2> (13..16
 * partitions without hspec_zero_block), 4: staged by LDS-DMA, k_spectral_mac_static_glds (13..21 partitions as two or three
 * units per capsule, any clip length; needs hspec_zero_block >= 0); *moving_code = 100*NJW + PT of the sliding-window kernel for moving events,
 * 0 = not launched. */
int al_spectral_mac_variant(const al_batch *b, int32_t *static_code, int32_t *moving_code);
int al_block_synthesis(const al_batch *b, al_stream_t stream); /* inverse FFT, A3 truncate/pad, A4/A5 statistics */
int al_event_levels(const al_batch *b, al_stream_t stream);    /* A9 composite level law -> event_scale */
/* The two halves of al_event_levels, for a scene whose capsules are sharded over several GPUs: every rank reduces
 * its own capsules into event_stats[e] = {sum|x|, max|x|, non-finite count, -}, the ranks all-reduce those E triples
 * (SUM, MAX, SUM), then every rank evaluates the level law with the TOTAL capsule count. */
int al_event_stats(const al_batch *b, al_stream_t stream);
/* The same split for normalize_irs (synthesize.py:404-428), whose mean runs over ALL capsules of the microphone:
 * al_emitter_norm_sums leaves sum_c ||h_{n,c}|| over this rank's capsules in emitter_gain[n]; the ranks all-reduce (SUM)
 * that device array; al_emitter_gains_from_sums turns it into total_capsules / sum in place.  No host round trip. */
int al_emitter_norm_sums(const al_batch *b, al_stream_t stream);
int al_emitter_gains_from_sums(const al_batch *b, int32_t total_capsules, al_stream_t stream);
int al_event_levels_from_stats(const al_batch *b, int32_t total_capsules, al_stream_t stream);
int al_render_batch(const al_batch *b, al_stream_t stream);

/* A11 mixdown and helpers. */
int al_mixdown(const al_mix *m, al_stream_t stream);
/* x[r, :] *= scale[r_index] for a (rows, cols) block: scales an event's spatial audio in place
 * (event.spatial_audio, synthesize.py:599,606). scale is a device pointer to ONE float. */
int al_scale_rows(float *x, int64_t n, const float *scale, al_stream_t stream);
/* The same with the scalar read from a DOUBLE on the device (rounded to float32 once): al_batch.event_stats[4 e + 3], the
 * noise-floor multiplier db_to_multiplier(ref_db + snr, mean|x|) alone, which scales the dry render (synthesize.py:598,608 ->
 * :432-504) -- event_scale[e] also carries the apply_snr factor and is not the reference's `event_scale`. */
int al_scale_rows_f64(float *x, int64_t n, const double *scale, al_stream_t stream);
/* A13 on the device, without a host round trip.  al_clip_scales: for every event of the batch, clip_scale[e] =
 * prescale[e] (mode[e] == 0) or prescale[e] / (|prescale[e]| * max|clip_e| + tiny(float32)) (mode[e] == 1): the
 * peak normalisation `a / max(|a| + tiny)` of event.py:535-536 applied to the clip prescale[e] * clip_e, i.e. behind a
 * chain of pure scalar FX (Gain: 10^(dB/20), Invert: -1; augmentation.py:1105-1136,1557-1580).  prescale (float32[E]) and
 * mode (int32[E]) are device arrays indexed like al_batch.events.  al_peak_scale: the same scalar for one buffer. */
int al_clip_scales(const al_batch *b, const float *prescale, const int32_t *mode, al_stream_t stream);
int al_peak_scale(const float *x, int64_t n, float prescale, float *scale_out, al_stream_t stream);
/* y += a * x over n floats; a = *a_dev (ambience add, synthesize.py:350-356). */
int al_axpy(float *y, const float *x, const float *a_dev, int64_t n, al_stream_t stream);
/* Row statistics of a (rows, cols) float32 matrix: out[r] = {sum|x|, max|x|, non-finite count, sum x^2} (doubles). */
int al_row_stats(const float *x, int32_t rows, int64_t cols, float *partials, double *out, al_stream_t stream);
int64_t al_row_stats_partials(int32_t rows, int64_t cols); /* floats needed in `partials` */

/* ---- Sample-wise clip operations (A13 peak normalisation, A14 stateless FX; audiblelight/augmentation.py).
 * All work on float32 device buffers of n samples; ops marked (o) are out-of-place (dst != src).
 * `params` / `iparams` are HOST pointers (scalars copied into the kernel arguments). */
#define AL_FX_GAIN 1      /* x *= p[0]                      Gain (augmentation.py:1105-1136), p[0] = 10^(gain_db/20) */
#define AL_FX_INVERT 2    /* x = -x                         Invert (1557-1580) */
#define AL_FX_REVERSE 3   /* (o) dst[t] = src[n-1-t]        Reverse (1583-1601) */
#define AL_FX_FADE 4      /* x *= fade_in(t) * fade_out(t)  Fade (1403-1554): ip = {n_in, n_out, shape_in, shape_out} */
#define AL_FX_CLIP 5      /* clamp(x, -p[0], p[0])          Clipping (832-868), p[0] = 10^(threshold_db/20) */
#define AL_FX_TANH 6      /* tanh(p[0] * x)                 Distortion (927-960), p[0] = 10^(drive_db/20) */
#define AL_FX_BITCRUSH 7  /* rint(x * p[0]) / p[0]          Bitcrush (266-300), p[0] = 2^bit_depth */
#define AL_FX_PREEMPH 8   /* (o) y[n] = x[n] - c x[n-1], y[0] = x[0] + (2x[0] - x[1])   Preemphasis (1350-1385) */
#define AL_FX_DEEMPH 9    /* (o) inverse of PREEMPH (IIR + extrapolation correction)    Deemphasis (1388-1400) */
#define AL_FADE_LINEAR 0
#define AL_FADE_EXPONENTIAL 1
#define AL_FADE_LOGARITHMIC 2
#define AL_FADE_QUARTER_SINE 3
#define AL_FADE_HALF_SINE 4
#define AL_FADE_NONE 5
int al_fx_apply(int op, const float *src, float *dst, int64_t n, const float *params, const int32_t *iparams,
                al_stream_t stream);
/* TimeWarp* (augmentation.py:1604-1790): dst[t] for t < n is taken from the concatenation of `n_rows` rows of
 * `row_len` samples, row q = {src row r = rows[2q], mode = rows[2q+1]: 0 copy, 1 zeros, 2 reversed}, where src row r
 * is src[r + frame_len * j], j < row_len (the reference iterates librosa.util.frame's (frame_len, n_frames) matrix
 * by rows); the concatenation is wrap-extended to n samples (Augmentation.process, augmentation.py:117-123). */
int al_fx_frame_shuffle(const float *src, float *dst, int64_t n, int32_t frame_len, int32_t row_len,
                        const int32_t *rows, int32_t n_rows, al_stream_t stream);
/* ---- Ambience (A12): Timmer-Koenig (1/f)^beta noise, audiblelight/ambience.py:271-375.
 * The host draws the two standard-normal sets with numpy's default_rng(seed) (PCG64 + ziggurat, ambience.py:351-356:
 * the reference's RNG stream is data-dependent and is not re-implemented on the device); everything after the draws
 * runs here: spectral shaping, DC/Nyquist fix-up, an inverse real FFT of ARBITRARY length n, the 1/sigma scale. */
/* floats of workspace al_noise_irfft needs for `rows` series of length n */
int64_t al_noise_workspace_floats(int32_t rows, int64_t n);
/* out[r, :] = irfft((zr + i zi) * shape)[r] / sigma ; zr, zi: (rows, n/2+1) float32 draws; shape: n/2+1 float32. */
int al_noise_irfft(const float *zr, const float *zi, const float *shape, int32_t rows, int64_t n, float inv_sigma,
                   float *out, float *workspace, al_stream_t stream);
/* The same with the draws made ON THE DEVICE (csrc/al_rng.h: Philox-4x32-10 counters + Box-Muller; (zr, zi) of bin f of row r
 * is a pure function of (seed, r, f)): nothing is drawn, cast or uploaded by the host.  The realisation differs from numpy's
 * for the same seed -- the reference's own tests of this function are statistical (tests/test_ambience.py:30-67) plus
 * fixed-seed reproducibility (:70-76), which this keeps.  shape == NULL: flat (white). */
int al_noise_irfft_seeded(uint64_t seed, const float *shape, int32_t rows, int64_t n, float inv_sigma, float *out,
                          float *workspace, al_stream_t stream);
/* out[i] = scale * N(0,1) for i < n, element i = normal (i % 4) of Philox counter block i / 4 under (seed, tag): "gaussian"
 * ambience (ambience.py:160-165) and white noise, whose Timmer-Koenig synthesis is iid Gaussian in time (flat spectrum). */
int al_normal_fill(float *out, int64_t n, uint64_t seed, uint32_t tag, float scale, al_stream_t stream);
/* Philox-4x32-10 on the HOST (the same function the kernels call), for known-answer tests: all arguments host pointers. */
int al_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);
/* scales[c] = db_to_multiplier(ref_db, mean|normalised noise|) / (normalize ? max|noise_c| + tiny : 1) from al_row_stats'
 * output, on the device (ambience.py:211-214 + synthesize.py:350-356 as one scalar per channel; rows <= 1024).
 * normalize == 2: scales[c] = 1 / (max|noise_c| + tiny) alone (Ambience.load_ambience(normalize=True) without the floor). */
int al_ambience_scales(const double *row_stats, int32_t rows, int64_t cols, float ref_db, int32_t normalize, float *scales,
                       al_stream_t stream);
/* y[r, :] += a_dev[r] * x[r, :] over a (rows, cols) matrix: second and further ambiences of a scene. */
int al_axpy_rows(float *y, const float *x, const float *a_dev, int32_t rows, int64_t cols, al_stream_t stream);
/* x[r, :] *= scale[r]: per-channel peak normalisation (ambience.py:211-214) after al_row_stats. */
int al_scale_matrix_rows(float *x, int32_t rows, int64_t cols, const float *scale, al_stream_t stream);

/* ---- STFT-domain intermediates of the moving-source path (A7).  The render path evaluates the same result in the
 * envelope form (al_signal_spectra / al_spectral_mac); these three give the reference's public helper functions of the
 * same names a device implementation with the reference's array layouts (C order, complex64 as interleaved floats).
 * Any fft_size: Stockham passes of radix 2 / 3 / 4 / 5 / 7 where it factors into those, Bluestein's chirp-z on a power-of-two length
 * otherwise (numpy's rfft / irfft in the reference take any size, synthesize.py:135,263); the workspace functions account for it. */
/* stft (synthesize.py:109-145): y (rows, n) float32 -> spec (rows, n_frames, fft_size/2+1) complex64,
 * n_frames = 2*ceil(n / (2*hop)) + 1, window sin^2(pi t / win), left pad win-hop, rfft norm="backward". */
int64_t al_stft_workspace_floats(int64_t series /* rows * n_frames */, int32_t fft_size);
int al_stft(const float *y, int64_t rows, int64_t n, int32_t fft_size, int32_t win_size, int32_t hop_size, float *spec,
            float *workspace, al_stream_t stream);
/* perform_time_variant_convolution (synthesize.py:184-252): s_audio (F_a, n_freq), s_ir (F_ir, n_freq, n_ch, n_irs)
 * complex64, w_ir (F_w, n_irs) float32 -> out (n_frames, n_freq, n_ch) complex64, n_frames = min(F_a, F_w) (any count: launched in groups of frames). */
int al_tv_stft_mac(const float *s_audio, const float *s_ir, const float *w_ir, int32_t n_frames, int32_t n_frames_ir,
                   int32_t n_freq, int32_t n_ch, int32_t n_irs, float *out, al_stream_t stream);
/* istft_overlap_synthesis (synthesize.py:255-274): spatial_stft (n_frames, n_freq, n_ch) complex64 -> out
 * (n_frames*hop - win, n_ch) float32: irfft(n=fft_size, norm="forward"), overlap-add at i*hop, slice [win, n_frames*hop). */
int64_t al_istft_workspace_floats(int32_t n_frames, int32_t n_ch, int32_t fft_size);
int al_istft_ola(const float *spatial_stft, int32_t n_frames, int32_t n_freq, int32_t n_ch, int32_t fft_size,
                 int32_t win_size, int32_t hop_size, float *out, float *workspace, al_stream_t stream);

/* IR ingest: WorldState.get_irs() hands out float64 (C, N, L) (worldstate.py:2183-2255); this
 * converts to the float32 layout al_batch wants (row pitch `dst_pitch` >= L, multiple of 4, pad zeroed) on the device,
 * so the host never casts 1.6 GB per scene.  rows = C * N. */
int al_pack_irs_f64(const double *src, float *dst, int64_t rows, int32_t len, int32_t dst_pitch, al_stream_t stream);
/* The same re-pitching for float32 IRs whose length is not a multiple of 4 (ragged listeners padded to the longest IR,
 * worldstate.py:2236-2253): the caller's array goes to HBM as it is and is laid out on the device. */
int al_pack_irs_f32(const float *src, float *dst, int64_t rows, int32_t len, int32_t dst_pitch, al_stream_t stream);

/* Ragged ingest: the ray tracer hands out one 1-D IR per (capsule, source) of varying length, which the reference
 * zero-pads into (C, N, maxlen) with a four-deep Python loop run twice (worldstate.py:2196-2253).  Here the IRs are
 * concatenated once (`src`, float32 or float64) and laid out on the device: row r = src[offsets[r] : offsets[r]+lens[r]]
 * followed by zeros up to dst_pitch (multiple of 4).  offsets / lens are device arrays. */
int al_pack_ragged_irs(const void *src, int32_t src_is_f64, const int64_t *offsets, const int32_t *lens, int64_t rows,
                       int32_t dst_pitch, float *dst, al_stream_t stream);
/* Rational-ratio resampling of `rows` series (SOFA IRs at another sample rate, worldstate.py:2995-3006): polyphase FIR,
 * out[m] = sum_j x[j] * taps[m*down - j*up + half_len], m < n_out (zeros up to out_pitch): scipy.signal.resample_poly
 * semantics with the caller's `taps` (2*half_len+1 floats, already scaled by `up`).  The reference calls
 * librosa.resample (soxr), an un-vendored dependency: parity with IT is unpinned; this is pinned to resample_poly. */
int al_resample_poly(const float *x, int32_t rows, int64_t n_in, const float *taps, int32_t half_len, int32_t up, int32_t down,
                     float *out, int64_t n_out, int64_t out_pitch, al_stream_t stream);

/* Output encoding for the WAV writer: (C, T) float32 scene -> (T, C) interleaved frames as
 * soundfile.write(mic_audio.T, sr) stores them (core.py:1840-1847).  AL_FRAMES_PCM16 is soundfile's default subtype for
 * WAV: python-soundfile enables clipping on every file, so libsndfile's f2s_clip_array applies: int16 = lrintf(x * 32768)
 * saturated to [-32768, 32767]; AL_FRAMES_F32 keeps float32.
 * Done on the device so the D2H copy is already the file payload (half the bytes for PCM_16).  `out` may be page-locked host
 * memory; capsule counts that are multiples of 8 (PCM_16) / 4 (float32) are stored 16 bytes per lane and need a 16-byte
 * aligned `out`. */
#define AL_FRAMES_F32 0
#define AL_FRAMES_PCM16 1
int al_encode_frames(const float *scene, int32_t n_capsules, int64_t n_samples, int32_t format, void *out, al_stream_t stream);

/* ---- Planning (host side, no device call; csrc/al_plan.cpp).  Everything al_batch / al_mix point to is index arithmetic on
 * shapes: which signal blocks a moving event's cross-fade windows touch (generate_interpolation_matrix, synthesize.py:148-181),
 * where every event's spectra, statistics and audio live, which events overlap which mixdown tile (event slots with Python's
 * round-half-even, synthesize.py:361-362).  A host builds the tables here, copies them to the device and fills al_batch / al_mix
 * with the device pointers.  The plan owns its arrays; the accessors return HOST pointers valid until al_plan_destroy. */
typedef struct {
  int32_t n_samples;   /* clip length La */
  int32_t n_emitters;  /* len(event): 0 = clip tiled over the capsules, 1 = static, > 1 = moving (synthesize.py:564-587) */
  int32_t emitter0;    /* first IR column of this event (synthesize.py:662) */
  int32_t is_moving;
  float snr;           /* Event.snr */
  float ref_db;        /* Scene.ref_db */
  float gain;          /* scalar folded into the clip (peak normalisation, FX gain / polarity); 1 = none */
  int32_t stft_len;    /* samples the STFT frame count is taken from; 0 = n_samples */
  double duration;     /* Event.duration in seconds (moving events) */
} al_event_spec;

typedef struct {
  int32_t log2_block, n_capsules, ir_len, n_events, n_streams, n_emitters, n_partitions, hop, fft_size;
  int32_t max_blocks;       /* al_batch.max_blocks of the whole batch */
  int32_t max_nj;           /* al_batch.max_nj */
  int32_t max_nj_sliding;   /* longest stream of the sliding-window events */
  int32_t xspec_blocks, yspec_blocks, n_partials;   /* blocks of B complex / entries of 4 floats */
  int32_t reserved;
  int64_t hspec_blocks;     /* n_emitters * n_capsules * n_partitions */
  int64_t audio_floats, spatial_floats, wtab_floats;
} al_plan_info;

typedef struct al_plan al_plan;
typedef struct al_mix_plan al_mix_plan;

typedef struct {            /* what al_mix needs, as host arrays owned by the al_mix_plan */
  int32_t n_capsules, n_samples, tile, n_tiles, n_slots, n_tile_events, n_skipped, reserved;
  const int32_t *tile_ptr;     /* n_tiles + 1 */
  const int32_t *tile_events;  /* max(n_tile_events, 1) */
  const int64_t *slot_src;     /* max(n_slots, 1) entries each, as al_mix documents them */
  const int32_t *slot_len, *slot_start, *slot_count, *slot_rows, *slot_event;
  const int32_t *skipped;      /* n_skipped: events whose slot is empty after rounding (synthesize.py:364-370 warns and skips) */
} al_mix_tables;

typedef struct {            /* the al_batch fields of a chunk of consecutive events (al_plan_chunk) */
  int32_t event0, n_events, stream0, n_streams, emitter0, n_emitters, xspec_block0, xspec_blocks, yspec_block0, yspec_blocks;
  int32_t max_blocks, max_nj;
} al_chunk;

const char *al_plan_last_error(void);
/* The planner is plain host code: it is exported by libaudiblelight_hip.so AND by libaudiblelight_plan.so (csrc/al_plan.cpp alone,
 * no HIP dependency), for hosts that plan on a machine without ROCm; al_plan_abi_version() is what that library answers to. */
int al_plan_abi_version(void);
int32_t al_choose_log2_block(int32_t ir_len, int32_t max_clip);
int32_t al_stft_frame_count(int64_t n_samples, int32_t hop);                          /* synthesize.py:123 */
int32_t al_interpolation_rows(int32_t n_irs, double duration, double sample_rate, int32_t hop);
/* generate_interpolation_matrix (synthesize.py:148-181) for ir_times = linspace(0, duration, n_irs): rows x n_irs float64 */
int al_interpolation_matrix(int32_t n_irs, double duration, double sample_rate, int32_t hop, int32_t rows, double *weights);
/* log2_block <= 0: chosen from the lengths (al_choose_log2_block: 8192 from 4096 samples of IR and clip up), and 16384 in the two
 * situations it was measured to pay (csrc/al_plan.cpp): batches of at least 100 000 (capsule, block) rows of static events with
 * 17..24 partitions of 8192, and batches whose moving events are off the sliding-window accumulate at 8192 but on it at 16384.  Read
 * the choice back from al_plan_info.log2_block and set the layout flags for it (AL_FLAG_SPLIT_SPECTRA at 13, + AL_FLAG_QUAD_SPECTRA
 * at 14).  Errors carry the reference's messages ("Moving Event has only one emitter!", ...). */
int al_plan_create(const al_event_spec *specs, int32_t n_events, int32_t n_capsules, int32_t ir_len, double sample_rate,
                   int32_t log2_block, int32_t hop, int32_t win, int32_t fft_size, al_plan **out);
void al_plan_destroy(al_plan *plan);
int al_plan_get_info(const al_plan *plan, al_plan_info *info);
const al_event *al_plan_events(const al_plan *plan);          /* n_events */
const al_stream *al_plan_streams(const al_plan *plan);        /* n_streams */
const float *al_plan_wtab(const al_plan *plan);               /* wtab_floats */
const int64_t *al_plan_audio_offsets(const al_plan *plan);    /* n_events: where each clip goes inside `audio` */
/* A run of consecutive events as one al_batch over the plan's global tables (a scene too large for one spectra workspace is
 * rendered chunk by chunk over ONE reused workspace sized for the largest chunk; results are identical to one batch). */
int al_plan_chunk(const al_plan *plan, int32_t event0, int32_t n_events, al_chunk *out);
/* bytes of the spectra workspaces + statistics of the whole batch as ONE chunk (hspec, xspec, yspec, ir_energy, emitter_gain, partials) */
int64_t al_workspace_bytes(const al_plan *plan);
/* al_batch.emitter_parts: returns 1 and fills out[n_emitters] if the batch needs the table, 0 if every IR needs all partitions */
int al_plan_emitter_parts(const al_plan *plan, int32_t *out);
/* Dispatch policy: al_batch.flags for a chunk of the plan (chunk == NULL: the whole plan as one batch) -- the layout flags for
 * the plan's block size (AL_FLAG_SPLIT_SPECTRA at 8192, + AL_FLAG_QUAD_SPECTRA at 16384) and the accumulate flags for the chunk's
 * event mix (AL_FLAG_STATIC_MAC when it has one-emitter events and at most AL_STATIC_MAC_MAX_PARTITIONS partitions, +
 * AL_FLAG_ONLY_STATIC when it has no multi-emitter event).  This IS the configuration bench.py times: a host that ORs the result
 * into al_batch.flags (plus AL_FLAG_NO_IR_NORM where it applies) runs the same kernels as audiblelight_amd/engine.py, which
 * calls this too; al_spectral_mac_variant on the finished descriptor tells which. */
int al_plan_batch_flags(const al_plan *plan, const al_chunk *chunk, int32_t *flags);
int al_plan_mixdown(const double *starts, const double *ends, const int32_t *lens, const int32_t *rows, const int64_t *src_offsets,
                    const int32_t *event_index, int32_t n, double duration, double sample_rate, int32_t n_capsules, int32_t tile,
                    al_mix_plan **out);
void al_mix_plan_destroy(al_mix_plan *plan);
int al_mix_plan_get(const al_mix_plan *plan, al_mix_tables *tables);

/* dst[t] = src[t mod m] for t < n: np.pad(..., mode="wrap") of Augmentation.process. */
int al_wrap_copy(const float *src, int64_t m, float *dst, int64_t n, al_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
