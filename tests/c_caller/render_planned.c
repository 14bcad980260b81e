/* TEST INFRASTRUCTURE: a plain C11 host (no Python, no torch, no C++) that renders a whole scene -- static, moving and tiled
 * events, an ambience, optionally in chunks over one reused workspace -- through the C ABI of include/audiblelight_hip.h,
 * with EVERY table coming from the library's own planner (al_plan_create / al_plan_chunk / al_plan_emitter_parts /
 * al_plan_mixdown): the host derives nothing itself.  tests/test_gpu_c_caller.py writes the inputs, runs this binary on the
 * MI355X and compares what it writes with the float64 oracle, every row.
 *
 * Reference call sequence reproduced: render_audio_for_all_scene_events (synthesize.py:613-677) then
 * generate_scene_audio_from_events (:314-401) with one noise ambience (:335-356, ambience.py:211-214 per-channel peak
 * normalisation).
 *
 *   render_planned <in.bin> <out.bin>
 *   in.bin : int32 C, E, N (IR columns), Lir, log2_block (0 = library's choice), chunk_events (0 = one batch), has_ambience,
 *                  reserved (0)
 *            float32 ref_db, sample_rate, duration, ambience_ref_db
 *            E x { int32 n_samples, n_emitters, emitter0, is_moving; float32 snr, scene_start }
 *            float32 clips (concatenated), float32 irs[C][N][Lir], float32 noise[C][T] if has_ambience (T = round(duration * sr))
 *   out.bin: float32 event_scale[E]; per event float32 spatial[C][n_samples] (unscaled); float32 scene[C][T]
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "audiblelight_hip.h"

#define HIP_OK(call)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return 2;                                                                             \
    }                                                                                       \
  } while (0)
#define AL_CALL(call)                                                                     \
  do {                                                                                  \
    if ((call) < 0) {                                                                   \
      fprintf(stderr, "%s:%d: %s -> %s / %s\n", __FILE__, __LINE__, #call, al_last_error(), al_plan_last_error()); \
      return 3;                                                                         \
    }                                                                                   \
  } while (0)

static void *dev_alloc(size_t bytes) {
  void *p = NULL;
  if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return NULL;
  return p;
}
static void *dev_copy(const void *src, size_t bytes, hipStream_t stream) {
  void *p = dev_alloc(bytes);
  if (p && bytes && hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return NULL;
  return p;
}

int main(int argc, char **argv) {
  if (argc != 3) return 1;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t hdr[8];
  float fl[4];
  if (fread(hdr, 4, 8, f) != 8 || fread(fl, 4, 4, f) != 4) return 1;
  const int C = hdr[0], E = hdr[1], N = hdr[2], Lir = hdr[3], want_lb = hdr[4], chunk_events = hdr[5], has_amb = hdr[6];
  const float ref_db = fl[0], sr = fl[1], duration = fl[2], amb_ref_db = fl[3];
  al_event_spec *specs = calloc(E, sizeof *specs);
  double *starts = malloc(8 * E), *ends = malloc(8 * E);
  int64_t total_clip = 0;
  for (int e = 0; e < E; ++e) {
    int32_t i4[4];
    float f2[2];
    if (fread(i4, 4, 4, f) != 4 || fread(f2, 4, 2, f) != 2) return 1;
    specs[e].n_samples = i4[0], specs[e].n_emitters = i4[1], specs[e].emitter0 = i4[2], specs[e].is_moving = i4[3];
    specs[e].snr = f2[0], specs[e].ref_db = ref_db, specs[e].gain = 1.0f, specs[e].stft_len = 0;
    specs[e].duration = (double)i4[0] / (double)sr;
    starts[e] = f2[1], ends[e] = (double)f2[1] + (double)i4[0] / (double)sr;
    total_clip += i4[0];
  }
  float *clips = malloc(4 * (size_t)total_clip);
  if (fread(clips, 4, total_clip, f) != (size_t)total_clip) return 1;
  const int64_t ir_pitch = (Lir + 3) / 4 * 4;
  float *irs = calloc((size_t)C * (N > 0 ? N : 1) * ir_pitch, 4);
  for (int64_t r = 0; r < (int64_t)C * N; ++r)
    if (fread(irs + r * ir_pitch, 4, Lir, f) != (size_t)Lir) return 1;
  const int64_t T = (int64_t)nearbyint((double)duration * (double)sr);
  float *noise = NULL;
  if (has_amb) {
    noise = malloc(4 * (size_t)C * T);
    if (fread(noise, 4, (size_t)C * T, f) != (size_t)C * T) return 1;
  }
  fclose(f);
  if (al_abi_version() != AL_ABI_VERSION) return 3;

  /* ---- plan: every table from the library */
  al_plan *plan = NULL;
  AL_CALL(al_plan_create(specs, E, C, Lir, (double)sr, want_lb, 128, 256, 512, &plan));
  al_plan_info info;
  AL_CALL(al_plan_get_info(plan, &info));
  const al_event *ev = al_plan_events(plan);
  const int64_t *audio_off = al_plan_audio_offsets(plan);
  const int B = 1 << info.log2_block, P = info.n_partitions;
  const size_t blk = (size_t)B * 8;
  const int step = chunk_events > 0 ? chunk_events : (E > 0 ? E : 1);
  int64_t h_max = 1, x_max = 1, y_max = 1;
  for (int e0 = 0; e0 < E; e0 += step) {
    al_chunk ch;
    AL_CALL(al_plan_chunk(plan, e0, e0 + step <= E ? step : E - e0, &ch));
    if ((int64_t)ch.n_emitters * C * P > h_max) h_max = (int64_t)ch.n_emitters * C * P;
    if (ch.xspec_blocks > x_max) x_max = ch.xspec_blocks;
    if (ch.yspec_blocks > y_max) y_max = ch.yspec_blocks;
  }
  if (chunk_events <= 0 && al_workspace_bytes(plan) < (int64_t)((h_max + x_max + y_max) * blk)) return 5;   /* the library's own size */

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  float *h_audio = calloc((size_t)info.audio_floats, 4);
  int64_t at = 0;
  for (int e = 0; e < E; ++e) {
    memcpy(h_audio + audio_off[e], clips + at, 4 * (size_t)specs[e].n_samples);
    at += specs[e].n_samples;
  }
  al_batch b;
  memset(&b, 0, sizeof b);
  b.struct_size = (int32_t)sizeof b, b.abi_version = AL_ABI_VERSION;
  b.log2_block = info.log2_block, b.n_capsules = C, b.ir_len = Lir, b.n_partitions = P, b.hop = info.hop;
  b.ir_stride_c = (int64_t)N * ir_pitch, b.ir_stride_n = ir_pitch;
  b.twiddle = dev_alloc(al_twiddle_bytes(info.log2_block));
  b.audio = dev_copy(h_audio, 4 * (size_t)info.audio_floats, stream);
  b.ir = dev_copy(irs, 4 * (size_t)C * (N > 0 ? N : 1) * ir_pitch, stream);
  b.wtab = dev_copy(al_plan_wtab(plan), 4 * (size_t)info.wtab_floats, stream);
  b.events = dev_copy(ev, sizeof(al_event) * (size_t)E, stream);
  b.streams = dev_copy(al_plan_streams(plan), sizeof(al_stream) * (size_t)info.n_streams, stream);
  b.ir_energy = dev_alloc(4 * (size_t)(info.hspec_blocks > 0 ? info.hspec_blocks : 1)), b.emitter_gain = dev_alloc(4 * (size_t)(info.n_emitters + 1));
  b.hspec = dev_alloc((size_t)(h_max + 1) * blk), b.xspec = dev_alloc((size_t)(x_max + 1) * blk), b.yspec = dev_alloc((size_t)y_max * blk);
  b.spatial = dev_alloc(4 * (size_t)info.spatial_floats), b.partials = dev_alloc(16 * (size_t)info.n_partials);
  b.event_stats = dev_alloc(32 * (size_t)(E > 0 ? E : 1)), b.event_scale = dev_alloc(4 * (size_t)(E > 0 ? E : 1));
  if (!b.twiddle || !b.audio || !b.ir || !b.wtab || !b.events || !b.streams || !b.hspec || !b.xspec || !b.yspec || !b.spatial) return 2;
  /* one all-zero block behind each spectra workspace (rows past an odd partition count, out-of-range signal blocks) */
  HIP_OK(hipMemsetAsync((char *)b.hspec + (size_t)h_max * blk, 0, blk, stream));
  HIP_OK(hipMemsetAsync((char *)b.xspec + (size_t)x_max * blk, 0, blk, stream));
  b.hspec_zero_block = (int32_t)h_max, b.xspec_zero_block = (int32_t)x_max;
  /* the dispatch policy is the library's (al_plan_batch_flags): layout flags for the block size, accumulate flags per chunk's
   * event mix -- this host decides nothing */
  int32_t whole = 0;
  AL_CALL(al_plan_batch_flags(plan, NULL, &whole));
  int32_t *parts = malloc(4 * (size_t)(info.n_emitters + 1));
  const int have_parts = al_plan_emitter_parts(plan, parts);
  if (have_parts < 0) return 3;
  if (have_parts) b.emitter_parts = dev_copy(parts, 4 * (size_t)info.n_emitters, stream);
  AL_CALL(al_twiddle_init((float *)b.twiddle, info.log2_block, stream));
  int n_chunks = 0;
  for (int e0 = 0; e0 < E; e0 += step, ++n_chunks) {
    al_chunk ch;
    AL_CALL(al_plan_chunk(plan, e0, e0 + step <= E ? step : E - e0, &ch));
    b.event0 = ch.event0, b.n_events = ch.n_events, b.stream0 = ch.stream0, b.n_streams = ch.n_streams;
    b.emitter0 = ch.emitter0, b.n_emitters = ch.n_emitters, b.xspec_block0 = ch.xspec_block0, b.yspec_block0 = ch.yspec_block0;
    int32_t policy = 0;
    AL_CALL(al_plan_batch_flags(plan, &ch, &policy));
    b.max_blocks = ch.max_blocks, b.max_nj = ch.max_nj, b.flags = policy;
    AL_CALL(al_render_batch(&b, stream));
  }

  /* ---- ambience: per-channel peak normalisation and noise-floor multiplier as one scalar per channel, on the device */
  float *d_noise = NULL, *d_amb_scale = NULL;
  if (has_amb) {
    d_noise = dev_copy(noise, 4 * (size_t)C * T, stream);
    d_amb_scale = dev_alloc(4 * (size_t)C);
    float *rs_part = dev_alloc(4 * (size_t)al_row_stats_partials(C, T));
    double *rs = dev_alloc(32 * (size_t)C);
    AL_CALL(al_row_stats(d_noise, C, T, rs_part, rs, stream));
    AL_CALL(al_ambience_scales(rs, C, T, amb_ref_db, 1, d_amb_scale, stream));
  }

  /* ---- mixdown tables from the library's planner */
  int32_t *lens = malloc(4 * (size_t)E), *rows = malloc(4 * (size_t)E), *idx = malloc(4 * (size_t)E);
  int64_t *src = malloc(8 * (size_t)E);
  for (int e = 0; e < E; ++e) lens[e] = specs[e].n_samples, rows[e] = C, idx[e] = e, src[e] = ev[e].out_off;
  al_mix_plan *mp = NULL;
  AL_CALL(al_plan_mixdown(starts, ends, lens, rows, src, idx, E, (double)duration, (double)sr, C, 4096, &mp));
  al_mix_tables mt;
  AL_CALL(al_mix_plan_get(mp, &mt));
  if (mt.n_samples != T) return 5;
  const size_t ns = (size_t)(mt.n_slots > 0 ? mt.n_slots : 1), nte = (size_t)(mt.n_tile_events > 0 ? mt.n_tile_events : 1);
  al_mix m;
  memset(&m, 0, sizeof m);
  m.struct_size = (int32_t)sizeof m, m.abi_version = AL_ABI_VERSION;
  m.n_capsules = C, m.n_samples = mt.n_samples, m.tile = mt.tile, m.n_tiles = mt.n_tiles, m.accumulate = 0;
  m.tile_ptr = dev_copy(mt.tile_ptr, 4 * (size_t)(mt.n_tiles + 1), stream), m.tile_events = dev_copy(mt.tile_events, 4 * nte, stream);
  m.slot_src = dev_copy(mt.slot_src, 8 * ns, stream), m.slot_len = dev_copy(mt.slot_len, 4 * ns, stream);
  m.slot_start = dev_copy(mt.slot_start, 4 * ns, stream), m.slot_count = dev_copy(mt.slot_count, 4 * ns, stream);
  m.slot_rows = dev_copy(mt.slot_rows, 4 * ns, stream), m.slot_event = dev_copy(mt.slot_event, 4 * ns, stream);
  m.spatial = b.spatial, m.event_scale = b.event_scale;
  float *scene = dev_alloc(4 * (size_t)C * T);
  m.scene = scene, m.ambience = d_noise, m.ambience_scale = d_amb_scale;
  AL_CALL(al_mixdown(&m, stream));

  float *h_scale = malloc(4 * (size_t)E), *h_spatial = malloc(4 * (size_t)info.spatial_floats), *h_scene = malloc(4 * (size_t)C * T);
  double *stats = malloc(32 * (size_t)E);
  HIP_OK(hipMemcpyAsync(h_scale, b.event_scale, 4 * (size_t)E, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(h_spatial, b.spatial, 4 * (size_t)info.spatial_floats, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(h_scene, scene, 4 * (size_t)C * T, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(stats, b.event_stats, 32 * (size_t)E, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  for (int e = 0; e < E; ++e)
    if (stats[4 * e + 2] != 0.0) return 4;   /* librosa.util.valid_audio, synthesize.py:603 */
  f = fopen(argv[2], "wb");
  if (!f) return 1;
  fwrite(h_scale, 4, (size_t)E, f);
  for (int e = 0; e < E; ++e) fwrite(h_spatial + ev[e].out_off, 4, (size_t)C * specs[e].n_samples, f);
  fwrite(h_scene, 4, (size_t)C * T, f);
  fclose(f);
  /* which accumulate kernels the whole plan as ONE batch reaches under the library's policy (the test compares these codes with
   * what audiblelight_amd/engine.py gets from the same plan) */
  int32_t sc = 0, mc = 0;
  al_chunk all;
  AL_CALL(al_plan_chunk(plan, 0, E, &all));
  b.event0 = 0, b.n_events = E, b.stream0 = all.stream0, b.n_streams = all.n_streams, b.emitter0 = all.emitter0, b.n_emitters = all.n_emitters;
  b.max_blocks = all.max_blocks, b.max_nj = all.max_nj, b.flags = whole;
  AL_CALL(al_spectral_mac_variant(&b, &sc, &mc));
  printf("rendered %d events x %d capsules in %d chunk(s), B = %d, P = %d, moving_code = %d, static_code = %d, flags = %d, skipped = %d\n",
         E, C, n_chunks, B, P, mc, sc, whole, mt.n_skipped);
  al_mix_plan_destroy(mp);
  al_plan_destroy(plan);
  return 0;
}
