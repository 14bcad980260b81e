/* TEST INFRASTRUCTURE: a plain C11 program (no Python, no torch, no C++) that renders static events through the C ABI
 * of include/audiblelight_hip.h -- the way a non-Python host would bind the library.  tests/test_gpu_c_caller.py writes
 * the inputs, runs this binary on the MI355X and compares what it writes with the float64 oracle.
 *
 * What it does is what render_event_audio does for one-emitter events (reference synthesize.py:507-608): normalize_irs,
 * time_invariant_convolution, pad_or_truncate, apply_snr, db_to_multiplier -- one al_render_batch call -- then the mixdown
 * of generate_scene_audio_from_events (:314-401) with every event starting at sample 0 -- one al_mixdown call.
 *
 *   render_static <in.bin> <out.bin>
 *   in.bin : int32 C, E, La, Lir, log2_block; float32 ref_db; float32 snr[E]; float32 clips[E][La]; float32 irs[C][E][Lir]
 *   out.bin: float32 event_scale[E]; float32 spatial[E][C][La] (unscaled); float32 scene[C][La]
 *
 * Build (see __graft_entry__.build): gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude render_static.c
 *        -Laudiblelight_amd/csrc -laudiblelight_hip -L/opt/rocm/lib -lamdhip64
 */
#include <hip/hip_runtime_api.h>
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
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, al_last_error()); \
      return 3;                                                                         \
    }                                                                                   \
  } while (0)

static int64_t round_up4(int64_t n) { return (n + 3) / 4 * 4; }

static void *dev_alloc(size_t bytes) {
  void *p = NULL;
  if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return NULL;
  return p;
}

int main(int argc, char **argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
    return 1;
  }
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t hdr[5];
  float ref_db;
  if (fread(hdr, 4, 5, f) != 5 || fread(&ref_db, 4, 1, f) != 1) return 1;
  const int C = hdr[0], E = hdr[1], La = hdr[2], Lir = hdr[3], lb = hdr[4];
  const int B = 1 << lb;
  const int K = (La + B - 1) / B, P = (Lir + B - 1) / B;
  float *snr = malloc(sizeof(float) * E);
  const int64_t clip_pitch = round_up4(La), ir_pitch = round_up4(Lir);
  float *clips = calloc((size_t)E * clip_pitch, 4);
  float *irs = calloc((size_t)C * E * ir_pitch, 4); /* (C, N = E, Lir) with rows padded to a multiple of 4 floats */
  if (fread(snr, 4, E, f) != (size_t)E) return 1;
  for (int e = 0; e < E; ++e)
    if (fread(clips + e * clip_pitch, 4, La, f) != (size_t)La) return 1;
  for (int c = 0; c < C; ++c)
    for (int e = 0; e < E; ++e)
      if (fread(irs + ((int64_t)c * E + e) * ir_pitch, 4, Lir, f) != (size_t)Lir) return 1;
  fclose(f);

  /* event / stream tables: one static stream per event, emitter e = IR column e (synthesize.py:655-675) */
  al_event *ev = calloc(E, sizeof(al_event));
  al_stream *st = calloc(E, sizeof(al_stream));
  const int64_t out_pitch = round_up4((int64_t)C * La);
  for (int e = 0; e < E; ++e) {
    ev[e].audio_off = e * clip_pitch;
    ev[e].out_off = e * out_pitch;
    ev[e].len = ev[e].valid_len = La;
    ev[e].n_blocks = K;
    ev[e].stream0 = e;
    ev[e].n_streams = 1;
    ev[e].yspec_base = e * C * K;
    ev[e].part_base = e * C * K;
    ev[e].snr = snr[e];
    ev[e].ref_db = ref_db;
    st[e].event = e;
    st[e].emitter = e;
    st[e].j_lo = 0;
    st[e].n_j = K;
    st[e].xspec_base = e * K;
    st[e].w_off = -1;
    st[e].gain = 1.0f;
  }

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  const size_t blk = (size_t)B * 8; /* one block spectrum: B complex floats */
  if (al_abi_version() != AL_ABI_VERSION) {   /* the library must implement the header this host was compiled against */
    fprintf(stderr, "libaudiblelight_hip implements ABI %d, this host was compiled against %d\n", al_abi_version(), AL_ABI_VERSION);
    return 3;
  }
  al_batch b;
  memset(&b, 0, sizeof b);
  b.struct_size = (int32_t)sizeof b;
  b.abi_version = AL_ABI_VERSION;
  b.log2_block = lb, b.n_capsules = C, b.n_events = E, b.n_streams = E, b.n_emitters = E, b.ir_len = Lir;
  b.ir_stride_c = (int64_t)E * ir_pitch, b.ir_stride_n = ir_pitch;
  b.n_partitions = P, b.max_blocks = K, b.max_nj = K, b.hop = 128;
  b.xspec_zero_block = b.hspec_zero_block = -1;
  float *twiddle = dev_alloc(al_twiddle_bytes(lb));
  float *d_audio = dev_alloc((size_t)E * clip_pitch * 4), *d_ir = dev_alloc((size_t)C * E * ir_pitch * 4);
  al_event *d_ev = dev_alloc(sizeof(al_event) * E);
  al_stream *d_st = dev_alloc(sizeof(al_stream) * E);
  float *d_wtab = dev_alloc(16);
  float *spatial = dev_alloc((size_t)E * out_pitch * 4), *scale = dev_alloc((size_t)E * 4);
  b.ir_energy = dev_alloc((size_t)E * C * P * 4), b.emitter_gain = dev_alloc((size_t)E * 4);
  b.hspec = dev_alloc((size_t)E * C * P * blk), b.xspec = dev_alloc((size_t)E * K * blk), b.yspec = dev_alloc((size_t)E * C * K * blk);
  b.partials = dev_alloc((size_t)E * C * K * 16), b.event_stats = dev_alloc((size_t)E * 32);
  if (!twiddle || !d_audio || !d_ir || !d_ev || !d_st || !spatial || !scale || !b.hspec || !b.xspec || !b.yspec) return 2;
  b.twiddle = twiddle, b.audio = d_audio, b.ir = d_ir, b.wtab = d_wtab, b.events = d_ev, b.streams = d_st;
  b.spatial = spatial, b.event_scale = scale;
  HIP_OK(hipMemcpyAsync(d_audio, clips, (size_t)E * clip_pitch * 4, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_ir, irs, (size_t)C * E * ir_pitch * 4, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_ev, ev, sizeof(al_event) * E, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_st, st, sizeof(al_stream) * E, hipMemcpyHostToDevice, stream));
  AL_CALL(al_twiddle_init(twiddle, lb, stream));
  AL_CALL(al_render_batch(&b, stream));

  /* mixdown: time tiles of 4096 samples, every event from sample 0 and so in every tile, insertion order
   * (synthesize.py:358-383) */
  const int tile = 4096, n_tiles = (La + tile - 1) / tile;
  int32_t *tile_ptr = malloc(4 * (n_tiles + 1)), *tile_events = malloc(4 * E * n_tiles), *i32 = malloc(4 * E * 5);
  int64_t *slot_src = malloc(8 * E);
  for (int t = 0; t <= n_tiles; ++t) tile_ptr[t] = t * E;
  for (int t = 0; t < n_tiles; ++t)
    for (int e = 0; e < E; ++e) tile_events[t * E + e] = e;
  for (int e = 0; e < E; ++e) {
    slot_src[e] = ev[e].out_off;
    i32[0 * E + e] = La;  /* slot_len */
    i32[1 * E + e] = 0;   /* slot_start */
    i32[2 * E + e] = La;  /* slot_count */
    i32[3 * E + e] = C;   /* slot_rows */
    i32[4 * E + e] = e;   /* slot_event */
  }
  int32_t *d_tp = dev_alloc(4 * (n_tiles + 1)), *d_te = dev_alloc(4 * E * n_tiles), *d_i32 = dev_alloc(4 * E * 5);
  int64_t *d_src = dev_alloc(8 * E);
  float *scene = dev_alloc((size_t)C * La * 4);
  HIP_OK(hipMemcpyAsync(d_tp, tile_ptr, 4 * (n_tiles + 1), hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_te, tile_events, 4 * E * n_tiles, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_i32, i32, 4 * E * 5, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(d_src, slot_src, 8 * E, hipMemcpyHostToDevice, stream));
  al_mix m;
  memset(&m, 0, sizeof m);
  m.struct_size = (int32_t)sizeof m;
  m.abi_version = AL_ABI_VERSION;
  m.n_capsules = C, m.n_samples = La, m.tile = tile, m.n_tiles = n_tiles, m.accumulate = 0;
  m.tile_ptr = d_tp, m.tile_events = d_te, m.slot_src = d_src;
  m.slot_len = d_i32, m.slot_start = d_i32 + E, m.slot_count = d_i32 + 2 * E, m.slot_rows = d_i32 + 3 * E, m.slot_event = d_i32 + 4 * E;
  m.spatial = spatial, m.event_scale = scale, m.scene = scene;
  AL_CALL(al_mixdown(&m, stream));

  float *h_scale = malloc(4 * E), *h_spatial = malloc((size_t)E * out_pitch * 4), *h_scene = malloc((size_t)C * La * 4);
  double stats[4];
  HIP_OK(hipMemcpyAsync(h_scale, scale, 4 * E, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(h_spatial, spatial, (size_t)E * out_pitch * 4, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(h_scene, scene, (size_t)C * La * 4, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(stats, b.event_stats, sizeof stats, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  if (stats[2] != 0.0) {
    fprintf(stderr, "event 0 is not finite\n"); /* librosa.util.valid_audio, synthesize.py:603 */
    return 4;
  }
  f = fopen(argv[2], "wb");
  if (!f) return 1;
  fwrite(h_scale, 4, E, f);
  for (int e = 0; e < E; ++e) fwrite(h_spatial + e * out_pitch, 4, (size_t)C * La, f);
  fwrite(h_scene, 4, (size_t)C * La, f);
  fclose(f);
  printf("rendered %d events x %d capsules, B = %d, K = %d, P = %d, abi %d\n", E, C, B, K, P, al_abi_version());
  return 0;
}
