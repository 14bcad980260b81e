// Host-side planner of the synthesis path behind the C ABI (include/audiblelight_hip.h, "Planning"): the tables al_batch and
// al_mix point to, from a shape-level description of the events.  Pure index arithmetic on the host -- no device call, no
// sample passes through here -- so that a non-Python host does not have to re-derive it (audiblelight_amd/plan.py is a thin
// caller of these entry points; tests/test_host_logic.py holds the two to bit-identical tables on random scenes).
//
// Reference semantics reproduced here (paths relative to the AudibleLight repository):
//   generate_interpolation_matrix   audiblelight/synthesize.py:148-181  (np.round = round-half-even, np.linspace)
//   stft frame count                synthesize.py:123                    (2 * ceil(n / (2 hop)) + 1)
//   n_frames = min(F_a, W.shape[0]) synthesize.py:208-210
//   event slots                     synthesize.py:361-362                (Python round() = round-half-even)
//   pad_or_truncate of the tail     synthesize.py:590
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "../../include/audiblelight_hip.h"

namespace {
thread_local char g_plan_err[256] = "";
int plan_fail(int code, const char *msg) {
  snprintf(g_plan_err, sizeof(g_plan_err), "%s", msg);
  return code;
}
int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// np.linspace(start, stop, n)[i]: i * step + start with step = (stop - start) / (n - 1), the last element exactly `stop`
double linspace_at(double start, double stop, int n, int i) {
  if (n <= 1) return start;
  if (i == n - 1) return stop;
  const double step = (stop - start) / (double)(n - 1);
  return (double)i * step + start;
}
}  // namespace

struct al_plan {
  int32_t log2_block = 0, n_capsules = 0, ir_len = 0, n_emitters = 0, hop = 0, fft_size = 0;
  std::vector<al_event> events;
  std::vector<al_stream> streams;
  std::vector<float> wtab;
  std::vector<int64_t> audio_offsets;
  int64_t audio_floats = 4, spatial_floats = 4;
  int32_t xspec_blocks = 1, yspec_blocks = 1, n_partials = 1;
  int32_t block() const { return 1 << log2_block; }
  int32_t n_partitions() const { return n_emitters ? (ir_len + block() - 1) / block() : 0; }
};

struct al_mix_plan {
  int32_t n_capsules = 0, n_samples = 0, tile = 0, n_tiles = 0, n_slots = 0;
  std::vector<int32_t> tile_ptr, tile_events, slot_len, slot_start, slot_count, slot_rows, slot_event, skipped;
  std::vector<int64_t> slot_src;
};

extern "C" {

const char *al_plan_last_error(void) { return g_plan_err; }
int al_plan_abi_version(void) { return AL_ABI_VERSION; }   // the planner is also linked into a library of its own (no HIP): same check

int32_t al_choose_log2_block(int32_t ir_len, int32_t max_clip) {
  // largest block that keeps two workgroups resident per CU (B = 8192), shrunk for short inputs so that the zero padding of
  // the last block stays small
  const int64_t want = std::max<int64_t>(std::min<int64_t>(ir_len, max_clip), 1);
  int lg = 13;
  while (lg > AL_MIN_LOG2_BLOCK && ((int64_t)1 << lg) > 2 * want) --lg;
  return lg;
}

int32_t al_stft_frame_count(int64_t n_samples, int32_t hop) { return 2 * (int32_t)ceil((double)n_samples / (2.0 * hop)) + 1; }

int32_t al_interpolation_rows(int32_t n_irs, double duration, double sample_rate, int32_t hop) {
  if (n_irs < 1) return 0;
  return (int32_t)nearbyint((linspace_at(0.0, duration, n_irs, n_irs - 1) * sample_rate + hop) / hop);
}

// weights[row * n_irs + l]: linear cross-fade between consecutive IRs (synthesize.py:148-181), float64, rows x n_irs, zeroed here
int al_interpolation_matrix(int32_t n_irs, double duration, double sample_rate, int32_t hop, int32_t rows, double *weights) {
  if (n_irs < 1 || rows < 0 || !weights) return plan_fail(AL_E_BADARG, "bad interpolation matrix arguments");
  std::fill(weights, weights + (int64_t)rows * n_irs, 0.0);
  std::vector<double> first(n_irs);
  for (int l = 0; l < n_irs; ++l) first[l] = nearbyint((linspace_at(0.0, duration, n_irs, l) * sample_rate + hop) / hop);
  for (int l = 0; l + 1 < n_irs; ++l) {
    // idx = arange(first[l], first[l+1] + 1, dtype=int) - 1: numpy takes the LENGTH from the float arguments, the values from int(start)
    const int64_t len = (int64_t)std::max(0.0, ceil(first[l + 1] + 1.0 - first[l]));
    const int64_t start = (int64_t)first[l] - 1;
    for (int64_t i = 0; i < len; ++i) {
      const int64_t row = start + i;
      if (row < 0 || row >= rows) continue;
      const double up = linspace_at(0.0, 1.0, (int)len, (int)i);
      weights[row * n_irs + l] = 1.0 - up;
      weights[row * n_irs + l + 1] = up;
    }
  }
  return AL_OK;
}

int al_plan_create(const al_event_spec *specs, int32_t n_events, int32_t n_capsules, int32_t ir_len, double sample_rate,
                   int32_t log2_block, int32_t hop, int32_t win, int32_t fft_size, al_plan **out) {
  if (!out || n_events < 0 || (n_events > 0 && !specs) || n_capsules <= 0) return plan_fail(AL_E_BADARG, "bad planner arguments");
  *out = nullptr;
  if (win != 2 * hop || fft_size < 2 * win - 1)
    return plan_fail(AL_E_UNSUPPORTED, "the HIP time-variant path needs win_size == 2*hop_size (sin^2 COLA) and fft_size >= 2*win_size-1");
  if (log2_block <= 0) {
    int32_t longest = 1;
    for (int i = 0; i < n_events; ++i) longest = std::max(longest, specs[i].n_samples);
    log2_block = al_choose_log2_block(ir_len, longest);
    // B = 16384 (four 4096-point transforms per window, csrc/al_quad16.h) in the two situations it was measured to pay:
    const int64_t p13 = ((int64_t)ir_len + 8191) / 8192;
    bool any_moving = false;
    for (int i = 0; i < n_events; ++i) any_moving = any_moving || specs[i].n_emitters > 1;
    if (log2_block == 13 && !any_moving && p13 >= 17 && p13 <= 24) {
      // (1) big batches of static events with 17..24 partitions of 8192: the accumulate is back in the capsule loop's register tile
      // (<= 12 partitions: 5.2 instead of 4.1 TB/s on cfg5), which pays for the slower transforms from about 100 000 (capsule, block)
      // rows on (profiles/r04u_quad16_ir_sweep_*.txt: -4..-11 % per batch from there on, -2..+5 % at half that size, slower
      // everywhere outside 17..24)
      int64_t blocks = 0;
      for (int i = 0; i < n_events; ++i) blocks += ((int64_t)specs[i].n_samples + 8191) / 8192;
      if (blocks * n_capsules >= 100000) log2_block = 14;
    } else if (log2_block == 13 && any_moving) {
      // (2) moving events that fall off the sliding-window accumulate at B = 8192 (a cross-fade window of more than AL_SPARSE_MAX_NJ
      // blocks, or more than AL_SPARSE_MAX_PARTITIONS partitions: the tile kernel sums over streams instead) but not at B = 16384:
      // -5..-31 % per batch (profiles/r04z_quad16_moving_sweep*.txt).  Where both sizes are eligible the smaller block wins
      // (shorter zero-padded windows: cfg3's shape with 2.3-4 s IRs is 7-14 % slower at 16384), so nothing changes there.
      auto all_sliding = [&](int32_t lb, al_plan **made) -> bool {
        if ((((int64_t)ir_len + ((int64_t)1 << lb) - 1) >> lb) > AL_SPARSE_MAX_PARTITIONS) return false;
        if (al_plan_create(specs, n_events, n_capsules, ir_len, sample_rate, lb, hop, win, fft_size, made) != AL_OK) return false;
        for (int i = 0; i < n_events; ++i)
          if (specs[i].n_emitters > 1 && (*made)->events[i].reserved != 1) return false;
        return true;
      };
      al_plan *p13_plan = nullptr, *p14_plan = nullptr;
      const bool ok13 = all_sliding(13, &p13_plan);
      const bool ok14 = !ok13 && all_sliding(14, &p14_plan);
      al_plan *keep = ok14 ? p14_plan : p13_plan;
      if (keep != p13_plan) al_plan_destroy(p13_plan);
      if (keep != p14_plan) al_plan_destroy(p14_plan);
      if (keep) {          // one of the two trial plans IS the result
        *out = keep;
        return AL_OK;
      }
      // neither could be made (the error of the explicit call below is the caller's)
    }
  }
  if (log2_block < AL_MIN_LOG2_BLOCK || log2_block > AL_MAX_LOG2_BLOCK) return plan_fail(AL_E_UNSUPPORTED, "log2_block must be in [10, 14]");
  al_plan *p = new (std::nothrow) al_plan;
  if (!p) return plan_fail(AL_E_BADARG, "out of memory");
  p->log2_block = log2_block; p->n_capsules = n_capsules; p->ir_len = ir_len; p->hop = hop; p->fft_size = fft_size;
  const int64_t B = (int64_t)1 << log2_block;
  p->events.assign(n_events, al_event{});
  p->audio_offsets.assign(n_events, 0);
  int64_t audio_off = 0, out_off = 0, x_blocks = 0, y_blocks = 0, parts = 0, w_floats = 0;
  int32_t n_emit_used = 0;
  std::vector<double> w;
  for (int i = 0; i < n_events; ++i) {
    const al_event_spec &sp = specs[i];
    const int64_t La = sp.n_samples;
    if (La <= 0) { delete p; return plan_fail(AL_E_BADARG, "event clip must have at least one sample"); }
    const int64_t K = (La + B - 1) / B;
    al_event &ev = p->events[i];
    p->audio_offsets[i] = audio_off;
    ev.audio_off = audio_off; ev.out_off = out_off; ev.len = (int32_t)La; ev.n_blocks = (int32_t)K;
    ev.snr = sp.snr; ev.ref_db = sp.ref_db;
    ev.stream0 = (int32_t)p->streams.size(); ev.yspec_base = (int32_t)y_blocks; ev.part_base = (int32_t)parts;
    ev.reserved = 0;
    int64_t valid = La;
    if (sp.n_emitters == 0) {
      ev.n_streams = 0;
      p->streams.push_back(al_stream{i, 0, 0, 0, 0, -1, 0, sp.gain});   // carries the gain only
    } else if (sp.n_emitters == 1) {
      if (sp.is_moving) { delete p; return plan_fail(AL_E_BADARG, "Moving Event has only one emitter!"); }
      ev.n_streams = 1;
      p->streams.push_back(al_stream{i, sp.emitter0, 0, (int32_t)K, (int32_t)x_blocks, -1, 0, sp.gain});
      x_blocks += K;
    } else {
      if (!sp.is_moving) { delete p; return plan_fail(AL_E_BADARG, "Expected a moving event!"); }
      if (!(sp.duration > 0.0)) { delete p; return plan_fail(AL_E_BADARG, "moving events need Event.duration"); }
      const int32_t N = sp.n_emitters;
      const int32_t rows = al_interpolation_rows(N, sp.duration, sample_rate, hop);
      w.resize((size_t)std::max(rows, 0) * N);
      al_interpolation_matrix(N, sp.duration, sample_rate, hop, rows, w.data());
      const int32_t n_frames = std::min(al_stft_frame_count(sp.stft_len > 0 ? sp.stft_len : La, hop), rows);
      valid = std::min<int64_t>(La, std::max<int64_t>((int64_t)n_frames * hop - win, 0));
      ev.n_streams = N;
      const size_t first_stream = p->streams.size();
      for (int l = 0; l < N; ++l) {
        int32_t lo = -1, hi = -1;
        for (int32_t r = 0; r < n_frames; ++r)
          if (w[(size_t)r * N + l] != 0.0) { if (lo < 0) lo = r; hi = r; }
        int32_t j_lo = 0, n_j = 0;
        if (lo >= 0) {
          const int64_t t_lo = std::max<int64_t>((int64_t)hop * (lo - 1), 0), t_hi = std::min<int64_t>((int64_t)hop * (hi + 1), La);
          j_lo = (int32_t)(t_lo / B);
          const int64_t j_hi = std::min<int64_t>(K - 1, (t_hi + B - 1) / B);
          n_j = t_hi > t_lo ? (int32_t)std::max<int64_t>(j_hi - j_lo + 1, 0) : 0;
        }
        p->streams.push_back(al_stream{i, sp.emitter0 + l, j_lo, n_j, (int32_t)x_blocks, (int32_t)w_floats, n_frames, sp.gain * (float)fft_size});
        x_blocks += n_j;
        for (int32_t r = 0; r < n_frames; ++r) p->wtab.push_back((float)w[(size_t)r * N + l]);
        w_floats += n_frames;
      }
      bool starts_ok = true;
      int32_t longest = 0;
      for (size_t s = first_stream; s < p->streams.size(); ++s) {
        longest = std::max(longest, p->streams[s].n_j);
        if (s + 1 < p->streams.size() && p->streams[s].n_j > 0 && p->streams[s + 1].n_j > 0 && p->streams[s].j_lo > p->streams[s + 1].j_lo)
          starts_ok = false;
      }
      if (starts_ok && longest <= AL_SPARSE_MAX_NJ) ev.reserved = 1;   // sliding-window accumulate
    }
    ev.valid_len = (int32_t)valid;
    n_emit_used = std::max(n_emit_used, sp.emitter0 + sp.n_emitters);
    y_blocks += sp.n_emitters ? (int64_t)n_capsules * K : 0;
    parts += (int64_t)n_capsules * K;
    audio_off += round_up(La, 4);
    out_off += round_up((int64_t)n_capsules * La, 4);
  }
  if (x_blocks > 0x7fffffff || y_blocks > 0x7fffffff || parts > 0x7fffffff) { delete p; return plan_fail(AL_E_UNSUPPORTED, "batch too large for 32-bit block indices"); }
  if (p->wtab.empty()) p->wtab.push_back(0.f);
  p->n_emitters = n_emit_used;
  p->audio_floats = std::max<int64_t>(audio_off, 4); p->spatial_floats = std::max<int64_t>(out_off, 4);
  p->xspec_blocks = (int32_t)std::max<int64_t>(x_blocks, 1); p->yspec_blocks = (int32_t)std::max<int64_t>(y_blocks, 1);
  p->n_partials = (int32_t)std::max<int64_t>(parts, 1);
  *out = p;
  return AL_OK;
}

void al_plan_destroy(al_plan *p) { delete p; }

int al_plan_get_info(const al_plan *p, al_plan_info *info) {
  if (!p || !info) return plan_fail(AL_E_BADARG, "null plan");
  memset(info, 0, sizeof(*info));
  info->log2_block = p->log2_block; info->n_capsules = p->n_capsules; info->ir_len = p->ir_len;
  info->n_events = (int32_t)p->events.size(); info->n_streams = (int32_t)p->streams.size(); info->n_emitters = p->n_emitters;
  info->n_partitions = p->n_partitions(); info->hop = p->hop; info->fft_size = p->fft_size;
  for (const al_event &e : p->events) info->max_blocks = std::max(info->max_blocks, e.n_blocks);
  for (const al_stream &s : p->streams) {
    info->max_nj = std::max(info->max_nj, s.n_j);
    if (p->events[s.event].reserved == 1) info->max_nj_sliding = std::max(info->max_nj_sliding, s.n_j);
  }
  info->audio_floats = p->audio_floats; info->spatial_floats = p->spatial_floats; info->wtab_floats = (int64_t)p->wtab.size();
  info->xspec_blocks = p->xspec_blocks; info->yspec_blocks = p->yspec_blocks;
  info->hspec_blocks = (int64_t)p->n_emitters * p->n_capsules * p->n_partitions();
  info->n_partials = p->n_partials;
  return AL_OK;
}

const al_event *al_plan_events(const al_plan *p) { return p ? p->events.data() : nullptr; }
const al_stream *al_plan_streams(const al_plan *p) { return p ? p->streams.data() : nullptr; }
const float *al_plan_wtab(const al_plan *p) { return p ? p->wtab.data() : nullptr; }
const int64_t *al_plan_audio_offsets(const al_plan *p) { return p ? p->audio_offsets.data() : nullptr; }

int64_t al_workspace_bytes(const al_plan *p) {
  if (!p) return -1;
  const int64_t b8 = (int64_t)p->block() * 8, h = (int64_t)p->n_emitters * p->n_capsules * p->n_partitions();
  return (h + p->xspec_blocks + p->yspec_blocks) * b8 + h * 4 + (int64_t)p->n_emitters * 4 + (int64_t)p->n_partials * 16;
}

// The al_batch fields of a run of consecutive events ("chunk") over the plan's global tables: a long scene is rendered as several
// chunks that reuse ONE spectra workspace (al_batch.event0 / stream0 / emitter0 / *_block0).
int al_plan_chunk(const al_plan *p, int32_t event0, int32_t n_events, al_chunk *out) {
  if (!p || !out || event0 < 0 || n_events < 0 || event0 > (int32_t)p->events.size() || n_events > (int32_t)p->events.size() - event0) return plan_fail(AL_E_BADARG, "bad chunk range");
  memset(out, 0, sizeof(*out));
  out->event0 = event0; out->n_events = n_events;
  if (n_events == 0) return AL_OK;
  const al_event &e_first = p->events[event0], &e_last = p->events[event0 + n_events - 1];
  const int32_t s0 = e_first.stream0, s1 = e_last.stream0 + std::max(e_last.n_streams, 1);
  out->stream0 = s0; out->n_streams = s1 - s0;
  int64_t em0 = INT64_MAX, em1 = 0, x0 = INT64_MAX, x1 = 0, y0 = INT64_MAX, y1 = 0;
  for (int32_t i = event0; i < event0 + n_events; ++i) {
    const al_event &e = p->events[i];
    out->max_blocks = std::max(out->max_blocks, e.n_blocks);
    if (e.n_streams > 0) {
      y0 = std::min<int64_t>(y0, e.yspec_base);
      y1 = std::max<int64_t>(y1, (int64_t)e.yspec_base + (int64_t)p->n_capsules * e.n_blocks);
      for (int32_t s = e.stream0; s < e.stream0 + e.n_streams; ++s) {   // streams of convolved events (pseudo-streams of tiled events carry no spectra)
        const al_stream &st = p->streams[s];
        em0 = std::min<int64_t>(em0, st.emitter); em1 = std::max<int64_t>(em1, (int64_t)st.emitter + 1);
        x0 = std::min<int64_t>(x0, st.xspec_base); x1 = std::max<int64_t>(x1, (int64_t)st.xspec_base + st.n_j);
      }
    }
  }
  for (int32_t s = s0; s < s1; ++s) out->max_nj = std::max(out->max_nj, p->streams[s].n_j);
  const bool conv = em1 > 0 || x1 > 0 || y1 > 0;
  out->emitter0 = conv && em0 != INT64_MAX ? (int32_t)em0 : 0; out->n_emitters = conv && em0 != INT64_MAX ? (int32_t)(em1 - em0) : 0;
  out->xspec_block0 = x0 != INT64_MAX ? (int32_t)x0 : 0; out->xspec_blocks = x0 != INT64_MAX ? (int32_t)(x1 - x0) : 0;
  out->yspec_block0 = y0 != INT64_MAX ? (int32_t)y0 : 0; out->yspec_blocks = y0 != INT64_MAX ? (int32_t)(y1 - y0) : 0;
  return AL_OK;
}

// al_batch.flags for a chunk of the plan: the dispatch policy the benchmarked path runs with, so that every host -- C, Python,
// anything else -- reaches the same kernels from the same plan (include/audiblelight_hip.h, "Dispatch policy").  Measured rules:
//   layout      B = 8192: split (every window as two 4096-point transforms, profiles/r02_split.txt: -5 % per scene);
//               B = 16384: split + quad tiles (four 4096-point transforms, csrc/al_quad16.h: cfg5 13.9 instead of 17.0 ms);
//               below 8192 the one-transform kernels (split is slower at B = 4096).
//   accumulate  one-emitter events of batches with at most 21 partitions go through the capsule loop (k_spectral_mac_static*:
//               -7 % on cfg2's accumulate, -14 % on cfg4's, -20..30 % at 13..17 partitions, profiles/r02_mac.txt, r03_p24_ab.txt);
//               a chunk WITHOUT multi-emitter events does not launch the other accumulate kernels at all.
int al_plan_batch_flags(const al_plan *p, const al_chunk *chunk, int32_t *flags) {
  if (!p || !flags) return plan_fail(AL_E_BADARG, "null plan");
  int32_t e0 = 0, n = (int32_t)p->events.size();
  if (chunk) {
    e0 = chunk->event0; n = chunk->n_events;
    if (e0 < 0 || n < 0 || e0 > (int32_t)p->events.size() || n > (int32_t)p->events.size() - e0) return plan_fail(AL_E_BADARG, "bad chunk range");
  }
  int32_t f = 0;
  if (p->log2_block == 13) f |= AL_FLAG_SPLIT_SPECTRA;
  else if (p->log2_block == 14) f |= AL_FLAG_SPLIT_SPECTRA | AL_FLAG_QUAD_SPECTRA;
  if (p->n_partitions() <= AL_STATIC_MAC_MAX_PARTITIONS) {
    bool any_static = false, any_multi = false;
    for (int32_t i = e0; i < e0 + n; ++i) {
      any_static |= p->events[i].n_streams == 1;
      any_multi |= p->events[i].n_streams > 1;
    }
    if (any_static) f |= AL_FLAG_STATIC_MAC | (any_multi ? 0 : AL_FLAG_ONLY_STATIC);
  }
  *flags = f;
  return AL_OK;
}

// al_batch.emitter_parts for the whole plan.  Returns 1 and fills out[n_emitters] when the batch needs the table (some IR has
// partitions no kept block hears), 0 when every IR needs all its partitions (out untouched).
int al_plan_emitter_parts(const al_plan *p, int32_t *out) {
  if (!p || !out) return plan_fail(AL_E_BADARG, "null plan");
  const int32_t P = p->n_partitions(), N = p->n_emitters;
  bool any_sliding = false;
  for (const al_event &e : p->events) any_sliding |= e.reserved == 1;
  if (P < 1 || N < 1 || p->streams.empty() || !any_sliding) return 0;
  std::vector<int32_t> need(N, 0);
  std::vector<char> used(N, 0);
  const bool trim = P > 1 && P <= AL_SPARSE_MAX_PARTITIONS;
  for (const al_stream &s : p->streams) {
    const al_event &e = p->events[s.event];
    if (e.n_streams <= 0 || s.emitter < 0 || s.emitter >= N) continue;
    used[s.emitter] = 1;
    int32_t want = P;
    if (e.reserved == 1) {
      const int32_t reach = std::min(std::max(e.n_blocks - s.j_lo, 0), P);
      want = trim ? (s.n_j > 0 ? reach : 0) : P;
    }
    need[s.emitter] = std::max(need[s.emitter], want);
  }
  bool any = false;
  for (int n = 0; n < N; ++n) {
    if (!used[n]) need[n] = P;
    any |= need[n] < P;
  }
  if (!any) return 0;
  memcpy(out, need.data(), sizeof(int32_t) * N);
  return 1;
}

// ---- mixdown (generate_scene_audio_from_events, synthesize.py:314-401): slots and per-tile event lists
int al_plan_mixdown(const double *starts, const double *ends, const int32_t *lens, const int32_t *rows, const int64_t *src_offsets,
                    const int32_t *event_index, int32_t n, double duration, double sample_rate, int32_t n_capsules, int32_t tile,
                    al_mix_plan **out) {
  if (!out || n < 0 || tile <= 0 || (n > 0 && (!starts || !ends || !lens || !rows || !src_offsets || !event_index)))
    return plan_fail(AL_E_BADARG, "bad mixdown planner arguments");
  al_mix_plan *m = new (std::nothrow) al_mix_plan;
  if (!m) return plan_fail(AL_E_BADARG, "out of memory");
  const int64_t n_scene = (int64_t)nearbyint(duration * sample_rate);   // Python round(): half to even
  m->n_capsules = n_capsules; m->n_samples = (int32_t)n_scene; m->tile = tile; m->n_tiles = (int32_t)((n_scene + tile - 1) / tile);
  for (int i = 0; i < n; ++i) {
    const int64_t a = std::max<int64_t>(0, (int64_t)nearbyint(starts[i] * sample_rate));
    const int64_t b = std::min<int64_t>((int64_t)nearbyint(ends[i] * sample_rate), n_scene);
    if (b <= a) { m->skipped.push_back(i); continue; }
    m->slot_start.push_back((int32_t)a);
    m->slot_count.push_back((int32_t)std::min<int64_t>(b - a, lens[i]));
    m->slot_src.push_back(src_offsets[i]); m->slot_len.push_back(lens[i]); m->slot_rows.push_back(rows[i]); m->slot_event.push_back(event_index[i]);
  }
  m->n_slots = (int32_t)m->slot_start.size();
  std::vector<std::vector<int32_t>> lists((size_t)std::max(m->n_tiles, 0));
  for (int32_t s = 0; s < m->n_slots; ++s) {
    const int64_t a = m->slot_start[s], cnt = m->slot_count[s];
    const int64_t t1 = std::min<int64_t>((a + cnt - 1) / tile, m->n_tiles - 1);
    for (int64_t t = a / tile; t <= t1; ++t) lists[(size_t)t].push_back(s);
  }
  m->tile_ptr.assign((size_t)m->n_tiles + 1, 0);
  for (int32_t t = 0; t < m->n_tiles; ++t) {
    m->tile_ptr[t + 1] = m->tile_ptr[t] + (int32_t)lists[t].size();
    m->tile_events.insert(m->tile_events.end(), lists[t].begin(), lists[t].end());
  }
  if (m->tile_events.empty()) m->tile_events.push_back(0);
  if (m->n_slots == 0) {   // one dummy entry so that no table is empty (never read: tile_ptr is all zeros)
    m->slot_start.push_back(0); m->slot_count.push_back(0); m->slot_src.push_back(0); m->slot_len.push_back(0);
    m->slot_rows.push_back(0); m->slot_event.push_back(0);
  }
  *out = m;
  return AL_OK;
}

void al_mix_plan_destroy(al_mix_plan *m) { delete m; }

int al_mix_plan_get(const al_mix_plan *m, al_mix_tables *t) {
  if (!m || !t) return plan_fail(AL_E_BADARG, "null mixdown plan");
  t->n_capsules = m->n_capsules; t->n_samples = m->n_samples; t->tile = m->tile; t->n_tiles = m->n_tiles;
  t->n_slots = m->n_slots; t->n_tile_events = m->tile_ptr.empty() ? 0 : m->tile_ptr.back(); t->n_skipped = (int32_t)m->skipped.size();
  t->tile_ptr = m->tile_ptr.data(); t->tile_events = m->tile_events.data(); t->slot_src = m->slot_src.data();
  t->slot_len = m->slot_len.data(); t->slot_start = m->slot_start.data(); t->slot_count = m->slot_count.data();
  t->slot_rows = m->slot_rows.data(); t->slot_event = m->slot_event.data(); t->skipped = m->skipped.data();
  return AL_OK;
}

}  // extern "C"
