/*
 * oracle/conv_direct.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Direct-form (time-domain, float64 accumulate) restatement of the static-event convolution of
 * AudibleLight: time_invariant_convolution (audiblelight/synthesize.py:71-106, a full linear
 * convolution of a mono clip with every capsule IR) followed by pad_or_truncate_audio to the
 * clip length (audiblelight/utils.py:667-688, called at synthesize.py:590).  It shares no code
 * and no algorithm (no FFT) with either the numpy oracle or the HIP path, so it is an independent
 * witness for small cases.  Parity status: pinned -- tests/test_oracle_golden.py::test_c_direct_form_witness
 * holds it to the reference goldens G1 / G1b (full convolution + truncation) and G8 (mixdown) through oracle/conv_direct.py.
 *
 *   y[c][t] = sum_m ir[c][m] * audio[t - m],   0 <= t < n_out
 */
#include <stddef.h>

void al_oracle_conv_direct(const float *audio, long n_audio, const double *ir, long n_caps, long n_ir, double *out,
                           long n_out) {
  for (long c = 0; c < n_caps; ++c) {
    const double *h = ir + c * n_ir;
    double *y = out + c * n_out;
    for (long t = 0; t < n_out; ++t) {
      long m_lo = t - (n_audio - 1);
      if (m_lo < 0) m_lo = 0;
      long m_hi = t < n_ir - 1 ? t : n_ir - 1;
      double acc = 0.0;
      for (long m = m_lo; m <= m_hi; ++m) acc += h[m] * (double)audio[t - m];
      y[t] = acc;
    }
  }
}

/* Segmented additive mixdown into a float32 scene buffer with per-event float32 rounding, as
 * numpy's in-place `scene[:, a:b] += x` does (synthesize.py:373-378). */
void al_oracle_mix_add(float *scene, long n_caps, long n_scene, const double *x, long x_len, long start, long count) {
  for (long c = 0; c < n_caps; ++c)
    for (long i = 0; i < count; ++i) {
      float *dst = scene + c * n_scene + start + i;
      *dst = (float)((double)*dst + x[c * x_len + i]);
    }
}
