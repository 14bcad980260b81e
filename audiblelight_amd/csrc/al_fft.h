// In-LDS Stockham FFT for gfx950 (CDNA4): the per-block transform of the partitioned
// overlap-save convolution (DESIGN.md "Kernels").
//
// Geometry: a transform of M = 2^LOG2M complex points is done by one workgroup of T = M/16
// threads (M = 8192 -> 512 threads = 8 wave64).  Every thread owns 16 complex values per pass,
// always the LDS slots  tid + T*m, m = 0..15  on the read side (so reads are 512 contiguous
// bytes per wave: conflict-free ds_read_b64), and scatters its butterfly outputs in Stockham
// auto-sort order on the write side.  The LDS image is padded by one float2 per 16 so the
// stride-16 scatter of the first pass and the 16-wide runs of the second land on distinct
// banks (ds_write_b64 is serviced 16 lanes at a time over 32 dword banks).
// Passes are radix 16 (4x4 in registers) with one final radix 2/4/8 pass when LOG2M is not a
// multiple of 4.  A pass is: read 16 -> twiddle -> butterflies -> barrier -> write 16 -> barrier,
// in place (all reads of a pass complete before any write).
//
// Twiddles: one table tw[k] = exp(-i*pi*k/M), k < M (the 2M-th roots over half a turn, made
// in float64 by al_twiddle_init).  A pass loads one base factor per butterfly and derives the
// other powers by at most four complex products.
#pragma once
#include <hip/hip_runtime.h>

namespace al {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// a += b * c
__device__ __forceinline__ void cfma(float2 &a, float2 b, float2 c) {
  a.x = fmaf(b.x, c.x, a.x);
  a.x = fmaf(-b.y, c.y, a.x);
  a.y = fmaf(b.x, c.y, a.y);
  a.y = fmaf(b.y, c.x, a.y);
}
// multiply by DIR*i  (DIR = -1 forward transform, +1 inverse)
template <int DIR>
__device__ __forceinline__ float2 rot90(float2 a) {
  return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

constexpr int lds_pad(int i) { return i + (i >> 4); }
constexpr int fft_threads(int log2m) { return (1 << log2m) / 16; }
constexpr int fft_lds_elems(int log2m) { return lds_pad(1 << log2m); }
constexpr int fft_npasses(int log2m) { return (log2m + 3) / 4; }
constexpr int fft_radix(int log2m, int pass) { return (log2m - 4 * pass) >= 4 ? 16 : (1 << (log2m - 4 * pass)); }

template <int DIR>
__device__ __forceinline__ void bfly2(float2 &a, float2 &b) {
  float2 t = csub(a, b);
  a = cadd(a, b);
  b = t;
}

template <int DIR>
__device__ __forceinline__ void bfly4(float2 &a0, float2 &a1, float2 &a2, float2 &a3) {
  float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = rot90<DIR>(csub(a1, a3));
  a0 = cadd(t0, t2);
  a1 = cadd(t1, t3);
  a2 = csub(t0, t2);
  a3 = csub(t1, t3);
}

// x[0..7] natural order in, X[0..7] natural order out.
template <int DIR>
__device__ __forceinline__ void bfly8(float2 (&x)[8]) {
  constexpr float H = 0.70710678118654752440f;
  bfly4<DIR>(x[0], x[2], x[4], x[6]);  // even samples  -> A0[k1] at x[0],x[2],x[4],x[6]
  bfly4<DIR>(x[1], x[3], x[5], x[7]);  // odd samples   -> A1[k1] at x[1],x[3],x[5],x[7]
  // A1[k1] *= w8^(k1), w8 = exp(DIR * 2*pi*i/8)
  float2 b1 = x[3], b3 = x[7];
  x[3] = make_float2(H * (b1.x - DIR * b1.y), H * (b1.y + DIR * b1.x));
  x[5] = rot90<DIR>(x[5]);
  x[7] = make_float2(H * (-b3.x - DIR * b3.y), H * (-b3.y + DIR * b3.x));
  float2 a0 = x[0], a1 = x[2], a2 = x[4], a3 = x[6];
  float2 c0 = x[1], c1 = x[3], c2 = x[5], c3 = x[7];
  x[0] = cadd(a0, c0); x[4] = csub(a0, c0);
  x[1] = cadd(a1, c1); x[5] = csub(a1, c1);
  x[2] = cadd(a2, c2); x[6] = csub(a2, c2);
  x[3] = cadd(a3, c3); x[7] = csub(a3, c3);
}

// x[0..15] natural order in; X[k] ends up at x[4*(k%4) + k/4].
template <int DIR>
__device__ __forceinline__ void bfly16(float2 (&x)[16]) {
  constexpr float C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f, H = 0.70710678118654752440f;
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) bfly4<DIR>(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);
  // now x[4*k1 + n2] = A[n2][k1]; multiply by w16^(n2*k1), w16 = exp(DIR*2*pi*i/16)
  const float2 w1 = make_float2(C1, DIR * S1), w2 = make_float2(H, DIR * H), w3 = make_float2(S1, DIR * C1);
  const float2 w6 = make_float2(-H, DIR * H), w9 = make_float2(-C1, -DIR * S1);
  x[5] = cmul(x[5], w1);   // k1=1,n2=1
  x[6] = cmul(x[6], w2);   // k1=1,n2=2
  x[7] = cmul(x[7], w3);   // k1=1,n2=3
  x[9] = cmul(x[9], w2);   // k1=2,n2=1
  x[10] = rot90<DIR>(x[10]);  // k1=2,n2=2 : w4
  x[11] = cmul(x[11], w6);  // k1=2,n2=3
  x[13] = cmul(x[13], w3);  // k1=3,n2=1
  x[14] = cmul(x[14], w6);  // k1=3,n2=2
  x[15] = cmul(x[15], w9);  // k1=3,n2=3
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) bfly4<DIR>(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
}

// Twiddle powers w^1..w^(R-1) from two exact table entries (w^1, w^4): at most three
// chained products per power.
template <int R>
__device__ __forceinline__ void twiddle_powers(float2 (&w)[16], float2 a, float2 b) {
  w[1] = a;
  if (R > 2) {
    w[2] = cmul(a, a);
    w[3] = cmul(w[2], a);
  }
  if (R > 4) {
    w[4] = b;
    w[5] = cmul(b, a);
    w[6] = cmul(b, w[2]);
    w[7] = cmul(b, w[3]);
  }
  if (R > 8) {
    w[8] = cmul(b, b);
    w[9] = cmul(w[8], a);
    w[10] = cmul(w[8], w[2]);
    w[11] = cmul(w[8], w[3]);
    w[12] = cmul(w[8], b);
    w[13] = cmul(w[12], a);
    w[14] = cmul(w[12], w[2]);
    w[15] = cmul(w[12], w[3]);
  }
}

// Twiddle base factors of one pass (one (w^1, w^4) pair per butterfly), loaded a pass ahead so
// the L2 round trip overlaps the previous pass's butterflies and barriers.
template <int LOG2M, int PASS>
struct PassTwiddles {
  static constexpr int NB = 16 / fft_radix(LOG2M, PASS);
  float2 a[NB], c[NB];
};

// TWS = stride of this transform's factors inside the table (1: the table was made for this size, 2: the
// table belongs to a transform twice as long, used by the split kernels).
template <int LOG2M, int DIR, int PASS, int TWS = 1>
__device__ __forceinline__ void load_pass_twiddles(PassTwiddles<LOG2M, PASS> &t, const float2 *__restrict__ tw, int tid) {
  constexpr int M = 1 << LOG2M, T = M / 16;
  constexpr int R = fft_radix(LOG2M, PASS);
  constexpr int NS = 1 << (4 * PASS);
  constexpr int NB = 16 / R;
  constexpr int STEP = 2 * M / (NS * R);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int k = (tid + T * b) & (NS - 1);
    t.a[b] = tw[TWS * k * STEP];
    t.c[b] = (R > 4) ? tw[TWS * 4 * k * STEP] : t.a[b];
    if (DIR > 0) {
      t.a[b].y = -t.a[b].y;
      t.c[b].y = -t.c[b].y;
    }
  }
}

// One Stockham pass.  On entry (PASS == 0) v[m] holds in[tid + T*m]; later passes read those
// slots from LDS themselves.  Outputs are scattered to LDS; the caller must have a barrier
// between the last read of the LDS image and this call's writes when PASS == 0.
//   butterfly j = tid + T*b reads in[j + r*M/R]      = v[b + r*NB]
//   k = j mod NS;  factor exp(DIR*2*pi*i*r*k/(NS*R))
//   out[(j - k)*R + k + q*NS] = X[q]
template <int LOG2M, int DIR, int PASS, bool TO_REGS = false>
__device__ __forceinline__ void fft_pass(float2 (&v)[16], float2 *s, const PassTwiddles<LOG2M, PASS> &t, int tid) {
  constexpr int M = 1 << LOG2M, T = M / 16;
  constexpr int R = fft_radix(LOG2M, PASS);
  constexpr int NS = 1 << (4 * PASS);
  constexpr int NB = 16 / R;
  static_assert(T % 16 == 0, "lds_pad(a + c) == lds_pad(a) + lds_pad(c) needs c % 16 == 0");
  if (PASS > 0) {
    const float2 *rd = s + lds_pad(tid);  // slot m lives at rd[lds_pad(T*m)]: one base register, immediate offsets
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = rd[lds_pad(T * m)];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float2 w[16];
      twiddle_powers<R>(w, t.a[b], t.c[b]);
#pragma unroll
      for (int r = 1; r < R; ++r) v[b + r * NB] = cmul(v[b + r * NB], w[r]);
    }
    // all LDS reads of this pass are done once every thread is here
    __syncthreads();
  }
  if (TO_REGS) {
    // Final pass (NS * R == M): output q of butterfly b is element tid + T*(b + q*NB), i.e. the thread's
    // own slot b + q*NB, so the natural-order result can stay in registers: v[m] = X[tid + T*m].
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (R == 16) {
        bfly16<DIR>(v);
        float2 o[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) o[q] = v[4 * (q & 3) + (q >> 2)];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = o[q];
      } else if (R == 8) {
        float2 x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = v[b + r * NB];
        bfly8<DIR>(x);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[b + q * NB] = x[q];
      } else if (R == 4) {
        bfly4<DIR>(v[b], v[b + NB], v[b + 2 * NB], v[b + 3 * NB]);
      } else {
        bfly2<DIR>(v[b], v[b + NB]);
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int j = tid + T * b;
    const int k = j & (NS - 1);
    const int base = (j - k) * R + k;
    // output q goes to base + q*NS.  NS == 1 (first pass, R == 16): lds_pad(16 j + q) = 17 j + q.
    // NS >= 16: q*NS is a multiple of 16, so lds_pad splits into a per-thread base and a constant offset.
    float2 *wr = s + (NS == 1 ? 17 * j : lds_pad(base));
    constexpr int QS = (NS == 1) ? 1 : lds_pad(NS);  // NS >= 16: lds_pad(q*NS) == q * lds_pad(NS)
    if (R == 16) {
      bfly16<DIR>(v);
#pragma unroll
      for (int q = 0; q < 16; ++q) wr[q * QS] = v[4 * (q & 3) + (q >> 2)];
    } else if (R == 8) {
      float2 x[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) x[r] = v[b + r * NB];
      bfly8<DIR>(x);
#pragma unroll
      for (int q = 0; q < 8; ++q) wr[q * QS] = x[q];
    } else if (R == 4) {
      float2 x0 = v[b], x1 = v[b + NB], x2 = v[b + 2 * NB], x3 = v[b + 3 * NB];
      bfly4<DIR>(x0, x1, x2, x3);
      wr[0] = x0;
      wr[QS] = x1;
      wr[2 * QS] = x2;
      wr[3 * QS] = x3;
    } else {
      float2 x0 = v[b], x1 = v[b + NB];
      bfly2<DIR>(x0, x1);
      wr[0] = x0;
      wr[QS] = x1;
    }
  }
  __syncthreads();
}

template <int LOG2M, int DIR, int PASS, bool LAST_REGS, int TWS = 1>
struct FftPasses {
  static __device__ __forceinline__ void run(float2 (&v)[16], float2 *s, const float2 *__restrict__ tw, int tid,
                                             const PassTwiddles<LOG2M, PASS> &cur) {
    if constexpr (PASS + 1 < fft_npasses(LOG2M)) {
      PassTwiddles<LOG2M, PASS + 1> nxt;
      load_pass_twiddles<LOG2M, DIR, PASS + 1, TWS>(nxt, tw, tid);  // in flight during this pass
      fft_pass<LOG2M, DIR, PASS>(v, s, cur, tid);
      FftPasses<LOG2M, DIR, PASS + 1, LAST_REGS, TWS>::run(v, s, tw, tid, nxt);
    } else {
      fft_pass<LOG2M, DIR, PASS, LAST_REGS>(v, s, cur, tid);
    }
  }
};

// Complex FFT of the M values held as v[m] = in[tid + T*m]; the natural-order result is left
// in the padded LDS image s[lds_pad(k)], visible to every thread (ends with a barrier).
template <int LOG2M, int DIR>
__device__ __forceinline__ void fft_regs_to_lds(float2 (&v)[16], float2 *s, const float2 *__restrict__ tw, int tid) {
  PassTwiddles<LOG2M, 0> none;  // pass 0 has unit twiddles
  FftPasses<LOG2M, DIR, 0, false>::run(v, s, tw, tid, none);
}

// Same transform, but the natural-order result stays in registers: v[m] = X[tid + T*m].  Every thread
// has finished READING the LDS image when this returns to any thread past its next barrier; the image
// may be overwritten after one __syncthreads() ... in fact the last pass already ends its reads with a
// barrier, so the caller may write LDS immediately.
template <int LOG2M, int DIR, int TWS = 1>
__device__ __forceinline__ void fft_regs_to_regs(float2 (&v)[16], float2 *s, const float2 *__restrict__ tw, int tid) {
  PassTwiddles<LOG2M, 0> none;
  FftPasses<LOG2M, DIR, 0, true, TWS>::run(v, s, tw, tid, none);
}

// ---- real <-> half-complex packing around the M-point complex transform (N = 2M real samples)
// Forward: Z = FFT_M(x[2n] + i x[2n+1]);  X[k] = E + w^k O,  X[M-k] = conj(E - w^k O),
//   E = (Z[k] + conj Z[M-k])/2,  O = -i (Z[k] - conj Z[M-k])/2,  w = exp(-i*pi/M).
// Spectrum layout: out[0] = (X[0], X[M]) (both real), out[k] = X[k] for 0 < k < M.
template <int LOG2M>
__device__ __forceinline__ void real_unpack_store(const float2 *s, const float2 *__restrict__ tw, int tid,
                                                  float2 *__restrict__ out) {
  constexpr int M = 1 << LOG2M, T = M / 16;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int k = tid + T * m;
    if (k == 0) {
      const float2 z0 = s[0], zh = s[lds_pad(M / 2)];
      out[0] = make_float2(z0.x + z0.y, z0.x - z0.y);
      out[M / 2] = cconj(zh);
    } else {
      const float2 zk = s[lds_pad(k)], zm = s[lds_pad(M - k)];
      const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
      const float2 d = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
      const float2 o = make_float2(d.y, -d.x);
      const float2 wo = cmul(tw[k], o);
      out[k] = cadd(e, wo);
      out[M - k] = cconj(csub(e, wo));
    }
  }
}

// Inverse: from the packed spectrum Y build Z[k] = E + iO (scaled) into LDS, ready for the
// inverse passes.  E = (Y[k] + conj Y[M-k])/2, O = conj(w^k) (Y[k] - conj Y[M-k])/2.
template <int LOG2M>
__device__ __forceinline__ void real_pack_load(const float2 *__restrict__ in, float2 *s,
                                               const float2 *__restrict__ tw, int tid, float scale) {
  constexpr int M = 1 << LOG2M, T = M / 16;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int k = tid + T * m;
    if (k == 0) {
      const float2 y0 = in[0], yh = in[M / 2];
      s[0] = make_float2(0.5f * scale * (y0.x + y0.y), 0.5f * scale * (y0.x - y0.y));
      s[lds_pad(M / 2)] = make_float2(scale * yh.x, -scale * yh.y);
    } else {
      const float2 yk = in[k], ym = in[M - k];
      const float hs = 0.5f * scale;
      const float2 e = make_float2(hs * (yk.x + ym.x), hs * (yk.y - ym.y));
      const float2 d = make_float2(hs * (yk.x - ym.x), hs * (yk.y + ym.y));
      const float2 o = cmul(cconj(tw[k]), d);
      s[lds_pad(k)] = make_float2(e.x - o.y, e.y + o.x);
      s[lds_pad(M - k)] = make_float2(e.x + o.y, o.x - e.y);
    }
  }
}

// Register-resident variants: the transform result / input lives in v[m] = Z[tid + T*m]; only the mirrored
// half (Z[M-k], owned by thread T-tid) goes through LDS, halving the LDS traffic of the packing step.
template <int LOG2M>
__device__ __forceinline__ void real_unpack_store_regs(const float2 (&v)[16], float2 *s, const float2 *__restrict__ tw,
                                                       int tid, float2 *__restrict__ out) {
  constexpr int M = 1 << LOG2M, T = M / 16;
  float2 w[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) w[m] = tw[tid + T * m];  // issued before the exchange
  float2 *own = s + lds_pad(tid);
#pragma unroll
  for (int m = 8; m < 16; ++m) own[lds_pad(T * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int k = tid + T * m;
    if (m == 0 && k == 0) {
      const float2 z0 = v[0], zh = v[8];  // thread 0 owns Z[0] and Z[M/2]
      out[0] = make_float2(z0.x + z0.y, z0.x - z0.y);
      out[M / 2] = cconj(zh);
    } else {
      const float2 zk = v[m], zm = s[lds_pad(M - k)];
      const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
      const float2 d = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
      const float2 o = make_float2(d.y, -d.x);
      const float2 wo = cmul(w[m], o);
      out[k] = cadd(e, wo);
      out[M - k] = cconj(csub(e, wo));
    }
  }
}

template <int LOG2M>
__device__ __forceinline__ void real_pack_issue(const float2 *__restrict__ in, float2 (&yk)[8], float2 (&ym)[8], int tid) {
  constexpr int M = 1 << LOG2M, T = M / 16;
#pragma unroll
  for (int m = 0; m < 8; ++m) {  // global loads only: the caller decides what runs while they are in flight
    const int k = tid + T * m;
    yk[m] = in[k];
    ym[m] = in[k == 0 ? M / 2 : M - k];
  }
}

template <int LOG2M>
__device__ __forceinline__ void real_pack_finish(const float2 (&yk)[8], const float2 (&ym)[8], float2 (&v)[16], float2 *s,
                                                 const float2 *__restrict__ tw, int tid, float scale) {
  constexpr int M = 1 << LOG2M, T = M / 16;
  float2 w[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) w[m] = tw[tid + T * m];
  const float hs = 0.5f * scale;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int k = tid + T * m;
    if (k == 0) {
      v[0] = make_float2(hs * (yk[0].x + yk[0].y), hs * (yk[0].x - yk[0].y));
      s[lds_pad(M / 2)] = make_float2(scale * ym[0].x, -scale * ym[0].y);
    } else {
      const float2 e = make_float2(hs * (yk[m].x + ym[m].x), hs * (yk[m].y - ym[m].y));
      const float2 d = make_float2(hs * (yk[m].x - ym[m].x), hs * (yk[m].y + ym[m].y));
      const float2 o = cmul(cconj(w[m]), d);
      v[m] = make_float2(e.x - o.y, e.y + o.x);
      s[lds_pad(M - k)] = make_float2(e.x + o.y, o.x - e.y);
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 8; m < 16; ++m) v[m] = s[lds_pad(tid + T * m)];
  __syncthreads();
}

template <int LOG2M>
__device__ __forceinline__ void real_pack_load_regs(const float2 *__restrict__ in, float2 (&v)[16], float2 *s,
                                                    const float2 *__restrict__ tw, int tid, float scale) {
  float2 yk[8], ym[8];
  real_pack_issue<LOG2M>(in, yk, ym, tid);
  real_pack_finish<LOG2M>(yk, ym, v, s, tw, tid, scale);
}

// ---- Split transforms: one block of M = 2^LOG2M complex points done as TWO transforms of M/2 points by a
// workgroup of M/32 threads, in an LDS image of M/2 points.  Halving the image doubles the workgroups a CU
// can hold (B = 8192: 4 x 35 KB instead of 2 x 70 KB), which is what hides HBM latency in these kernels.
//   forward (decimation in frequency): Z[2k]   = FFT_{M/2}( z[n] + z[n+M/2] )[k]
//                                      Z[2k+1] = FFT_{M/2}( (z[n] - z[n+M/2]) * w^n )[k],  w = exp(-2*pi*i/M)
//   inverse (decimation in time):      z[n+M/2] = A[n] - conj(w)^n B[n],  A = IFFT_{M/2}(Z[2k]), B = IFFT_{M/2}(Z[2k+1])
// Real packing works per half because bins pair up within a parity class: (2k, M-2k) and (2k+1, M-2k-1).
// `tw` is the table of the FULL size: tw[j] = exp(-i*pi*j/M).
//
// Even half, forward: v[m] = Z[2k], k = tid + T*m (T = M/32).  Stores X[2k] and X[M-2k].
template <int LOG2M>
__device__ __forceinline__ void split_unpack_store_even(const float2 (&v)[16], float2 *s, const float2 *__restrict__ tw,
                                                        int tid, float2 *__restrict__ out) {
  constexpr int M = 1 << LOG2M, MH = M / 2, T = MH / 16;
  float2 w[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) w[m] = tw[2 * (tid + T * m)];
#pragma unroll
  for (int m = 8; m < 16; ++m) s[lds_pad(tid + T * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int k = tid + T * m;
    if (k == 0) {
      const float2 z0 = v[0], zh = v[8];  // Z[0] and Z[M/2] = even-half element MH/2
      out[0] = make_float2(z0.x + z0.y, z0.x - z0.y);
      out[M / 2] = cconj(zh);
    } else {
      const float2 zk = v[m], zm = s[lds_pad(MH - k)];
      const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
      const float2 d = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
      const float2 wo = cmul(w[m], make_float2(d.y, -d.x));
      out[2 * k] = cadd(e, wo);
      out[M - 2 * k] = cconj(csub(e, wo));
    }
  }
  __syncthreads();
}

// Odd half, forward: v[m] = Z[2k+1].  Partner of bin 2k+1 is bin M-2k-1 = odd-half element MH-1-k.
template <int LOG2M>
__device__ __forceinline__ void split_unpack_store_odd(const float2 (&v)[16], float2 *s, const float2 *__restrict__ tw,
                                                       int tid, float2 *__restrict__ out) {
  constexpr int M = 1 << LOG2M, MH = M / 2, T = MH / 16;
  float2 w[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) w[m] = tw[2 * (tid + T * m) + 1];
#pragma unroll
  for (int m = 8; m < 16; ++m) s[lds_pad(tid + T * m)] = v[m];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int k = tid + T * m;
    const float2 zk = v[m], zm = s[lds_pad(MH - 1 - k)];
    const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
    const float2 d = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
    const float2 wo = cmul(w[m], make_float2(d.y, -d.x));
    out[2 * k + 1] = cadd(e, wo);
    out[M - 2 * k - 1] = cconj(csub(e, wo));
  }
  __syncthreads();
}

// Inverse, one parity class: builds the half-size transform input v[m] = Z[2k + ODD] (scaled) from the packed
// spectrum `in` of the full block.
template <int LOG2M, int ODD>
__device__ __forceinline__ void split_pack_load(const float2 *__restrict__ in, float2 (&v)[16], float2 *s,
                                                const float2 *__restrict__ tw, int tid, float scale) {
  constexpr int M = 1 << LOG2M, MH = M / 2, T = MH / 16;
  float2 yk[8], ym[8], w[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int j = 2 * (tid + T * m) + ODD;
    yk[m] = in[j];
    ym[m] = in[j == 0 ? M / 2 : M - j];
    w[m] = tw[j];
  }
  const float hs = 0.5f * scale;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int k = tid + T * m;
    if (ODD == 0 && k == 0) {
      v[0] = make_float2(hs * (yk[0].x + yk[0].y), hs * (yk[0].x - yk[0].y));
      s[lds_pad(MH / 2)] = make_float2(scale * ym[0].x, -scale * ym[0].y);
    } else {
      const float2 e = make_float2(hs * (yk[m].x + ym[m].x), hs * (yk[m].y - ym[m].y));
      const float2 d = make_float2(hs * (yk[m].x - ym[m].x), hs * (yk[m].y + ym[m].y));
      const float2 o = cmul(cconj(w[m]), d);
      v[m] = make_float2(e.x - o.y, e.y + o.x);
      s[lds_pad(MH - ODD - k)] = make_float2(e.x + o.y, o.x - e.y);  // Z[M - j] = element MH-k (even) / MH-1-k (odd)
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 8; m < 16; ++m) v[m] = s[lds_pad(tid + T * m)];
  __syncthreads();
}

}  // namespace al
