// eaqhm_ls_common.h — device pieces shared by the LS kernels (gfx950 only, FP64).
//
// Column order of the LS unknowns everywhere: [n negative columns | DC | n positive columns], Kc = 2n+1,
// then the same again for the slopes (functions.py:455 / :519: E = [E2, n*E2]).
#pragma once
#include "eaqhm_common.h"

namespace eaqhm {

// Singular systems.  The reference solves with inv() (functions.py:465, :530), which raises only on an EXACTLY zero LU
// pivot and otherwise returns whatever the ill-conditioned system gives.  The kernels factorise by Cholesky; what
// corresponds to the exact zero is a breakdown of the factorisation: a pivot (the squared diagonal of L) that is not
// positive or is lost in the rounding of its own update, i.e. <= PIVOT_TOL = order * eps (order <= ~1200) of the ORIGINAL
// diagonal entry.  Exactly duplicated basis columns land there (their pivot is d - d(1 +- a few eps)); systems that are
// merely ill-conditioned (scaled cond up to ~1e12) do not, and are solved like the reference solves them.
#define PIVOT_TOL 2.5e-13

#define LS_NCLS 7
#define LS_BIG_CLASS (LS_NCLS - 1)

struct LsArgs {
  int mode;  // 0: adaptation 0 (stationary harmonics), 1: adaptation >= 1 (tracks)
  const double* s; long long L; double fs;
  // dense tracks of the previous adaptation: rows of Lt samples that hold samples [t0, t0 + Lt) of the file.  The two
  // pointers (and zloc) are BIASED by -t0 on the host side of the C ABI, so that row k, absolute sample t is at
  // [k * Lt + t] — the kernels index with absolute sample numbers whatever window of the file is resident.
  const double* am_cur; const double* fm_cur; long long Lt; long long trk_t0; int Kmax;
  const int* frame_inst; const int* frame_c; const int* frame_wl; const double* frame_f0; const int* frame_K;
  const int* ncol; const int* cols; const unsigned char* seeded; const int* any_seed;
  int n_frames; int a_iter; double f0_stale; double f0min;
  double* records; double* raw_amp; double* raw_slope;
  double* scratch; size_t scratch_stride; int nmax; int Nmax; int Kcmax;
  int* work_counter;  // dynamic frame queue; may be null
  // zero counts of the frequency tracks (tile variant, mode 1; eaqhm_ls_zero_prefix_kernel): zloc[k][t] = zeros of
  // fm_cur[k] from the start of t's 1024-sample chunk up to t, ztot[k][chunk] = zeros of the whole chunk
  const unsigned short* zloc; const int* ztot; int zchunks;
  unsigned char* zflag;   // [zchunks] chunks some frame window of this launch touches (only those are counted)
  // frames bucketed by size (tile variant): cls[0..6] counts, cls[8..14] cursors, cls[16 + c*n_frames + i] frame ids.
  // Classes 0-5 are the register budgets of eaqhm_ls_tile_kernel, class LS_BIG_CLASS is left to eaqhm_ls_mfma_kernel.
  int* cls;
  unsigned long long* debug;  // phase stamps (16 x u64)
  int debug_diag;             // also time diag_D and the gaps between two of them (slots 9, 11, 13, 14; costs ~8 %)
  int* fault;                 // device counters (eaqhm_ctx::faults): [0] singular systems, [1] stalled diagonal pipelines,
                              // [2] frames whose window is not inside the resident track window (dropped, never read)
};

// The contract of eaqhm_ls_batch checked on the device: the frame's window [c - wl, c + wl] lies inside the signal and —
// adaptations >= 1, together with the sample before it (the zero counts are differences of running counts) — inside the
// resident window of the tracks.  A frame that fails is dropped and counted (fault[2]) instead of being read out of bounds.
__device__ inline bool frame_window_ok(const LsArgs& A, long long c, long long wl) {
  bool inside = wl > 0 && c - wl >= 0 && c + wl < A.L;
  if (A.mode == 1) inside = inside && c + wl < A.trk_t0 + A.Lt && (c - wl - 1 >= A.trk_t0 || (A.trk_t0 == 0 && c - wl == 0));
  return inside;
}

// wave-uniform values that the compiler cannot prove uniform (loaded through per-lane pointers, passed in vector
// registers): moved to scalar registers, where the arithmetic on them costs no VGPRs and no VALU cycles
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline double uni(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <typename T>
__device__ inline T* uni(T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}

// seed-aware track access (functions.py:209-210; see eaqhm_frame_prep): a seeded row shows 140 Hz / 10e-4
// in slot 0 to the frames at or after it, exactly like the sequential write of the reference
__device__ inline double track_fm(const LsArgs& A, int k, long long t, int c, bool seeds) {
  if (seeds && k == 0 && t <= c && A.seeded[t]) return 140.0;
  return A.fm_cur[(size_t)k * A.Lt + t];
}
__device__ inline double track_am(const LsArgs& A, int k, long long t, int c, bool seeds) {
  if (seeds && k == 0 && t <= c && A.seeded[t]) return 10e-4;
  return A.am_cur[(size_t)k * A.Lt + t];
}

// sin and cos of a double in one go with a small register footprint: 3-term Cody-Waite reduction by pi/2
// through FMAs (exact products), then the fdlibm kernel polynomials on [-pi/4, pi/4].  Error < 1 ulp for
// |x| < ~1e8 (basis phases here are below 1e4 rad).  The libm sincos() spills the MFMA accumulators that are
// live across the basis build; this one does not.
__device__ inline void sincos_cw(double x, double* sn, double* cs) {
  const double n = rint(x * 6.36619772367581382433e-01);  // 2/pi
  double r = fma(-n, 1.57079632679489655800e+00, x);       // pi/2 split in three doubles
  r = fma(-n, 6.12323399573676603587e-17, r);
  r = fma(-n, -1.49738490485916983294e-33, r);
  const double z = r * r;
  // fdlibm __kernel_sin / __kernel_cos coefficients
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double s0 = fma(r * z, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double c0 = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)n & 3;
  const double sv = (q & 1) ? c0 : s0, cv = (q & 1) ? s0 : c0;
  *sn = (q & 2) ? -sv : sv;
  *cs = ((q + 1) & 2) ? -cv : cv;
}

// numpy.blackman / numpy.hamming (symmetric form: n = 2u - (N-1))
__device__ inline double window_value(int blackman, int u, int N) {
  double n = (double)(2 * u - (N - 1));
  double den = (double)(N - 1);
  double s1, c1;
  sincos_cw(M_PI * n / den, &s1, &c1);
  if (!blackman) return 0.54 + 0.46 * c1;
  double s2, c2;
  sincos_cw(2.0 * M_PI * n / den, &s2, &c2);
  return 0.42 + 0.5 * c1 + 0.08 * c2;
}

// ---- adaptation 0: the Gramian in closed form ------------------------------------------------------------------
// At adaptation 0 the basis columns are exp(j h theta n), h = -K..K, theta = 2 pi f0 / fs, on the symmetric grid
// n = -wl..wl with a symmetric window (functions.py:444-455), so every entry of the three Gramian blocks depends on
// the DIFFERENCE of the two harmonic numbers only (the blocks are Toeplitz):
//     sum_n w^2 n^p exp(j m theta n)   =   c0[|m|]  (p = 0, real) | j sgn(m) s1[|m|]  (p = 1) | c2[|m|]  (p = 2, real),
// 3 (2K+1) real sums of wl terms instead of a contraction over N x (2 Kc)^2, and the right-hand sides are K+1 complex
// sums  r_p[h] = sum_n w^2 n^p s_n exp(j h theta n).  The tables are built here (pairs +-t share a rotation;
// exp(j m theta t) advances by complex rotation and is re-seeded with an exactly evaluated value every 64 samples;
// fixed two-level summation order), then the kernels fill their system tiles from them: no basis image, no MFMA
// contraction for adaptation 0.
//   tab [TZ_NQ][TB], part [NCH][TZ_NQ][TB], W2 / PA / PB [wl+1] (LDS);  win, sig: the frame's window and signal (LDS)
#define TZ_NQ 7       // c0, s1, c2, Re r0, Im r0, Re r1, Im r1
__device__ inline void toeplitz_tables(double* tab, double* part, double* W2, double* PA, double* PB, double* ssq,
                                       const double* win, const double* sig, int n, int wl, double theta, int tid,
                                       int TB, int NCH_) {
  const int mid = wl, nthr = blockDim.x;
  for (int t = tid; t <= wl; t += nthr) {
    const double w = win[mid + t], w2 = w * w, sp = sig[mid + t], sm = sig[mid - t];
    W2[t] = w2;
    PA[t] = w2 * (sp + sm);
    PB[t] = w2 * (sp - sm);
  }
  __syncthreads();
  if (tid < 64) {   // signal energy sum w^2 s^2 (fixed summation order)
    double e = 0.0;
    for (int t = tid; t <= wl; t += 64) {
      const double sp = sig[mid + t], sm = sig[mid - t];
      e += W2[t] * ((t == 0) ? sp * sp : (sp * sp + sm * sm));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
    if (tid == 0) ssq[0] = e;
  }
  // chunks of the t range: as many (up to NCH_) as make rounds-of-tasks x terms-per-task smallest — 8 chunks of 65 sums are
  // 520 tasks for 512 threads, two rounds of 15 terms where 7 chunks give one of 18
  const int nm = 2 * n + 1;
  int NCH = 1;
  {
    int best = 0x7fffffff;
    for (int q = 1; q <= NCH_; ++q) {
      const int cost = ((nm * q + nthr - 1) / nthr) * ((wl + q - 1) / q + 4);   // (+4: the seeds of a task)
      if (cost < best) { best = cost; NCH = q; }
    }
  }
  const int CL = (wl + NCH - 1) / NCH;
  for (int task = tid; task < nm * NCH; task += nthr) {
    const int ch = task / nm, m = task - ch * nm;
    const int t0 = 1 + ch * CL, t1 = (t0 + CL - 1 < wl) ? (t0 + CL - 1) : wl;
    const double phi = (double)m * theta;
    double zr, zi, sr, si;
    sincos_cw((double)t0 * phi, &zi, &zr);
    sincos_cw(phi, &si, &sr);
    double a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    const bool rhs = m <= n;
    for (int t = t0; t <= t1; ++t) {
      if (t > t0 && ((t - t0) & 63) == 0) sincos_cw((double)t * phi, &zi, &zr);   // re-seed: bounds the rotation's drift
      const double w2 = W2[t], tt = (double)t, w2t = w2 * tt;
      a0 = fma(w2, zr, a0);
      a1 = fma(w2t, zi, a1);
      a2 = fma(w2t * tt, zr, a2);
      if (rhs) {
        const double pa = PA[t], pb = PB[t];
        b0 = fma(pa, zr, b0);
        b1 = fma(pb, zi, b1);
        b2 = fma(tt * pb, zr, b2);
        b3 = fma(tt * pa, zi, b3);
      }
      const double nr = zr * sr - zi * si, ni = zr * si + zi * sr;
      zr = nr; zi = ni;
    }
    double* pp = part + (size_t)(ch * TZ_NQ) * TB + m;
    pp[0] = a0; pp[TB] = a1; pp[2 * TB] = a2;
    pp[3 * TB] = b0; pp[4 * TB] = b1; pp[5 * TB] = b2; pp[6 * TB] = b3;
  }
  __syncthreads();
  for (int q = tid; q < TZ_NQ * nm; q += nthr) {
    const int k = q / nm, m = q - k * nm;
    double v = 0.0;
    for (int ch = 0; ch < NCH; ++ch) v += part[(size_t)(ch * TZ_NQ + k) * TB + m];
    if (k < 3) v *= 2.0;                                   // the pair (+t, -t)
    if (k == 0) v += W2[0];                                // t = 0
    if (k == 3 && m <= n) v += W2[0] * sig[mid];
    tab[k * TB + m] = v;
  }
  __syncthreads();
}

// harmonic number of basis column c (0 <= c < Kc): [negative block: -(c+1) | DC | positive block]
__device__ inline int toeplitz_harm(int c, int n) { return (c < n) ? -(c + 1) : (c - n); }
// Gramian entry  sum_n w^2 n^p conj(x_a) x_b  of columns a, b (p = number of slope columns among them)
__device__ inline void toeplitz_gram(const double* tab, int TB, int p, int a, int b, int n, double& re, double& im) {
  const int m = toeplitz_harm(b, n) - toeplitz_harm(a, n), am = (m < 0) ? -m : m;
  re = 0.0; im = 0.0;
  if (p == 1) im = (m < 0) ? -tab[TB + am] : tab[TB + am];
  else re = tab[p * TB + am];
}
// right-hand-side row entry  sum_n w^2 n^p s_n x_b  for column b of block p (0: amplitudes, 1: slopes)
__device__ inline void toeplitz_rhs(const double* tab, int TB, int p, int b, int n, double& re, double& im) {
  const int h = toeplitz_harm(b, n), ah = (h < 0) ? -h : h;
  re = tab[(3 + 2 * p) * TB + ah];
  im = tab[(4 + 2 * p) * TB + ah];
  if (h < 0) im = -im;                                      // r[-h] = conj r[h]
}

// logical column cc of the chunk rows of sample pair el lives at XCOL(cc, el): the 16 lanes that write one column of 16
// different pairs hit 16 different LDS banks, and the MFMA operand reads (column (lcol + row / 2) & 15) stay conflict-free
#define XCOL(cc, el) (((cc) & ~15) | (((cc) + (el)) & 15))

// ---- Phase A1 -----------------------------------------------------------------------------------------------
// Per-slot set-up.  A slot whose track has no zero inside the frame's window (two look-ups in the zero counts;
// the common case) needs nothing but its centre values: the basis build reads the track itself and integrates
// the frequency on the fly, outwards from the centre.  A slot with gaps gets its window bridged
// (functions.py:251-278) into the workgroup's scratch rows, which the build then reads instead of the track.
//   ci[j]: [0] running sum of fm over (mid, mid+d], [1] over [mid-d, mid]  (carried from chunk to chunk),
//          [2] / [3] pointers to the slot's fm / am window (track or bridged copy),
//          [STRIDE-3] 1/(am_mid+eps), [STRIDE-2..STRIDE-1] rho = exp(j 2 pi fm_mid / fs)      (functions.py:508-518, :284-285)
// masks: [n][nchs] nonzero masks of the 64-sample chunks of the slots with gaps, gappy: [n] flags (work space).
template <int STRIDE, int NWAVES>
__device__ inline void prepare_slots(const LsArgs& A, double* Qf, double* Af, int Npad, double* ci,
                                     unsigned long long* masks, int* gappy, const int* mycols, int n, int N, int mid,
                                     int c, int wl, bool seeds, int lane, int wave, int nchs) {
  const double eps = 10e-5;  // functions.py:517
  const int nch = (N + 63) >> 6;
  const long long t0 = (long long)c - wl;
  // the argument block is read through a per-lane pointer: fetch what this function uses once, into scalars
  const unsigned short* zloc = uni(A.zloc);
  const int* ztot = uni(A.ztot);
  const double* fm_all = uni(A.fm_cur);
  const double* am_all = uni(A.am_cur);
  const long long L = ((long long)uni((int)(A.Lt >> 32)) << 32) | (unsigned)uni((int)(A.Lt & 0xffffffffll));   // row stride of the tracks
  const int zchunks = uni(A.zchunks);
  const double w1 = uni(2.0 * M_PI / A.fs);
  auto centre = [&](int j, double fv, double av) {
    ci[j * STRIDE + STRIDE - 3] = 1.0 / (av + eps);
    double sn, cs;
    sincos_cw(fv * w1, &sn, &cs);
    ci[j * STRIDE + STRIDE - 2] = cs;
    ci[j * STRIDE + STRIDE - 1] = sn;
  };
  int anyg = 0;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    // zeros of the track inside [c-wl, c+wl] from the chunked zero counts;
    // a seeded slot 0 (functions.py:209-210) shows substituted values: route it through the bridged copy too
    const int k = mycols[j];
    const long long b = (long long)c + wl, a1 = (long long)c - wl - 1;
    const int cb = (int)(b >> 10), ca = (a1 >= 0) ? (int)(a1 >> 10) : 0;
    // (the centre values are requested together with the zero counts: one memory round trip, not two)
    const double fmc = fm_all[(size_t)k * L + c], amc = am_all[(size_t)k * L + c];
    int zc = zloc[(size_t)k * L + b];
    for (int ch = ca; ch < cb; ++ch) zc += ztot[(size_t)k * zchunks + ch];   // (one chunk boundary inside the window, rarely two)
    if (a1 >= 0) zc -= zloc[(size_t)k * L + a1];
    const int g = (zc != 0 || (seeds && k == 0)) ? 1 : 0;
    gappy[j] = g;
    anyg |= g;
    ci[j * STRIDE + 0] = 0.0;
    ci[j * STRIDE + 1] = 0.0;
    // where the build reads this slot's window from: the bridged copy or the track itself
    const size_t trk = (size_t)k * L + t0;
    ((const double**)(ci + j * STRIDE))[2] = g ? (Qf + (size_t)j * Npad) : (fm_all + trk);
    ((const double**)(ci + j * STRIDE))[3] = g ? (Af + (size_t)j * Npad) : (am_all + trk);
    if (!g) centre(j, fmc, amc);   // (a seeded slot 0 is never gap-free)
  }
  if (!__syncthreads_or(anyg)) return;
  // slots with gaps: nonzero masks of every 64-sample chunk first, then the bridged window
  for (int it = wave; it < n * nch; it += NWAVES) {
    const int j = it / nch, ch = it - j * nch, t = (ch << 6) + lane;
    if (!gappy[j]) continue;
    const double fv = (t < N) ? track_fm(A, mycols[j], t0 + t, c, seeds) : 0.0;
    const unsigned long long m = __ballot(fv != 0.0);
    if (lane == 0) masks[j * nchs + ch] = m;
  }
  __syncthreads();
  for (int it = wave; it < n * nch; it += NWAVES) {
    const int j = it / nch, ch = it - j * nch, t = (ch << 6) + lane, k = mycols[j];
    if (!gappy[j] || t >= N) continue;
    double fv = track_fm(A, k, t0 + t, c, seeds), av = track_am(A, k, t0 + t, c, seeds);
    if (fv == 0.0) {  // nearest nonzero samples on both sides (functions.py:251-278)
      int p = -1, q = -1;
      {
        unsigned long long m = masks[j * nchs + ch] & ((lane == 0) ? 0ull : (~0ull >> (64 - lane)));
        int cc = ch;
        while (m == 0ull && cc > 0) { --cc; m = masks[j * nchs + cc]; }
        if (m != 0ull) p = (cc << 6) + 63 - __clzll((long long)m);
      }
      {
        unsigned long long m = masks[j * nchs + ch] & ((lane == 63) ? 0ull : (~0ull << (lane + 1)));
        int cc = ch;
        while (m == 0ull && cc < nch - 1) { ++cc; m = masks[j * nchs + cc]; }
        if (m != 0ull) q = (cc << 6) + __ffsll((long long)m) - 1;
      }
      if (p < 0) {         // leading gap: hold the first nonzero (functions.py:259-263)
        fv = track_fm(A, k, t0 + q, c, seeds); av = track_am(A, k, t0 + q, c, seeds);
      } else if (q < 0) {  // trailing gap: hold the last nonzero (functions.py:265-271)
        fv = track_fm(A, k, t0 + p, c, seeds); av = track_am(A, k, t0 + p, c, seeds);
      } else {             // interior gap: linear (functions.py:277-278)
        const double f0v = track_fm(A, k, t0 + p, c, seeds), f1v = track_fm(A, k, t0 + q, c, seeds);
        const double a0v = track_am(A, k, t0 + p, c, seeds), a1v = track_am(A, k, t0 + q, c, seeds);
        const double dx = (double)(q - p), xx = (double)(t - p);
        fv = ((f1v - f0v) / dx) * xx + f0v;
        av = ((a1v - a0v) / dx) * xx + a0v;
      }
    }
    Qf[(size_t)j * Npad + t] = fv;
    Af[(size_t)j * Npad + t] = av;
    if (t == mid) centre(j, fv, av);
  }
  __syncthreads();
}

// inclusive sum over each aligned group of 16 lanes (DPP row shifts: a row is 16 lanes, zeros are shifted in)
template <int CTRL>
__device__ inline double dpp_row(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double scan16(double x) {
  x += dpp_row<0x111>(x);  // row_shr:1
  x += dpp_row<0x112>(x);  // row_shr:2
  x += dpp_row<0x114>(x);  // row_shr:4
  x += dpp_row<0x118>(x);  // row_shr:8
  return x;
}

// Phase C (round-1 form): left-looking complex Cholesky on transposed storage Lt[k][i] = R[i][k] (coalesced
// over rows), the right-hand side carried as row M (so the forward solve comes for free), then back
// substitution by wave 0.  Result x (M complex, interleaved) in LDS `xs`.
__device__ inline void cholesky_solve(double* __restrict__ Lt, int M, int ldl, double* rowj, double* xs, double* sh,
                                      int* fault) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int j = 0; j < M; ++j) {
    for (int k = tid; k < j; k += nt) {  // row j of L (entries k < j) -> LDS
      size_t o = ((size_t)k * ldl + j) * 2;
      rowj[2 * k] = Lt[o];
      rowj[2 * k + 1] = Lt[o + 1];
    }
    __syncthreads();
    for (int i = j + tid; i <= M; i += nt) {
      size_t oj = ((size_t)j * ldl + i) * 2;
      double ar = Lt[oj], ai = Lt[oj + 1];
      for (int k = 0; k < j; ++k) {
        size_t o = ((size_t)k * ldl + i) * 2;
        double lr = Lt[o], li = Lt[o + 1];
        double cr = rowj[2 * k], ci = rowj[2 * k + 1];
        ar -= lr * cr + li * ci;  // L[i][k] * conj(L[j][k])
        ai -= li * cr - lr * ci;
      }
      if (i == j) {
        if (!(ar > PIVOT_TOL * Lt[oj])) { atomicAdd(fault, 1); ar = 1.0; }   // Cholesky breakdown: singular normal matrix
        double d = sqrt(ar);
        sh[0] = d;
        Lt[oj] = d; Lt[oj + 1] = 0.0;
      } else {
        Lt[oj] = ar; Lt[oj + 1] = ai;  // scaled below
      }
    }
    __syncthreads();
    double inv = 1.0 / sh[0];
    for (int i = j + 1 + tid; i <= M; i += nt) {
      size_t oj = ((size_t)j * ldl + i) * 2;
      Lt[oj] *= inv; Lt[oj + 1] *= inv;
    }
    __syncthreads();
  }
  // back substitution: y_j = conj(L[M][j]); x_j = (y_j - sum_{i>j} conj(L[i][j]) x_i) / L[j][j]
  if (tid < 64) {
    for (int j = M - 1; j >= 0; --j) {
      const double* row = Lt + (size_t)j * ldl * 2;
      double sr = 0, si = 0;
      for (int i = j + 1 + tid; i < M; i += 64) {
        double lr = row[2 * i], li = row[2 * i + 1];
        double xr = xs[2 * i], xi = xs[2 * i + 1];
        sr += lr * xr + li * xi;  // conj(l) * x
        si += lr * xi - li * xr;
      }
      for (int o = 32; o > 0; o >>= 1) {
        sr += __shfl_xor(sr, o);
        si += __shfl_xor(si, o);
      }
      if (tid == 0) {
        double d = row[2 * j];
        double yr = row[2 * M], yi = -row[2 * M + 1];
        xs[2 * j] = (yr - sr) / d;
        xs[2 * j + 1] = (yi - si) / d;
      }
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
    }
  }
  __syncthreads();
}

// Phase D: raw solution (optional), frequency mismatch (functions.py:297), amplitude floor and acceptance
// (:309-315), record row (:316-324, :303).  xs = [amplitudes (Kc) | slopes (Kc)] interleaved complex in LDS;
// sh needs 8 doubles (one partial maximum per wave at sh[1..]).
__device__ inline void write_record(const LsArgs& A, const double* xs, double* sh, const int* mycols, int f, int n,
                                    int inst, int c, double f0, bool seeds) {
  const int tid = threadIdx.x, nt = blockDim.x, Kc = 2 * n + 1;
  if (A.raw_amp) {
    // in the order the seam functions return them.  Adaptation 0: f0range = (-K..K) f0 (functions.py:189), while the
    // kernels keep the negative block as the conjugates of the positive one in ITS order: reversed on the way out.
    // Adaptations >= 1: the negative block keeps the column order of the positive one (functions.py:284).
    const int stride = 2 * (2 * A.Kmax + 1);
    for (int q = tid; q < 2 * Kc; q += nt) {
      const int col = q >> 1, src = (A.mode == 0 && col < n) ? (n - 1 - col) : col;
      A.raw_amp[(size_t)f * stride + q] = xs[2 * src + (q & 1)];
      A.raw_slope[(size_t)f * stride + q] = xs[2 * Kc + 2 * src + (q & 1)];
    }
  }
  double amax = 0.0;  // amplitude floor over the positive slots (functions.py:309)
  for (int j = tid; j < n; j += nt) {
    double ar = xs[2 * (n + 1 + j)], ai = xs[2 * (n + 1 + j) + 1];
    amax = fmax(amax, sqrt(ar * ar + ai * ai));   // (amplitudes are O(1): no overflow to guard against)
  }
  for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_xor(amax, o));
  if ((tid & 63) == 0) sh[1 + (tid >> 6)] = amax;
  __syncthreads();
  amax = 0.0;
  for (int w = 0; w < (nt >> 6); ++w) amax = fmax(amax, sh[1 + w]);
  // 20 log10 |a| > 20 log10 max|a| - 150  (functions.py:309, :315)  <=>  |a| > max|a| * 10^-7.5; a silent frame
  // (max|a| = 0: -inf > -inf in the reference) accepts nothing either way
  const double floor_mag = amax * 3.1622776601683795e-08;
  const double h = f0 / (double)(A.a_iter + 1);  // functions.py:310
  double* rec = A.records + (size_t)inst * (3 * A.Kmax + 1);
  for (int k = tid; k < 3 * A.Kmax; k += nt) rec[k] = 0.0;
  __syncthreads();
  for (int j = tid; j < n; j += nt) {
    const int k = (A.mode == 0) ? j : mycols[j];
    double ar = xs[2 * (n + 1 + j)], ai = xs[2 * (n + 1 + j) + 1];
    double br = xs[2 * (Kc + n + 1 + j)], bi = xs[2 * (Kc + n + 1 + j) + 1];
    double mag = sqrt(ar * ar + ai * ai);
    double eta = 0.0;
    if (A.mode == 1) eta = A.fs / (2.0 * M_PI) * ((ar * bi - ai * br) / (mag * mag));  // functions.py:297
    if (mag > floor_mag && fabs(eta) < h) {
      rec[k] = mag;
      rec[2 * A.Kmax + k] = atan2(ai, ar);
      double fmv;
      if (A.mode == 0) fmv = (double)(k + 1) * f0;
      else {
        double cur = track_fm(A, k, c, c, seeds);
        fmv = (f0 > A.f0min) ? cur + eta : cur;
      }
      rec[A.Kmax + k] = fmv;
    }
  }
  if (tid == 0) rec[3 * A.Kmax] = xs[2 * n];  // Re(a_DC) (functions.py:303)
  __syncthreads();
}

}  // namespace eaqhm
