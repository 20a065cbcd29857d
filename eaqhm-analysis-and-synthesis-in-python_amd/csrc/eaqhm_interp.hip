// eaqhm_interp.hip — track interpolation, phase integration, additive synthesis and SRER.
// gfx950 (MI355X) only, FP64.
//
//   eaqhm_spline_kernel   one thread per (instant, slot): runs of consecutive accepted instants
//                         (functions.py:350-362) and the not-a-knot cubic second derivatives on their knots
//                         (functions.py:340, :367; interp1d(kind=3)) through the closed-form inverse of the
//                         (1,4,1) system — a local 69-term sum instead of a sequential tridiagonal sweep.
//   eaqhm_spline_edge_kernel  not-a-knot end conditions of every run from its neighbours' moments.
//   eaqhm_eval_kernel     one block per 16-64 samples x all slots: each knot interval is integrated once (linear
//                         am, functions.py:364; cubic fm, :367-371 incl. the padded <4-knot case; phase by
//                         frequency integration with the sine-bump correction, :537-575) into LDS tables; then
//                         per (sample, slot) the next-iteration frequency from the unwrapped phase (:375) and
//                         am*cos(ph); per sample the synthesis a0 + 2*sum am*cos(ph) (:385) and the partial sums
//                         of the reconstruction error (:388).  Every (slot, sample) cell of the two dense
//                         outputs is written, so no memset is needed between adaptations.
//   eaqhm_srer_kernel     deterministic final reduction + SRER in dB.
//
// Bandwidth-type stage: per adaptation it reads (1+3*Kmax)*8*No_ti bytes of records and writes
// 2*Kmax*8*L bytes of dense tracks (the reference keeps seven dense L x Kmax arrays; only am_current and
// fm_current are ever read again, so only those two are materialised here).
#include "eaqhm_common.h"

namespace eaqhm {

// ------------------------------------------------------------------------------------------------
// Not-a-knot cubic spline moments on uniformly spaced knots, one thread per (instant, slot).
//
// For a run of m >= 4 knots the second derivatives satisfy 6*M_1 = d_1, 6*M_{m-2} = d_{m-2},
// M_0 = 2*M_1 - M_2, M_{m-1} = 2*M_{m-2} - M_{m-3}, and for the n = m-4 inner knots the (1,4,1) Toeplitz
// system T x = dt with dt = d (minus M_1 / M_{m-2} in its first / last row).  T^-1 is known in closed form:
//   (T^-1)_{ij} = (-1)^(i+j) lam^(|i-j|+1) (1-lam^(2 min(i,j))) (1-lam^(2 (n+1-max(i,j))))
//                 / ((1-lam^2) (1-lam^(2(n+1)))),         lam = 2 - sqrt(3),
// and lam^35 < 1e-20, so every moment is a 69-term local sum: no sequential sweep over the (possibly tens
// of thousands of) knots of a run, and only the distance to the run's ends up to 40 knots is ever needed.
#define SPL_W 34
#define SPL_CAP 40
#define SPL_BIG 1000000

struct SplineCol {
  const double* rec; int RS, off_y, off_acc, No_ti; bool all_acc; double h2;
  __device__ double y(int i) const { return rec[(size_t)i * RS + off_y]; }
  __device__ bool acc(int i) const {
    return i >= 0 && i < No_ti && (all_acc || rec[(size_t)i * RS + off_acc] != 0.0);
  }
  __device__ double d2(int i) const { return 6.0 * ((y(i - 1) - 2.0 * y(i)) + y(i + 1)) / h2; }
};

__device__ inline double lam_pow(const double* lp, int e) { return e > 79 ? 0.0 : lp[e]; }

// moment of a knot that is neither the first nor the last of its run; A / B = distance (in knots) to the
// first / last knot of the run (>= 1), SPL_BIG when farther than SPL_CAP
__device__ double inner_moment(const SplineCol& C, const double* lp, int i, int A, int B) {
  if (A == 1 || B == 1) return C.d2(i) / 6.0;
  const int dlo = (A >= SPL_BIG) ? -SPL_W : max(-SPL_W, 2 - A);
  const int dhi = (B >= SPL_BIG) ? SPL_W : min(SPL_W, B - 2);
  const double lam2 = lp[2];
  const double np1 = (A >= SPL_BIG || B >= SPL_BIG) ? 0.0 : lam_pow(lp, 2 * (A + B - 2));  // lam^(2(n+1))
  const double iden = 1.0 / ((1.0 - lam2) * (1.0 - np1)), ih2 = 6.0 / C.h2;   // no division inside the 69-term loop
  double ym = C.y(i + dlo - 1), y0 = C.y(i + dlo), yp;
  double acc = 0.0;
  for (int d = dlo; d <= dhi; ++d) {
    yp = C.y(i + d + 1);
    double rhs = ((ym - 2.0 * y0) + yp) * ih2;
    if (A < SPL_BIG && A + d == 2) rhs -= C.d2(i + d - 1) / 6.0;   // first inner row: - M_1
    if (B < SPL_BIG && B - d == 2) rhs -= C.d2(i + d + 1) / 6.0;   // last inner row:  - M_{m-2}
    const int ad = d < 0 ? -d : d;
    // min(i', j') = A-1+min(d,0);  n+1-max(i', j') = B-1-max(d,0)
    const double fa = (A >= SPL_BIG) ? 1.0 : 1.0 - lam_pow(lp, 2 * (A - 1 + (d < 0 ? d : 0)));
    const double fb = (B >= SPL_BIG) ? 1.0 : 1.0 - lam_pow(lp, 2 * (B - 1 - (d > 0 ? d : 0)));
    const double w = (lam_pow(lp, ad + 1) * fa) * (fb * iden);
    acc += (ad & 1) ? -w * rhs : w * rhs;
    ym = y0; y0 = yp;
  }
  return acc;
}

extern "C" __global__ void __launch_bounds__(256) eaqhm_spline_kernel(const double* __restrict__ records, int No_ti,
                                                                      int i0, int ni, int Kmax, int step,
                                                                      unsigned char* __restrict__ code,
                                                                      double* __restrict__ mom) {
  __shared__ double lp[80];
  if (threadIdx.x < 80) lp[threadIdx.x] = pow(2.0 - sqrt(3.0), (double)threadIdx.x);
  __syncthreads();
  const int ld = Kmax + 1;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)ni * ld) return;
  const int i = i0 + (int)(idx / ld), k = (int)(idx % ld);
  SplineCol C;
  C.rec = records; C.RS = 3 * Kmax + 1; C.No_ti = No_ti; C.all_acc = (k == Kmax);
  C.off_y = (k == Kmax) ? 3 * Kmax : Kmax + k;
  C.off_acc = (k == Kmax) ? 0 : k;
  C.h2 = (double)step * (double)step;
  double M = 0.0;
  unsigned char cd = 0;
  if (C.acc(i)) {
    // accepted neighbours on both sides as bit masks (independent loads, no dependent chain), then the run
    // lengths are the trailing ones of the masks
    unsigned long long ml = 0ull, mr = 0ull;
#pragma unroll 8
    for (int d = 0; d < SPL_CAP; ++d) {
      if (C.acc(i - d - 1)) ml |= 1ull << d;
      if (C.acc(i + d + 1)) mr |= 1ull << d;
    }
    const int dl = min(SPL_CAP, __ffsll((long long)~ml) - 1), dr = min(SPL_CAP, __ffsll((long long)~mr) - 1);
    const bool kl = dl < SPL_CAP, kr = dr < SPL_CAP;
    const int m = dl + dr + 1;
    if (kl && kr && m < 4) {
      cd = (m == 1) ? 1 : (unsigned char)(16 + 4 * m + dl);
    } else {
      cd = 2;
      const int A = kl ? dl : SPL_BIG, B = kr ? dr : SPL_BIG;
      // the first / last knot of a run takes 2 M_1 - M_2 from its neighbours' moments: eaqhm_spline_edge_kernel
      // (computing them here would make every wave that holds one edge knot run the 69-term sum three times)
      if (A != 0 && B != 0) M = inner_moment(C, lp, i, A, B);
    }
  }
  mom[(size_t)i * ld + k] = M;
  if (k < Kmax) code[(size_t)i * Kmax + k] = cd;
}

// not-a-knot end conditions: M_0 = 2 M_1 - M_2 and M_{m-1} = 2 M_{m-2} - M_{m-3} for runs of m >= 4 knots (code 2)
extern "C" __global__ void __launch_bounds__(256) eaqhm_spline_edge_kernel(const unsigned char* __restrict__ code, int No_ti,
                                                                           int i0, int ni, int Kmax, double* __restrict__ mom) {
  const int ld = Kmax + 1;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)ni * ld) return;
  const int i = i0 + (int)(idx / ld), k = (int)(idx % ld);
  auto acc = [&](int q) { return q >= 0 && q < No_ti && (k == Kmax || code[(size_t)q * Kmax + k] != 0); };
  if (!(k == Kmax || code[(size_t)i * Kmax + k] == 2)) return;
  const bool first = !acc(i - 1), last = !acc(i + 1);
  if (first == last) return;   // inner knot (or isolated: not code 2)
  const int s1 = first ? 1 : -1;
  mom[(size_t)i * ld + k] = 2.0 * mom[(size_t)(i + s1) * ld + k] - mom[(size_t)(i + 2 * s1) * ld + k];
}

// ------------------------------------------------------------------------------------------------
struct EvalArgs {
  const double* records; const unsigned char* code; const double* mom;
  int No_ti; int Kmax; int step; double fs; long long L; long long t_lo; long long t_hi; long long s_lo; long long s_hi;
  // am_out / fm_out: biased by -track_t0, rows of Lt samples (only samples [t_lo, t_hi) are written); NULL: no track
  // output.  s_hat NULL: no synthesis, no phase rows, no error sums (a pass that only regenerates tracks).
  const double* target; double* am_out; double* fm_out; long long Lt; double* ph_knot; double* s_hat; long long* partials;
};

// ---- error sums that do not depend on how the samples are grouped -------------------------------------------------
// SRER (functions.py:388) needs sum d and sum d^2 over the whole file, d = target - s_hat.  They are collected per
// block, per rank and (long files) per time block; a floating-point sum would make the last digits of the SRER — and
// with them the stop rule `SRER[a] <= SRER[a-1]` — depend on that grouping.  So every sample is turned into fixed
// point, three signed 64-bit limbs in base 2^32 (d * 2^60 and d^2 * 2^64, truncated: 2^-60 / 2^-64 absolute), and the
// limbs are added as integers: exact, associative, identical for every block size, world size and streaming block.
// |d| < 2^20 is required (anything larger, and NaN / Inf, is counted in limb 6 and reported by the host).
#define ES_LIMBS 8   // {d: l0, l1, l2,  d^2: l0, l1, l2,  non-finite or huge samples, -}
__device__ inline void fixed_limbs(double x, long long& l0, long long& l1, long long& l2) {
  // x already scaled by a power of two (exact); |x| < 2^104
  const double h2 = trunc(x * 0x1p-64);
  const double r1 = x - h2 * 0x1p64;          // exact: the low bits of x
  const double h1 = trunc(r1 * 0x1p-32);
  const double r0 = r1 - h1 * 0x1p32;         // exact
  l2 = (long long)h2; l1 = (long long)h1; l0 = (long long)trunc(r0);
}

// Rows [r0, r1] of records / code / mom staged in LDS by the block (eaqhm_eval_kernel); anything outside (only the
// padded <4-knot case reaches back to rows 0..3) is read from memory.
struct RowCache {
  const double* rec; const double* mom; const unsigned char* code; int r0, r1;
};

struct Slot {
  const EvalArgs& A;
  const RowCache& C;
  int k;
  __device__ bool in(int i) const { return i >= C.r0 && i <= C.r1; }
  __device__ double recv(int i, int col) const {
    const int RS = 3 * A.Kmax + 1;
    return in(i) ? C.rec[(size_t)(i - C.r0) * RS + col] : A.records[(size_t)i * RS + col];
  }
  __device__ double am(int i) const { return recv(i, k); }
  __device__ double fm(int i) const { return recv(i, A.Kmax + k); }
  __device__ double ph(int i) const { return recv(i, 2 * A.Kmax + k); }
  __device__ int code(int i) const {
    if (i < 0 || i >= A.No_ti) return 0;
    return in(i) ? C.code[(size_t)(i - C.r0) * A.Kmax + k] : A.code[(size_t)i * A.Kmax + k];
  }
  __device__ double mom(int i) const {
    return in(i) ? C.mom[(size_t)(i - C.r0) * (A.Kmax + 1) + k] : A.mom[(size_t)i * (A.Kmax + 1) + k];
  }
};

// cubic piece of the interval (i, i+1) at offset r samples from knot i
__device__ inline double spline_piece(double y0, double y1, double m0, double m1, double r, double h) {
  double u = r / h, v = 1.0 - u;
  return v * y0 + u * y1 + (h * h / 6.0) * ((v * v * v - v) * m0 + (u * u * u - u) * m1);
}

// fm_recon on the interval (i, i+1), both ends accepted, r in [0, step].  Evaluated 2(step+1) times per interval,
// so everything that does not depend on r is prepared once: no division and no array indexing per sample (a
// wave that holds a single short-run interval runs both branches for all of its lanes).
struct FmPiece {
  int kind;  // 2: spline piece, 3: single cubic through 4 points (short run, functions.py:368-371)
  double y0, y1, m0, m1, hinv, h2_6;
  double p0, p1, p2, p3, w0, w1, w2, w3, x0;  // nodes and weights y_p / prod_{q != p}(x_p - x_q) of the short-run cubic
  __device__ double operator()(int r) const {
    if (kind == 2) {
      const double u = (double)r * hinv, v = 1.0 - u;
      return v * y0 + u * y1 + h2_6 * ((v * v * v - v) * m0 + (u * u * u - u) * m1);
    }
    const double x = x0 + (double)r;
    const double d0 = x - p0, d1 = x - p1, d2 = x - p2, d3 = x - p3;
    return ((w0 * d1) * (d2 * d3) + (w1 * d0) * (d2 * d3)) + ((w2 * d3) * (d0 * d1) + (w3 * d2) * (d0 * d1));
  }
};

__device__ inline FmPiece make_piece(const Slot& S, int i, int ci) {
  FmPiece P;
  const int step = S.A.step;
  const double h = (double)step;
  P.hinv = 1.0 / h;
  P.h2_6 = h * h / 6.0;
  P.x0 = (double)i * h;
  P.y0 = P.y1 = P.m0 = P.m1 = 0.0;
  P.p0 = 0.0; P.p1 = 1.0; P.p2 = 2.0; P.p3 = 3.0; P.w0 = P.w1 = P.w2 = P.w3 = 0.0;
  if (ci == 2) {
    P.kind = 2;
    P.y0 = S.fm(i); P.y1 = S.fm(i + 1); P.m0 = S.mom(i); P.m1 = S.mom(i + 1);
  } else {
    P.kind = 3;
    const int m = (ci - 16) >> 2, pos = (ci - 16) & 3;
    const int rs = i - pos;
    const int npad = 4 - m;  // knots at samples 0, step, ... carry whatever fm_recon holds there
    double px[4], py[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const bool pad = p < npad;
      const int row = pad ? p : (rs + p - npad);
      px[p] = pad ? (double)(p * step) : (double)row * h;
      py[p] = (pad && !(p < S.A.No_ti && S.code(p) != 0)) ? 0.0 : S.fm(row);
    }
    P.p0 = px[0]; P.p1 = px[1]; P.p2 = px[2]; P.p3 = px[3];
    P.w0 = py[0] / ((px[0] - px[1]) * (px[0] - px[2]) * (px[0] - px[3]));
    P.w1 = py[1] / ((px[1] - px[0]) * (px[1] - px[2]) * (px[1] - px[3]));
    P.w2 = py[2] / ((px[2] - px[0]) * (px[2] - px[1]) * (px[2] - px[3]));
    P.w3 = py[3] / ((px[3] - px[0]) * (px[3] - px[1]) * (px[3] - px[2]));
  }
  return P;
}

// numpy.unwrap on one step: the unwrapped difference
__device__ inline double unwrap_diff(double dd) {
  if (fabs(dd) < M_PI) return dd;
  double m = fmod(dd + M_PI, 2.0 * M_PI);
  if (m < 0) m += 2.0 * M_PI;
  m -= M_PI;
  if (m == -M_PI && dd > 0) m = M_PI;
  return m;
}

// Block of TBS consecutive samples x all slots, two stages.
//   stage 1  one thread per (knot interval, slot) touching the block: phase_integr_interpolation
//            (functions.py:537-575) of the whole interval ONCE, in the reference's summation order — cumulative
//            sum of the instantaneous frequency, shifted to start at the analysed phase, minus the cumulative
//            sine bump that closes the phase error at the next knot.  The values at the block's samples go to
//            LDS:  X1[k][s] phase at sample s, X2[k][s] phase one sample earlier (same interval),
//            X3[k][knot] fm_recon at the first sample of the interval that starts at the knot.
//   stage 2  one thread per (sample, slot group): amplitudes, next-iteration frequency from the unwrapped phase
//            (functions.py:375), knot bookkeeping, am*cos(ph) into LDS; then one thread per sample adds the
//            slots in slot order, the a0 spline and the reconstruction error (functions.py:385-388).
extern "C" __global__ void __launch_bounds__(256) eaqhm_eval_kernel(EvalArgs A, int TBS, int NK, int NR) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int D = A.step, K = A.Kmax;
  double* ft = lds;                           // D+1
  double* X1 = lds + ((D + 1 + 1) & ~1);      // [K][TP]
  const int TP = TBS + 1;                     // row stride: stage 1 writes a column of slots at once (bank spread)
  double* X2 = X1 + (size_t)K * TP;           // [K][TP]
  double* X3 = X2 + (size_t)K * TP;           // [K][NK]
  double* crec = X3 + (size_t)K * NK;         // [NR][3K+1]   staged rows of the records
  double* cmom = crec + (size_t)NR * (3 * K + 1);                    // [NR][K+1]
  unsigned char* ccode = (unsigned char*)(cmom + (size_t)NR * (K + 1));   // [NR][K]
  const int tid = threadIdx.x;
  const long long t0 = A.t_lo + (long long)blockIdx.x * TBS;
  const long long t1 = (t0 + TBS < A.t_hi) ? (t0 + TBS) : A.t_hi;      // block covers [t0, t1)
  const long long tk0 = ((t0 + D - 1) / D) * D;                        // first knot at or after t0
  // the instants this block touches: one before its first interval to two after its last (coalesced rows)
  RowCache C;
  {
    const int jlo = (int)((t0 > 0) ? ((t0 - 1) / D) : 0), jhi = (int)((t1 - 1) / D);
    C.r0 = (jlo > 0) ? jlo - 1 : 0;
    C.r1 = (jhi + 2 < A.No_ti) ? jhi + 2 : A.No_ti - 1;
    if (C.r1 > C.r0 + NR - 1) C.r1 = C.r0 + NR - 1;
    if (C.r1 < C.r0) C.r1 = C.r0 - 1;   // nothing staged (block beyond the last instant)
    C.rec = crec; C.mom = cmom; C.code = ccode;
    const int nrow = C.r1 - C.r0 + 1, RS = 3 * K + 1;
    for (int q = tid; q < nrow * RS; q += blockDim.x) crec[q] = A.records[(size_t)C.r0 * RS + q];
    for (int q = tid; q < nrow * (K + 1); q += blockDim.x) cmom[q] = A.mom[(size_t)C.r0 * (K + 1) + q];
    for (int q = tid; q < nrow * K; q += blockDim.x) ccode[q] = A.code[(size_t)C.r0 * K + q];
  }
  for (int u = tid; u <= D; u += blockDim.x) ft[u] = sin(M_PI * (double)u / (double)D);
  __syncthreads();
  // ---- stage 1: intervals (j, j+1) whose samples j*D .. (j+1)*D meet the block
  {
    const int jlo = (int)((t0 > 0) ? ((t0 - 1) / D) : 0), jhi = (int)((t1 - 1) / D);
    const int nint = jhi - jlo + 1;
    const double scale = 2.0 * M_PI / A.fs;
    for (int p = tid; p < nint * K; p += blockDim.x) {
      const int jj = p / K, k = p - jj * K, j = jlo + jj;
      if (j + 1 >= A.No_ti) continue;
      Slot S{A, C, k};
      const int cj = S.code(j);
      if (cj == 0 || S.code(j + 1) == 0) continue;
      const FmPiece P = make_piece(S, j, cj);
      const double p0 = P(0), w0 = scale * p0;
      double acc = w0;
      for (int u = 1; u <= D; ++u) acc += scale * P(u);
      const double shift = S.ph(j) - w0;
      const double e = (acc + shift) - S.ph(j + 1);
      const double Mr = rint(e / (2.0 * M_PI));
      const double er = M_PI * (e - 2.0 * M_PI * Mr) / (2.0 * (double)D);
      const long long tb = (long long)j * D;
      if (tb >= t0 && tb < t1) X3[(size_t)k * NK + (int)((tb - tk0) / D)] = p0;
      acc = w0;
      double c = ft[0] * er;
      double prev = (acc + shift) - c;
      for (int u = 1; u <= D; ++u) {
        acc += scale * P(u);
        c += ft[u] * er;
        const double ph = (acc + shift) - c;
        const long long t = tb + u;
        if (t >= t0 && t < t1) {
          X1[(size_t)k * TP + (int)(t - t0)] = ph;
          X2[(size_t)k * TP + (int)(t - t0)] = prev;
        }
        prev = ph;
      }
    }
  }
  __syncthreads();
  // ---- stage 2
  const int s = tid % TBS, g = tid / TBS, G = blockDim.x / TBS;
  const long long t = t0 + s;
  const bool live = t < A.t_hi;
  int i = 0, r = 0;
  bool past = false;
  if (live) {
    i = (int)(t / D);
    r = (int)(t - (long long)i * D);
    if (i >= A.No_ti) { i = A.No_ti - 1; r = (int)(t - (long long)i * D); }  // beyond the last instant
    past = (i == A.No_ti - 1) && (r > 0);
    for (int k = g; k < K; k += G) {
      Slot S{A, C, k};
      double amv = 0.0, phv = 0.0, fnext = 0.0;
      const int ci = S.code(i);
      if (!past && r > 0) {
        const int cn = S.code(i + 1);
        if (ci != 0 && cn != 0) {  // inside the active interval (i, i+1)
          const double x0 = (double)i * (double)D, x1 = (double)(i + 1) * (double)D;
          const double a0v = S.am(i), a1v = S.am(i + 1);
          amv = ((a1v - a0v) / (x1 - x0)) * ((double)t - x0) + a0v;
          const double pr = X1[(size_t)k * TP + s], pm = X2[(size_t)k * TP + s];
          phv = pr;
          fnext = A.fs / (2.0 * M_PI) * unwrap_diff(pr - pm);
        }
      } else if (r == 0) {
        if (ci != 0) {  // on a knot
          amv = S.am(i);
          const bool prev = S.code(i - 1) != 0, next = S.code(i + 1) != 0;
          if (!prev && !next) {
            phv = S.ph(i);  // isolated accepted instant: frame-centre values stay as written
          } else {
            double pD = 0.0, pDm1 = 0.0;
            if (prev) { pD = X1[(size_t)k * TP + s]; pDm1 = X2[(size_t)k * TP + s]; }
            if (next) {
              const double p0 = X3[(size_t)k * NK + (int)((t - tk0) / D)];
              const double w0 = (2.0 * M_PI / A.fs) * p0;
              phv = w0 + (S.ph(i) - w0);  // first sample of the next interval overwrites the knot
              if (!prev) fnext = p0;      // first sample of the run keeps fm_recon (functions.py:375)
            } else {
              phv = pD;                   // last knot of a run keeps the integrated phase
            }
            if (prev) fnext = A.fs / (2.0 * M_PI) * unwrap_diff(phv - pDm1);
          }
          if (A.s_hat) A.ph_knot[(size_t)i * K + k] = phv;
        } else {
          if (A.s_hat) A.ph_knot[(size_t)i * K + k] = 0.0;
        }
      }
      if (A.am_out) {
        A.am_out[(size_t)k * A.Lt + t] = amv;
        A.fm_out[(size_t)k * A.Lt + t] = fnext;
      }
      if (A.s_hat) X1[(size_t)k * TP + s] = (amv != 0.0) ? amv * cos(phv) : 0.0;   // same thread read this cell above
    }
  }
  if (!A.s_hat) return;   // (uniform over the grid: a track-only pass)
  __syncthreads();
  long long e[ES_LIMBS - 1] = {0, 0, 0, 0, 0, 0, 0};
  if (tid < TBS && live) {
    double synth = 0.0;
#pragma unroll 8
    for (int k = 0; k < K; ++k) synth += X1[(size_t)k * TP + s];   // slot order; the loads of eight slots in flight
    // a0: not-a-knot spline through every instant, extrapolated past the last one (functions.py:340)
    int ia = i;
    if (ia > A.No_ti - 2) ia = A.No_ti - 2;
    const size_t RS = 3 * (size_t)K + 1;
    Slot S0{A, C, K};   // column K of mom = the a0 spline; its knots are the last record column
    double a0v = spline_piece(S0.recv(ia, (int)RS - 1), S0.recv(ia + 1, (int)RS - 1), S0.mom(ia), S0.mom(ia + 1),
                              (double)(t - (long long)ia * D), (double)D);
    const double sh = a0v + 2.0 * synth;
    A.s_hat[t] = sh;
    if (t >= A.s_lo && t < A.s_hi) {
      const double d = A.target[t] - sh;
      if (fabs(d) < 0x1p20) {   // (false for NaN)
        fixed_limbs(d * 0x1p60, e[0], e[1], e[2]);
        fixed_limbs((d * d) * 0x1p64, e[3], e[4], e[5]);
      } else {
        e[6] = 1;
      }
    }
  }
  if (tid < 64) {   // TBS <= 64: the block's samples sit in the first wave
#pragma unroll
    for (int q = 0; q < ES_LIMBS - 1; ++q) {
      for (int o = 32; o > 0; o >>= 1) e[q] += __shfl_xor(e[q], o);
      if (tid == 0) A.partials[(size_t)q * gridDim.x + blockIdx.x] = e[q];   // [limb][block]: the reduction reads coalesced
    }
  }
}

// Adds the blocks' limbs (integers: any order gives the same result), carries them into two 128-bit totals and leaves
//   sums_out[0..3]  sum d, sum d^2, n, SRER in dB — doubles, a convenience for C callers; the Python host derives the
//                   SRER itself from the limbs (one formula for every world size and block count)
//   sums_out[4..6]  LS breakdowns / stalled diagonal pipelines / dropped frames since the last read (counters are cleared)
//   sums_out[8..15] the limbs as int64 bit patterns: what ranks and time blocks add up
extern "C" __global__ void __launch_bounds__(1024) eaqhm_srer_kernel(const long long* partials, long long nblocks, double n,
                                                                    double std_det, double* sums_out, int* faults) {
  __shared__ long long red[16][ES_LIMBS];
  long long e[ES_LIMBS - 1] = {0, 0, 0, 0, 0, 0, 0};
  for (long long b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
    for (int q = 0; q < ES_LIMBS - 1; ++q) e[q] += partials[(size_t)q * nblocks + b];
#pragma unroll
  for (int q = 0; q < ES_LIMBS - 1; ++q) {
    for (int o = 32; o > 0; o >>= 1) e[q] += __shfl_xor(e[q], o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = e[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long* lim = (long long*)(sums_out + 8);
    for (int q = 0; q < ES_LIMBS - 1; ++q) {
      e[q] = 0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) e[q] += red[w][q];
      lim[q] = e[q];
    }
    lim[ES_LIMBS - 1] = 0;
    const double a = ((double)e[2] * 0x1p64 + (double)e[1] * 0x1p32 + (double)e[0]) * 0x1p-60;
    const double b = ((double)e[5] * 0x1p64 + (double)e[4] * 0x1p32 + (double)e[3]) * 0x1p-64;
    const double mean = a / n;
    const double var = b / n - mean * mean;
    sums_out[0] = a; sums_out[1] = b; sums_out[2] = n;
    sums_out[3] = e[6] ? __builtin_nan("") : 20.0 * log10(std_det / sqrt(var));
    sums_out[4] = (double)faults[0];   // LS systems whose factorisation broke down in this adaptation (eaqhm_ls_faults)
    sums_out[5] = (double)faults[1];   // diagonal pipelines that timed out (a bug of the library if ever nonzero)
    sums_out[6] = (double)faults[2];   // frames dropped because their window lay outside the resident track window
    faults[0] = 0; faults[1] = 0; faults[2] = 0;
  }
}


// ------------------------------------------------------------------------------------------------
// The third inner seam as a stand-alone entry: phase_integr_interpolation(fm_recon, ph_recon, indices)
// (functions.py:537-575) for arbitrary knot spacing.  One thread per sample of [knots[0], knots[m-1]]; the
// sample's interval is found by binary search; each interval is integrated in the reference's summation
// order (cumulative sum of the instantaneous frequency, shifted to start at the analysed phase, minus the
// cumulative sine bump that closes the phase error at the next knot).  The shared knot of two intervals takes
// the value of the LATER interval (its first sample), the very last knot keeps the integrated value.
extern "C" __global__ void eaqhm_phase_integrate_kernel(const double* __restrict__ omega, const double* __restrict__ ph,
                                                        const int* __restrict__ knots, int m, double* __restrict__ out) {
  const int first = knots[0], last = knots[m - 1];
  const int t = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (t > last) return;
  int lo = 0, hi = m - 1;  // interval i with knots[i] <= t < knots[i+1]  (t == last -> i = m-2)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (knots[mid] <= t) lo = mid; else hi = mid;
  }
  const int i0 = knots[lo], i1 = knots[lo + 1], D = i1 - i0, r = t - i0;
  double acc = omega[i0], sr = acc;
  for (int u = 1; u <= D; ++u) {
    acc += omega[i0 + u];
    if (u == r) sr = acc;
  }
  const double shift = ph[i0] - omega[i0];
  const double e = (acc + shift) - ph[i1];
  const double Mr = rint(e / (2.0 * M_PI));
  const double er = M_PI * (e - 2.0 * M_PI * Mr) / (2.0 * (double)D);
  double c = 0.0, cr = 0.0;
  for (int u = 0; u <= D; ++u) {
    c += sin(M_PI * (double)u / (double)D) * er;
    if (u == r) cr = c;
  }
  out[t - first] = (sr + shift) - cr;
}
}  // namespace eaqhm

using namespace eaqhm;

extern "C" int eaqhm_spline_solve_range(eaqhm_ctx* ctx, const double* records, int32_t No_ti, int32_t Kmax, int32_t step,
                                        int32_t i_lo, int32_t i_hi, uint8_t* code, double* mom) {
  if (!ctx) return EAQHM_EINVAL;
  if (!records || !code || !mom || Kmax <= 0 || step <= 0 || i_lo < 0 || i_hi > No_ti || i_lo >= i_hi)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_spline_solve: bad argument");
  if (No_ti < 4) return ctx->fail(EAQHM_EINVAL, "eaqhm_spline_solve: need at least 4 analysis instants (interp1d kind=3)");
  const int ld = Kmax + 1;
  auto blocks = [&](int n) { return dim3((unsigned)(((long long)n * ld + 255) / 256)); };
  // moments two instants beyond the range feed the end conditions of runs that start / stop inside it
  const int a = (i_lo - 2 > 0) ? i_lo - 2 : 0, b = (i_hi + 2 < No_ti) ? i_hi + 2 : No_ti;
  hipLaunchKernelGGL(eaqhm_spline_kernel, blocks(b - a), dim3(256), 0, ctx->stream, records, No_ti, a, b - a, Kmax, step,
                     code, mom);
  HIP_TRY(ctx, hipGetLastError());
  if (a > 0) {   // the padded <4-knot case looks at the run codes of instants 0..3 wherever it is evaluated
    const int n0 = (a < 4) ? a : 4;
    hipLaunchKernelGGL(eaqhm_spline_kernel, blocks(n0), dim3(256), 0, ctx->stream, records, No_ti, 0, n0, Kmax, step, code, mom);
    HIP_TRY(ctx, hipGetLastError());
  }
  hipLaunchKernelGGL(eaqhm_spline_edge_kernel, blocks(i_hi - i_lo), dim3(256), 0, ctx->stream, code, No_ti, i_lo,
                     i_hi - i_lo, Kmax, mom);
  HIP_TRY(ctx, hipGetLastError());
  return EAQHM_OK;
}

extern "C" int eaqhm_spline_solve(eaqhm_ctx* ctx, const double* records, int32_t No_ti,
                                  int32_t Kmax, int32_t step, uint8_t* code, double* mom) {
  return eaqhm_spline_solve_range(ctx, records, No_ti, Kmax, step, 0, No_ti, code, mom);
}

// samples per block of eaqhm_eval_kernel: the largest of 64/32/16 whose LDS tables fit
static int eval_block_samples(int Kmax, int step, size_t* lds_bytes, int* nk, int* nr) {
  for (int tbs = 64; tbs >= 16; tbs >>= 1) {
    const int NK = tbs / step + 2, NR = tbs / step + 5;   // staged instants: intervals of the block, one before, two after
    const size_t bytes = (((size_t)step + 2) & ~(size_t)1) * 8 + ((size_t)2 * Kmax * (tbs + 1) + (size_t)Kmax * NK) * 8 +
                         (size_t)NR * ((3 * (size_t)Kmax + 1) + (Kmax + 1)) * 8 + (((size_t)NR * Kmax + 7) & ~(size_t)7);
    if (bytes <= 78 * 1024 || tbs == 16) {
      *lds_bytes = bytes; *nk = NK; *nr = NR;
      return tbs;
    }
  }
  return 16;
}

extern "C" int64_t eaqhm_eval_partials_len(int64_t t_lo, int64_t t_hi, int32_t step) {
  (void)step;
  if (t_hi <= t_lo) return ES_LIMBS;
  return ES_LIMBS * ((t_hi - t_lo + 15) / 16);   // eight 8-byte words per block of >= 16 samples
}

extern "C" int eaqhm_eval_synth(eaqhm_ctx* ctx, const double* records, const uint8_t* code,
                                const double* mom, int32_t No_ti, int32_t Kmax, int32_t step, double fs, int64_t L,
                                int64_t t_lo, int64_t t_hi, int64_t s_lo, int64_t s_hi, const double* target,
                                double std_det, double* am_out, double* fm_out, int64_t track_t0, int64_t track_len,
                                double* ph_knot, double* s_hat, double* partials, double* sums_out) {
  if (!ctx) return EAQHM_EINVAL;
  const bool tracks = am_out != nullptr, synth = s_hat != nullptr;
  if (!records || !code || !mom || (!tracks && !synth) || (tracks && !fm_out) || No_ti < 4 || Kmax <= 0 || step <= 0 ||
      fs <= 0 || L <= 0 || t_lo < 0 || t_hi > L || t_lo >= t_hi)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_eval_synth: bad argument");
  if (synth && (!target || !ph_knot || !partials || !sums_out || s_lo < t_lo || s_hi > t_hi || s_lo >= s_hi))
    return ctx->fail(EAQHM_EINVAL, "eaqhm_eval_synth: bad argument (synthesis outputs / error range)");
  if (tracks && (track_t0 < 0 || track_len <= 0 || t_lo < track_t0 || t_hi > track_t0 + track_len))
    return ctx->fail(EAQHM_EINVAL, "eaqhm_eval_synth: [t_lo, t_hi) outside the track window");
  if ((int64_t)(No_ti - 1) * step >= L) return ctx->fail(EAQHM_EINVAL, "eaqhm_eval_synth: instants beyond the signal");
  EvalArgs A{records, code, mom, No_ti, Kmax, step, fs, (long long)L, (long long)t_lo, (long long)t_hi,
             (long long)s_lo, (long long)s_hi, target, tracks ? am_out - track_t0 : nullptr,
             tracks ? fm_out - track_t0 : nullptr, (long long)track_len, ph_knot, s_hat, (long long*)partials};
  size_t lds_bytes = 0;
  int NK = 0, NR = 0;
  const int TBS = eval_block_samples(Kmax, step, &lds_bytes, &NK, &NR);
  if (lds_bytes > 160 * 1024) return ctx->fail(EAQHM_EINVAL, "eaqhm_eval_synth: Kmax too large for the LDS tables");
  const long long nblocks = (t_hi - t_lo + TBS - 1) / TBS;
  HIP_TRY(ctx, hipFuncSetAttribute((const void*)eaqhm_eval_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(eaqhm_eval_kernel, dim3((unsigned)nblocks), dim3(256), lds_bytes, ctx->stream, A, TBS, NK, NR);
  HIP_TRY(ctx, hipGetLastError());
  if (!synth) return EAQHM_OK;
  hipLaunchKernelGGL(eaqhm_srer_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const long long*)partials, nblocks,
                     (double)(s_hi - s_lo), std_det, sums_out, ctx->faults);
  HIP_TRY(ctx, hipGetLastError());
  return EAQHM_OK;
}

extern "C" int eaqhm_phase_integrate(eaqhm_ctx* ctx, const double* omega, const double* ph, const int32_t* knots,
                                     int32_t n_knots, int32_t first, int32_t last, double* out) {
  if (!ctx) return EAQHM_EINVAL;
  if (!omega || !ph || !knots || !out || n_knots < 2 || last <= first)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_phase_integrate: bad argument");
  const int n = last - first + 1;
  hipLaunchKernelGGL(eaqhm_phase_integrate_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, omega, ph, knots,
                     n_knots, out);
  HIP_TRY(ctx, hipGetLastError());
  return EAQHM_OK;
}
