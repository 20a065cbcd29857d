// eaqhm_ls_mfma.hip — the per-frame LS with the Gramian on the FP64 matrix cores (gfx950).
//
// One workgroup of 512 threads (8 waves, 2 per SIMD) owns one frame at a time; frames are pulled from an
// atomic queue (frames differ in size, so static striding leaves tails).
//
//   A1   per-slot set-up (prepare_slots, eaqhm_ls_common.h): a slot without a zero inside the window needs only its
//        centre values, one with gaps gets its window bridged into the workgroup's scratch rows
//   A3+B time is cut into chunks of up to 16 sample PAIRS (mid-d-1, mid+d) taken from the window centre outwards: the
//        pair shares its sincos work because the negative-frequency column at u is the time-reversed positive one
//        (functions.py:284-285), and the phase integral relative to the centre (functions.py:508-515) is a running
//        sum: the 16 lanes of a slot load their fm / am straight from the [Kmax][L] tracks, a 16-lane DPP scan plus a
//        per-slot carry gives the phase.  All threads build the basis rows of a chunk in LDS (planar re/im,
//        [row][column], row stride ≡ 16 (mod 32) doubles and a per-pair rotation of the columns inside each block of
//        16: transposing writes and MFMA operand reads are both conflict-free); then every wave runs
//        v_mfma_f64_16x16x4_f64 on its tiles:  G_p[I][J] += X_I^H diag(w^2 n^p) X_J,  p = 0,1,2, three real products
//        per complex one, operands read once for the three weights.  The accumulators of a pass stay in registers;
//        a frame takes ceil(tiles / 16) passes over its window.  The signal window is one more basis column, so the
//        right-hand sides are tiles of the same contraction.
//   C    system matrix [[G0,G1],[G1,G2]] + RHS row -> scratch, Cholesky + back substitution
//   D    frequency mismatch, acceptance, record row                                     (eaqhm_ls_common.h)
#include "eaqhm_ls_common.h"
#include "eaqhm_ls_chol.h"
#include "eaqhm_ls_a0.h"

namespace eaqhm {

#define MF_THREADS 512
#define MF_WAVES 8
#define MF_A0_M 19   // adaptation 0 on chip: real systems of up to 19 tile rows (Kc + 1 <= 304)
#define MF_A0_NCH 4
#define MF_CI 8      // doubles of per-slot info: carries, window pointers, 1/(am_mid+eps), rho (prepare_slots)
// workgroup of eaqhm_ls_mfma_kernel: MFK_WAVES waves with MF_NT base Gramian tiles per wave and pass, each tile with its
// three weights (9 accumulators of 8 VGPRs) — 16 tiles per pass either way
#ifndef MFK_WAVES
#define MFK_WAVES 8
#endif
#define MFK_THREADS (64 * MFK_WAVES)
// MF_FOURPROD: four real products per complex one on TWO accumulators per (tile, weight) instead of three products on
// three — 12 MFMAs per tile and k-step instead of 9, 48 accumulator registers per tile instead of 72 (experiment: twelve
// waves of 168 registers with two tiles each, 24 tiles per pass)
#if MFK_WAVES == 12
#define MF_FOURPROD 1
#endif
#ifndef MF_NT
#define MF_NT ((MFK_WAVES == 12) ? 2 : (16 / MFK_WAVES))
#endif

struct MfScratch {
  double* Q;   // Npad * nmax   bridged frequency windows
  double* r;   // Npad * nmax   bridged amplitude windows
  double* T;   // stacked padded system, 16x16 complex tiles (eaqhm_ls_chol.h)
  double* WT;  // inverse diagonal tiles
  double* D0;  // original diagonal of the system (16 per tile row)
};

__host__ __device__ inline size_t mf_tile_doubles(int Kcmax) {
  const size_t nt = 2 * (((size_t)Kcmax + 15) / 16) + 1;
  return nt * (nt + 1) / 2 * 512;
}
__host__ __device__ inline size_t mf_scratch_doubles(int nmax, int Nmax, int Kcmax) {
  const size_t nt = 2 * (((size_t)Kcmax + 15) / 16) + 1;
  return 2 * (size_t)(((Nmax + 63) >> 6) << 6) * nmax + mf_tile_doubles(Kcmax) + nt * 2 * TL_TILE + nt * 16;
}

__device__ inline void tile_of(int q, int& I, int& J) {
  I = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
  while ((I + 1) * (I + 2) / 2 <= q) ++I;
  while (I * (I + 1) / 2 > q) --I;
  J = q - I * (I + 1) / 2;
}

extern "C" __global__ void __launch_bounds__(MFK_THREADS) eaqhm_ls_mfma_kernel(LsArgs A, int plane, int min_nb, int a0_onchip) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (min_nb > 0 && A.cls[LS_BIG_CLASS] == 0) return;   // nothing left over by eaqhm_ls_tile_kernel (uniform across the grid)
  const int tid = threadIdx.x, nt = MFK_THREADS;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the unit bookkeeping in SGPRs
  const int Mmax = 2 * A.Kcmax;
  const int Npad = ((A.Nmax + 63) >> 6) << 6, nchs = Npad >> 6;
  double* const Xbase = lds;
  double* Wp = lds + (size_t)2 * plane;          // 3 * 32  weights w^2 n^p per chunk row   (the chunk planes come first)
  double* ci = Wp + 96;                      // nmax * MF_CI  per-slot info (prepare_slots)
  double* xs = ci + (size_t)A.nmax * MF_CI;  // 2 * Mmax
  double* sh = xs + 2 * Mmax;                // 16
  int* shi = (int*)(sh + 12);
  unsigned long long* masks = (unsigned long long*)lds;           // [nmax][nchs]  (over the planes, set-up only)
  int* gappy = (int*)(masks + (size_t)A.nmax * nchs);             // [nmax]
  MfScratch S;
  {
    double* base = A.scratch + (size_t)blockIdx.x * A.scratch_stride;
    S.Q = base;                                  // bridged fm[j][t] of the slots with gaps
    S.r = S.Q + (size_t)Npad * A.nmax;           // bridged am[j][t]
    S.T = S.r + (size_t)Npad * A.nmax;
    S.WT = S.T + mf_tile_doubles(A.Kcmax);
    S.D0 = S.WT + (2 * (((size_t)A.Kcmax + 15) / 16) + 1) * 2 * TL_TILE;
  }
  const bool seeds = (A.mode == 1) && A.any_seed && (*A.any_seed != 0);
  // phase stamps of thread 0 (diagnostic build -DEAQHM_EXPERIMENT_STAMPS only, see eaqhm_ls_chol.h; tools/phase_probe_big.py)
#ifndef EAQHM_EXPERIMENT_STAMPS
  unsigned long long* const dbg = nullptr;
#define MF_STAMP(ph) do { } while (0)
#define MF_STAMP_START() do { } while (0)
#else
  unsigned long long* dbg = uni(A.debug);
  unsigned long long t_prev = 0;
#define MF_STAMP_START() do { if (dbg && tid == 0) t_prev = __builtin_amdgcn_s_memtime(); } while (0)
#define MF_STAMP(ph)                                              \
  do {                                                            \
    if (dbg && tid == 0) {                                        \
      const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
      atomicAdd(dbg + (ph), t_now - t_prev);                      \
      t_prev = t_now;                                             \
    }                                                             \
  } while (0)
#endif

  // after eaqhm_ls_tile_kernel (min_nb > 0) only the frames of the last size class are left, usually none
  const int n_items = (min_nb > 0) ? A.cls[LS_BIG_CLASS] : A.n_frames;
  for (;;) {
    if (tid == 0) shi[0] = atomicAdd((min_nb > 0) ? (A.cls + 8 + LS_BIG_CLASS) : A.work_counter, 1);
    __syncthreads();
    const int item = uni(shi[0]);
    __syncthreads();
    if (item >= n_items) break;
    // (wave-uniform, but out of LDS / memory: scalar registers for everything derived from them)
    const int f = uni((min_nb > 0) ? A.cls[16 + (size_t)LS_BIG_CLASS * A.n_frames + item] : item);
    // (adaptation 0: frames of up to MF_A0_M real tile rows were solved on chip by eaqhm_ls_a0big_kernel)
    if (A.mode == 0 && a0_onchip && ((2 * uni(A.frame_K[f]) + 2 + 15) >> 4) <= MF_A0_M) continue;
    const int c = uni(A.frame_c[f]), wl = uni(A.frame_wl[f]), inst = uni(A.frame_inst[f]);
    if (min_nb == 0 && !frame_window_ok(A, c, wl)) continue;   // (counted by the classification kernel; class lists never hold such a frame)
    const int N = 2 * wl + 1, mid = wl;
    MF_STAMP_START();
    const int n = uni((A.mode == 0) ? A.frame_K[f] : A.ncol[f]);
    const int Kc = 2 * n + 1, C1 = Kc + 1;
    const int nbk = (Kc + 15) >> 4, ntl = 2 * nbk + 1;   // stacked padded tile rows (eaqhm_ls_chol.h)
    const int nb = (C1 + 15) >> 4, C1p = nb << 4;
    const int ldx = C1p + ((nb & 1) ? 0 : 16);  // ≡ 16 (mod 32)
    const int ntiles = nb * (nb + 1) / 2;
    const int npass = (ntiles + MFK_WAVES * MF_NT - 1) / (MFK_WAVES * MF_NT);
    const double f0 = uni((A.mode == 0) ? A.frame_f0[f] : A.f0_stale);
    const int* mycols = (A.mode == 1) ? (A.cols + (size_t)f * A.Kmax) : nullptr;
    const int npairs = mid + 1;  // sample pairs (u, v) = (mid-d-1, mid+d), d = 0..mid (u = -1: the virtual sample before the window)
    // rows per chunk: what the planes hold at this frame's row stride (16 pairs for the usual frame sizes)
    int TSf = (plane / ldx) & ~7;
    TSf = (TSf > 32) ? 32 : TSf;
    const int PE = TSf >> 1;
    double* Xre = Xbase;
    double* Xim = Xbase + (size_t)TSf * ldx;

    if (A.mode == 1)
      prepare_slots<MF_CI, MFK_WAVES>(A, S.Q, S.r, Npad, ci, masks, gappy, mycols, n, N, mid, c, wl, seeds, lane, wave, nchs);
    // the factorisation and the slot set-up used the planes as work space: finite everywhere, padding columns zero
    for (int q = tid; q < 2 * plane; q += nt) Xbase[q] = 0.0;
    // right-hand-side tile row: zero, its diagonal tile the identity (row 0 is filled from the signal row below)
    for (int q = tid; q < ntl * 512; q += nt) {
      const int Qt = q >> 9, e = q & 511;
      S.T[tile_off(ntl - 1, Qt) + e] = (Qt == ntl - 1 && e < 256 && (e >> 4) == (e & 15)) ? 1.0 : 0.0;
    }
    __syncthreads();
    MF_STAMP(0);

    if (A.mode == 0) {
      // ---- adaptation 0: the stacked padded system straight from the Toeplitz tables (eaqhm_ls_common.h): no basis,
      //      no contraction passes.  Work space: the chunk planes.
      const int TB = A.Kcmax + 1, NCH = 2;
      double* tab = Xbase;                                 // [TZ_NQ][TB]
      double* part = tab + (size_t)TZ_NQ * TB;             // [NCH][TZ_NQ][TB]
      double* W2 = part + (size_t)NCH * TZ_NQ * TB;        // [wl+1] each
      const int wpad = ((A.Nmax >> 1) + 8) & ~7;
      double* PA = W2 + wpad;
      double* PB = PA + wpad;
      double* win = PB + wpad;                             // [N]
      double* sig = win + ((A.Nmax + 7) & ~7);             // [N]
      for (int t = tid; t < N; t += nt) {
        win[t] = window_value(1, t, N);
        sig[t] = A.s[(size_t)(c - wl) + t];
      }
      __syncthreads();
      toeplitz_tables(tab, part, W2, PA, PB, sh, win, sig, n, wl, f0 * (2.0 * M_PI / A.fs), tid, TB, NCH);
      const double ssq = sh[0];
      const int ntiles_s = ntl * (ntl + 1) / 2;
      for (int q = tid; q < ntiles_s * 256; q += nt) {
        const int x = q >> 8, e = q & 255, row = e >> 4, col = e & 15;
        int P, Q;
        tile_of(x, P, Q);
        double re = 0.0, im = 0.0;
        if (P == ntl - 1) {                  // right-hand-side tile row: row 0 holds conj(rhs), the rest is padding
          if (Q == ntl - 1) re = (row == col) ? ((row == 0) ? ssq : 1.0) : 0.0;
          else if (row == 0) {
            const int pb = (Q >= nbk) ? 1 : 0, b = 16 * (Q - pb * nbk) + col;
            if (b < Kc) toeplitz_rhs(tab, TB, pb, b, n, re, im);
          }
        } else {
          const int pa = (P >= nbk) ? 1 : 0, pb = (Q >= nbk) ? 1 : 0;
          const int a = 16 * (P - pa * nbk) + row, b = 16 * (Q - pb * nbk) + col;
          if (a < Kc && b < Kc) toeplitz_gram(tab, TB, pa + pb, a, b, n, re, im);
          else if (P == Q && row == col) re = 1.0;          // identity padding inside the two diagonal blocks
        }
        double* t = S.T + tile_off(P, Q) + e;
        t[0] = re; t[256] = im;
      }
    } else
    for (int pass = 0; pass < npass; ++pass) {
      // every slot is one 16x16 tile of the base Gramian with all three weights: the operands are read once per k-step
      // for the three products, and every complex product is three real ones (P1 = re re, P2 = im im,
      // P3 = (re + im)(im' - re'):  Re = P1 + P2,  Im = P3 + P1 - P2): 9 MFMAs per tile and k-step instead of 12
#ifdef MF_FOURPROD
      d4 P1[MF_NT][3], P3[MF_NT][3];          // Re, Im
#else
      d4 P1[MF_NT][3], P2[MF_NT][3], P3[MF_NT][3];
#endif
      int tI[MF_NT], tJ[MF_NT];
      bool live[MF_NT];
#pragma unroll
      for (int sl = 0; sl < MF_NT; ++sl) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          P1[sl][w] = (d4){0, 0, 0, 0};
#ifndef MF_FOURPROD
          P2[sl][w] = (d4){0, 0, 0, 0};
#endif
          P3[sl][w] = (d4){0, 0, 0, 0};
        }
        const int x = (pass * MF_NT + sl) * MFK_WAVES + wave;
        live[sl] = x < ntiles;
        int I = 0, J = 0;
        tile_of(live[sl] ? x : 0, I, J);
        tI[sl] = I; tJ[sl] = J;
      }

      // The two columns of slot j sit next to each other (2j, 2j+1), so the tiles of this pass — lower-triangle tiles in
      // row-major order up to tile row Ihi — only read the columns of the slots j < 8 (Ihi + 1): the others are not built
      // (the passes of the first tile rows build a fraction of the basis; a fifth of the build over a frame)
      int jmax;
      {
        const int xl = (((pass + 1) * MF_NT * MFK_WAVES < ntiles) ? ((pass + 1) * MF_NT * MFK_WAVES) : ntiles) - 1;
        int Ihi, Jhi;
        tile_of(xl, Ihi, Jhi);
        jmax = uni((8 * (Ihi + 1) < n) ? (8 * (Ihi + 1)) : n);
      }
      if (pass > 0) {   // the running sums start again at the window centre
        for (int j = tid; j < n; j += nt) { ci[j * MF_CI] = 0.0; ci[j * MF_CI + 1] = 0.0; }
        __syncthreads();
      }
      const double w1 = 2.0 * M_PI / A.fs;   // phases are q * (2 pi / fs) (<= 2 ulp from (2 pi q) / fs of functions.py:513)
      for (int d0 = 0; d0 < npairs; d0 += PE) {
        // ---- build the chunk: rows 2*el (sample u = mid-d-1) and 2*el+1 (sample v = mid+d), d = d0 + el; logical
        //      column cc of those two rows lives at XCOL(cc, el)
#ifndef EAQHM_EXPERIMENT_NOBUILD   /* (timing experiment: stale basis rows; wrong results) */
#pragma clang loop unroll(disable)
        for (int idx = tid; idx < 16 * jmax; idx += nt) {
          const int el = idx & 15, j = idx >> 4, d = d0 + el;
          const bool act = (el < PE) && (d <= mid);
          const int u = mid - d - 1;
          double* cj = ci + j * MF_CI;
          const double* fb = ((const double**)cj)[2];   // bridged copy or the track itself (prepare_slots)
          const double* ab = ((const double**)cj)[3];
          // every load of the item up front, indices clamped into the window (results of clamped ones unused)
          const int dc = act ? d : mid;
          const double fu1 = fb[mid - dc], fv = fb[mid + dc];
          const double au = ab[(mid - dc - 1 >= 0) ? (mid - dc - 1) : 0], au1 = ab[mid - dc];
          const double av = ab[mid + dc], av1 = ab[(mid + dc + 1 < N) ? (mid + dc + 1) : (N - 1)];
          __builtin_amdgcn_sched_barrier(0);   // all six requests in flight before anything waits on one
          // F(v) - F(mid) = sum of fm over (mid, v];  F(u) - F(mid) = -sum over [u+1, mid]
          const double xu = act ? fu1 : 0.0, xv = (act && d >= 1) ? fv : 0.0;
          const double qv = cj[0] + scan16(xv), qu = -(cj[1] + scan16(xu));
          if (el == 15) { cj[0] = qv; cj[1] = -qu; }   // carried to the next chunk (read again after two barriers)
          if (!act) continue;
          const double ainv = cj[MF_CI - 3], pr = cj[MF_CI - 2], pi = cj[MF_CI - 1];
          double su, cu, sv, cv;
          sincos_cw(qu * w1, &su, &cu);
          sincos_cw(qv * w1, &sv, &cv);
          const double eps = 10e-5;
          double* xr = Xre + (2 * el) * ldx;
          double* xi = Xim + (2 * el) * ldx;
          const int cpos = XCOL(2 * j + 1, el), cneg = XCOL(2 * j, el);   // (column order: [neg 0, pos 0, neg 1, pos 1, ..., DC, signal])
          // positive column at t uses E1(t); negative column at t uses ratio[mirror+1] * E1(mirror) * rho
          if (u >= 0) {
            const double ru = (eps + au) * ainv, rv1 = (eps + av1) * ainv;
            xr[cpos] = ru * cu;                    xi[cpos] = ru * su;
            xr[cneg] = rv1 * (cv * pr - sv * pi);  xi[cneg] = rv1 * (cv * pi + sv * pr);
          }
          const double rv = (eps + av) * ainv, ru1 = (eps + au1) * ainv;
          xr += ldx; xi += ldx;
          xr[cpos] = rv * cv;                      xi[cpos] = rv * sv;
          xr[cneg] = ru1 * (cu * pr - su * pi);    xi[cneg] = ru1 * (cu * pi + su * pr);
        }
#endif
        if (tid >= nt - TSf) {   // weights, DC and signal columns: one thread per chunk row (the last wave has the fewest build items)
          const int row = tid - (nt - TSf), el = row >> 1, d = d0 + el;
          const int t = (row & 1) ? (mid + d) : (mid - d - 1);
          double w0 = 0.0, sv = 0.0;
          if (d <= mid && t >= 0) {
            const double w = window_value(false, t, N);
            w0 = w * w;
            sv = A.s[(size_t)(c - wl) + t];
          }
          const double nn = (double)(t - mid);
          Wp[row] = w0; Wp[32 + row] = w0 * nn; Wp[64 + row] = w0 * nn * nn;
          Xre[row * ldx + XCOL(2 * n, el)] = 1.0;  Xim[row * ldx + XCOL(2 * n, el)] = 0.0;   // DC column
          Xre[row * ldx + XCOL(Kc, el)] = sv;  Xim[row * ldx + XCOL(Kc, el)] = 0.0;  // signal column
        }
        __syncthreads();
        MF_STAMP(1);
        // (touching the next chunk's track lines here, as the tile kernel does, costs 5 % in this kernel: 630 vs 598 ms)
        // ---- contraction of the chunk (only the k-steps that hold samples: the rest of the last chunk has weight 0)
        const int pcs = (npairs - d0 < PE) ? (npairs - d0) : PE;
        const int ksn = (pcs + 1) >> 1;
        const int lq = lane >> 4, lcol = lane & 15;
        const double* wrow = Wp + lq;
#ifndef EAQHM_EXPERIMENT_NOCONTRACT   /* (timing experiment: no matrix products; wrong results) */
        {
          // k-step outermost: the operands of both tiles and the three weights are requested together, then the 18
          // independent MFMAs follow (row = 4 ks + lq, column rotation (lcol + row / 2) & 15)
          const int sw0 = lcol + (lq >> 1);
          int rb = lq * ldx;
#pragma clang loop unroll(disable)
          for (int ks = 0; ks < ksn; ++ks) {
            const int sw = (sw0 + 2 * ks) & 15;
            double aR[MF_NT], aI[MF_NT], bR[MF_NT], bI[MF_NT];
#pragma unroll
            for (int sl = 0; sl < MF_NT; ++sl) {
              const int ca = 16 * tI[sl], cb = 16 * tJ[sl];
              aR[sl] = Xre[rb + ca + sw]; aI[sl] = Xim[rb + ca + sw]; bR[sl] = Xre[rb + cb + sw]; bI[sl] = Xim[rb + cb + sw];
            }
            const double w0 = wrow[4 * ks], w1v = wrow[32 + 4 * ks], w2 = wrow[64 + 4 * ks];
            rb += 4 * ldx;
            // all weighted operands first, then the MFMAs back to back: with a multiply in front of every MFMA a wave
            // issues one only every ~186 cycles (93 per SIMD with its two waves) instead of every ~141
#ifdef MF_FOURPROD
            double nI[MF_NT], bRw[MF_NT][3], bIw[MF_NT][3];
#pragma unroll
            for (int sl = 0; sl < MF_NT; ++sl) {
              nI[sl] = -aI[sl];
#pragma unroll
              for (int w = 0; w < 3; ++w) {
                const double wv = (w == 0) ? w0 : (w == 1) ? w1v : w2;
                bRw[sl][w] = wv * bR[sl]; bIw[sl][w] = wv * bI[sl];
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            // Re += aR bR + aI bI,  Im += aR bI - aI bR  (conj(a) b): the two products of an accumulator are issued a
            // round of the other accumulators apart
#pragma unroll
            for (int sl = 0; sl < MF_NT; ++sl) {
              if (!live[sl]) continue;
#pragma unroll
              for (int w = 0; w < 3; ++w) {
                P1[sl][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR[sl], bRw[sl][w], P1[sl][w], 0, 0, 0);
                P3[sl][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR[sl], bIw[sl][w], P3[sl][w], 0, 0, 0);
              }
#pragma unroll
              for (int w = 0; w < 3; ++w) {
                P1[sl][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(aI[sl], bIw[sl][w], P1[sl][w], 0, 0, 0);
                P3[sl][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(nI[sl], bRw[sl][w], P3[sl][w], 0, 0, 0);
              }
            }
#else
            double sA[MF_NT], bRw[MF_NT][3], bIw[MF_NT][3], dw[MF_NT][3];
#pragma unroll
            for (int sl = 0; sl < MF_NT; ++sl) {
              sA[sl] = aR[sl] + aI[sl];
#pragma unroll
              for (int w = 0; w < 3; ++w) {
                const double wv = (w == 0) ? w0 : (w == 1) ? w1v : w2;
                bRw[sl][w] = wv * bR[sl]; bIw[sl][w] = wv * bI[sl];
                dw[sl][w] = bIw[sl][w] - bRw[sl][w];
              }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sl = 0; sl < MF_NT; ++sl) {
              if (!live[sl]) continue;
#pragma unroll
              for (int w = 0; w < 3; ++w) {
                P1[sl][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR[sl], bRw[sl][w], P1[sl][w], 0, 0, 0);
                P2[sl][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(aI[sl], bIw[sl][w], P2[sl][w], 0, 0, 0);
                P3[sl][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[sl], dw[sl][w], P3[sl][w], 0, 0, 0);
              }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#endif
        __syncthreads();
        MF_STAMP(2);
      }

      // ---- accumulators -> tiles of the stacked padded system [[G0,G1^H],[G1,G2]] + RHS row (eaqhm_ls_chol.h).
      // Base tiles are aligned with the stacked ones; positions beyond Kc inside a block are identity padding.
#pragma unroll
      for (int sl = 0; sl < MF_NT; ++sl) {
        if (!live[sl]) continue;
        const int I = tI[sl], J = tJ[sl];
        const int bl = lane & 15, b = 16 * J + bl;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int al = (lane >> 4) + 4 * rr, a = 16 * I + al;
#ifdef MF_FOURPROD
          const double gr = P1[sl][p][rr], gi = P3[sl][p][rr];
#else
          const double gr = P1[sl][p][rr] + P2[sl][p][rr], gi = P3[sl][p][rr] + (P1[sl][p][rr] - P2[sl][p][rr]);
#endif
          if (a == Kc) {   // signal row: conj(rhs) into row 0 of the RHS tile row, its energy on the diagonal
            if (b < Kc && p == 0) { double* t = S.T + tile_off(ntl - 1, J) + bl; t[0] = gr; t[256] = gi; }
            if (b < Kc && p == 1) { double* t = S.T + tile_off(ntl - 1, nbk + J) + bl; t[0] = gr; t[256] = gi; }
            if (b == Kc && p == 0) { double* t = S.T + tile_off(ntl - 1, ntl - 1); t[0] = gr; t[256] = 0.0; }
          }   // ... and inside the blocks the signal's row / column is one more identity padding position
          if (I >= nbk || J >= nbk) continue;   // the base tile row / column that holds only the signal
          const bool in = (a < Kc) && (b < Kc);
          if (p == 0 || p == 2) {
            const int o = (p == 0) ? 0 : nbk;
            double* t = S.T + tile_off(o + I, o + J) + al * 16 + bl;
            t[0] = in ? gr : ((a == b) ? 1.0 : 0.0);
            t[256] = in ? gi : 0.0;
          } else {   // cross block: G1 is Hermitian, the stacked system needs all of it
            double* t = S.T + tile_off(nbk + I, J) + al * 16 + bl;
            t[0] = in ? gr : 0.0;
            t[256] = in ? gi : 0.0;
            if (I != J) {
              double* u = S.T + tile_off(nbk + J, I) + bl * 16 + al;
              u[0] = in ? gr : 0.0;
              u[256] = in ? -gi : 0.0;
            }
          }
        }
        }
      }
    }
    __syncthreads();
    MF_STAMP(3);

#ifndef EAQHM_EXPERIMENT_NOCHOL   /* (timing experiment: the frame without its factorisation; wrong results) */
    tile_cholesky_memory<MFK_WAVES>(S.T, S.WT, S.D0, ntl, Kc, nbk, Xbase, xs, A.fault, dbg, (A.mode == 1) ? n : -1);
#endif
    MF_STAMP_START();
    write_record(A, xs, sh, mycols, f, n, inst, c, f0, seeds);
    MF_STAMP(10);
  }
}

// Adaptation 0 of the large frames: two real systems of order Kc + 1 per frame, factorised in the register file
// (eaqhm_ls_a0.h); three register budgets by the number of tile rows (12 / 17 / 24 tiles per wave on eight waves).  A
// kernel of its own: called from eaqhm_ls_mfma_kernel the three instantiations cost that kernel's other paths their
// register allocation (adaptation >= 1 launches 630 -> 741 ms).  Frames beyond MF_A0_M tile rows are left to
// eaqhm_ls_mfma_kernel (complex system through memory).
extern "C" __global__ void __launch_bounds__(MF_THREADS) eaqhm_ls_a0big_kernel(LsArgs A, int min_nb, int* cursor) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ int nxt;
  if (min_nb > 0 && A.cls[LS_BIG_CLASS] == 0) return;
  const int n_items = (min_nb > 0) ? A.cls[LS_BIG_CLASS] : A.n_frames;
  const int TB = A.Kcmax + 1, WP = ((A.Nmax >> 1) + 8) & ~7, NP = (A.Nmax + 7) & ~7;
  for (;;) {
    if (threadIdx.x == 0) nxt = atomicAdd(cursor, 1);
    __syncthreads();
    const int item = uni(nxt);
    __syncthreads();
    if (item >= n_items) break;
    const int f = uni((min_nb > 0) ? A.cls[16 + (size_t)LS_BIG_CLASS * A.n_frames + item] : item);
    if (min_nb == 0 && !frame_window_ok(A, uni(A.frame_c[f]), uni(A.frame_wl[f]))) continue;   // (see eaqhm_ls_mfma_kernel)
    const int m0 = (2 * uni(A.frame_K[f]) + 2 + 15) >> 4;
    if (m0 <= 13) a0_frame<12, MF_A0_M, 1>(A, lds, f, TB, MF_A0_NCH, WP, NP);
    else if (m0 <= 16) a0_frame<17, MF_A0_M, 1>(A, lds, f, TB, MF_A0_NCH, WP, NP);
    else if (m0 <= MF_A0_M) a0_frame<24, MF_A0_M, 1>(A, lds, f, TB, MF_A0_NCH, WP, NP);
  }
}

size_t ls_mfma_scratch_stride(int nmax, int Nmax, int Kcmax) {
  return (mf_scratch_doubles(nmax, Nmax, Kcmax) + 15) & ~(size_t)15;
}

// A.scratch / A.scratch_stride / A.work_counter / A.zloc / A.ztot are set by the caller (eaqhm_ls_batch), which has
// also run the zero-count pass (launch_ls_prepass) for adaptations >= 1
int launch_ls_mfma(eaqhm_ctx* ctx, LsArgs A, int grid, int min_nb) {
  const int nmax = A.nmax, Kcmax = A.Kcmax;
  const int nbmax = (Kcmax + 1 + 15) / 16;
  const int ldx_max = 16 * nbmax + 16;
  const int Npad = ((A.Nmax + 63) >> 6) << 6;
  // everything but the two chunk planes: weights, per-slot info, solution vector, small shared values
  const size_t fixed = (size_t)(96 + MF_CI * nmax + 4 * Kcmax + 16) * sizeof(double);
  if (fixed + (size_t)2 * 8 * ldx_max * sizeof(double) > 159 * 1024)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: problem too large for the MFMA variant");
  // doubles per chunk plane: as many basis rows per chunk as the LDS holds (up to 32; fewer, longer chunks amortise the
  // barriers); the planes double as work space of the slot set-up, the closed-form Gramian and the factorisation
  const int plane = (int)(((159 * 1024 - fixed) / (2 * sizeof(double))) & ~(size_t)15);
  const size_t lds_bytes = (size_t)2 * plane * sizeof(double) + fixed;
  const size_t tz_doubles = (size_t)3 * TZ_NQ * (Kcmax + 1) + 3 * (((A.Nmax >> 1) + 8) & ~7) + 2 * ((A.Nmax + 7) & ~7);
  if ((size_t)2 * plane < tz_doubles)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: frame size outside the closed-form Gramian's work space");
  if ((size_t)2 * plane < CH_LDS_DOUBLES || 2 * ((Kcmax + 15) / 16) + 1 > CH_NTMAX)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: frame size outside the tile factorisation's work space");
  if ((size_t)2 * plane * sizeof(double) < (size_t)nmax * (Npad >> 6) * 8 + (size_t)nmax * 4)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: frame size outside the slot set-up's work space");
  HIP_TRY(ctx, hipFuncSetAttribute((const void*)eaqhm_ls_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds_bytes));
  const size_t a0_bytes = a0_lds_doubles(MF_A0_M, 1, Kcmax + 1, MF_A0_NCH, ((A.Nmax >> 1) + 8) & ~7, (A.Nmax + 7) & ~7, Kcmax) * sizeof(double);
  // (159 KiB: the kernel also has a static __shared__ word, and dynamic + static must fit the 160 KiB of a workgroup)
  const int a0_onchip = (A.mode == 0 && a0_bytes <= 159 * 1024) ? 1 : 0;
  if (a0_onchip) {
    // its frame cursor: a free slot of the class header / of the zeroed counter block (eaqhm_ls_batch)
    int* cursor = (min_nb > 0) ? (A.cls + 15) : (A.work_counter + 2);
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)eaqhm_ls_a0big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)a0_bytes));
    hipLaunchKernelGGL(eaqhm_ls_a0big_kernel, dim3(grid), dim3(MF_THREADS), a0_bytes, ctx->stream, A, min_nb, cursor);
    HIP_TRY(ctx, hipGetLastError());
  }
  hipLaunchKernelGGL(eaqhm_ls_mfma_kernel, dim3(grid), dim3(MFK_THREADS), lds_bytes, ctx->stream, A, plane, min_nb, a0_onchip);
  HIP_TRY(ctx, hipGetLastError());
  return EAQHM_OK;
}

}  // namespace eaqhm
