// eaqhm_ls_tile.hip — the per-frame LS entirely on chip: Gramian AND factorisation on the FP64 matrix cores,
// the system matrix never leaves the register file (gfx950).
//
// A workgroup of 512 threads (8 waves, every wave owns system tiles in its registers) works on one frame at a
// time; frames are bucketed by size on the device and handed out by per-class atomic cursors, largest first.
//
// Stacked basis.  With Y[t] = w_t * [ E2(t) | n_t E2(t) | s_t ]  (2Kc+1 columns; E2 = [negative | DC | positive]
// columns of functions.py:516-519, w the analysis window, n_t = t - mid) the normal equations of
// functions.py:521-530 are the leading 2Kc x 2Kc block of  Y^H Y  and the right-hand side is its last column.
// Factorising  Y^H Y  WITH that last row/column leaves conj(L^-1 rhs) in the last row: the forward substitution
// costs nothing.  Every 16x16 complex tile of Y^H Y is one MFMA accumulation over time, owned by one wave
// (tiles numbered column by column, tile x -> wave x%8, slot x/8) from the first sample to the last back-substitution step.
//
//   A1     per-slot set-up: zero-count look-ups decide whether the slot's window has a gap; gap-free slots need
//          only their centre values, the others are bridged into per-workgroup scratch rows
//   A3+B   chunks of 16 sample PAIRS (mid-d-1, mid+d), centre outwards, built in LDS by all threads: track values
//          loaded directly, phase = running sum (16-lane DPP scan + per-slot carry); the pair shares its sincos
//          because the negative-frequency column at u is the time-reversed positive one (functions.py:284-285);
//          contracted with v_mfma_f64_16x16x4_f64, LDS operand reads software-pipelined one k-step ahead
//   B0     adaptation 0: no basis at all — the Gramian in closed form from Toeplitz tables, then two real systems of
//          half the order factorised side by side (a0_frame, eaqhm_ls_a0.h)
//   C      right-looking tile Cholesky with look-ahead, 2 workgroup barriers per tile row: trailing update with the
//          previous panel (the next diagonal tile first); its owner wave factorises that tile on the matrix cores
//          (diag_D: 2x2 block pivots, one rank-2 MFMA per plane and step, rows handed round by ds_bpermute) while a
//          second wave follows one step behind and builds the inverse (diag_Z) and the others finish their updates;
//          panel tiles are multiplied by the inverse (MFMA) and published in LDS; no global memory traffic
//   C'     back substitution from the L tiles still sitting in the owners' registers; z, x vectors in LDS
//   D      frequency mismatch, acceptance, record row (eaqhm_ls_common.h)
//
// Frames with more than 13 tile rows (Kc > 103) do not fit the register budget and are left to
// eaqhm_ls_mfma_kernel (same Gramian, factorisation through scratch memory).
#include "eaqhm_ls_common.h"
#include "eaqhm_ls_chol.h"
#include "eaqhm_ls_a0.h"
#include "eaqhm_ls_tilemap.h"

namespace eaqhm {


// wave that owns the diagonal tile (jb, jb): see the ownership rules in tile_frame
#define DIAG_OWNER(jb, nt) ((int)TL_DIAG[nt][jb])
#define TL_THREADS 512
#define TL_WAVES 8
#define TL_CW 8         // all waves own tiles
#define TL_NTMAX 13
#define CI_STRIDE 20    // per-slot info: 16 chunk carries, qmid, 1/(am_mid+eps), rho.re, rho.im
#define CI_NCH 16
#define A0_NS 7        // adaptation 0: tiles per wave of a real system of <= 7 tile rows (order Kc + 1 <= 104) on 4 waves
#define A0_M 7

__device__ inline void sys_tile_of(int x, int& P, int& Q) {
  P = (int)((sqrtf(8.0f * (float)x + 1.0f) - 1.0f) * 0.5f);
  while ((P + 1) * (P + 2) / 2 <= x) ++P;
  while (P * (P + 1) / 2 > x) --P;
  Q = x - P * (P + 1) / 2;
}
// (adaptation 0: closed-form Gramian and two real systems — eaqhm_ls_a0.h; table sizes for frames of this kernel)
#define TZ_TB 104     // table stride: m = 0 .. 2 n <= 102
#define TZ_NCH 8      // chunks of the t range (deterministic two-level summation)

// One frame with NS tiles per wave.  Not inlined: each register budget gets its own register allocation (inlining
// the five budgets into one kernel body spills several hundred VGPRs).
template <int NS>
__device__ __attribute__((noinline)) void tile_frame(const LsArgs& A, int TS_, int ldx_max_, double* lds, int f_) {
  const int TS = uni(TS_), ldx_max = uni(ldx_max_), f = uni(f_);
  const int tid = threadIdx.x, nt_thr = TL_THREADS;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- LDS carve-up: region U is the basis chunk (+ slot info) in the Gramian phase, tile storage afterwards
  double* U = uni(lds);
  double* Xre = U;
  double* Xim = Xre + (size_t)TS * ldx_max;
  const size_t usize_g = (size_t)2 * TS * ldx_max;
  double* ci = U + usize_g;                          // [52][CI_STRIDE]
  unsigned long long* masks = (unsigned long long*)(ci + (size_t)CI_STRIDE * 52);  // [52][16]  (n <= 51 for Kc <= 103)
  double* PanR = U;                                  // [NT][TILE]  published panel tiles, [k][row]
  double* PanI = PanR + TL_NTMAX * TL_TILE;
  double* WtR = PanI + TL_NTMAX * TL_TILE;           // [NT][TILE]  (W^H)[k][j] of every diagonal tile
  double* WtI = WtR + TL_NTMAX * TL_TILE;
  double* LdR = WtI + TL_NTMAX * TL_TILE;            // [TILE]      last diagonal factor, [row][col]
  double* LdI = LdR + TL_TILE;
  double* Dc = LdI + TL_TILE;                        // [DG_TILE]   diagonal tile being factorised (complex, rows DG_LD apart)
  double* Zc = Dc + DG_TILE;                         // [DG_TILE]   its inverse in the making
  double* zv = Zc + DG_TILE;                        // 2*16*NT
  double* xv = zv + 2 * 16 * TL_NTMAX;               // 2*16
  const size_t usize_c = (size_t)(xv + 32 - U);
  double* xs = U + usize_c;                          // 4*Kcmax   solution in natural order
  double* sh = xs + 4 * uni(A.Kcmax);                // 16
  double* win = sh + 16;                             // [64*CI_NCH] analysis window of the frame
  double* sig = win + 64 * CI_NCH;                   // [64*CI_NCH] signal window of the frame
  double* dorig = sig + 64 * CI_NCH;                 // [16*NT]     original diagonal of the system (singularity check)

  const int Npad = ((uni(A.Nmax) + 63) >> 6) << 6;
  double* Qs = uni(A.scratch) + (size_t)blockIdx.x * (size_t)uni((int)A.scratch_stride);  // bridged fm[j][t]
  double* Rs = Qs + (size_t)Npad * uni(A.nmax);                                           // bridged am[j][t]
  const bool seeds = uni((int)(A.any_seed && (*A.any_seed != 0))) != 0;
  // phases are q * (2 pi / fs) here, (2 pi q) / fs in the reference (functions.py:513, :453): one rounding each way,
  // <= 2 ulp of the phase apart, and no IEEE division per basis sample
  const double w1 = uni(2.0 * M_PI / A.fs);
  const int PE = TS / 2;
  const double* sA = uni(A.s);
  const int lcol = lane & 15, lq = lane >> 4;
  unsigned long long* dbg = uni(A.debug);
  unsigned long long t_prev = 0;
#define STAMP(ph)                                                   \
  do {                                                              \
    if (dbg && tid == 0) {                                          \
      unsigned long long t_now = __builtin_amdgcn_s_memtime();      \
      atomicAdd(dbg + (ph), t_now - t_prev);                        \
      t_prev = t_now;                                               \
    }                                                               \
  } while (0)

  {
    const int n = uni(A.ncol[f]);
    const int Kc = 2 * n + 1, Ms = 2 * Kc + 1;   // stacked columns incl. the signal
    const int nt = (Ms + 15) >> 4;
    const int c = uni(A.frame_c[f]), wl = uni(A.frame_wl[f]), inst = uni(A.frame_inst[f]);
    const int N = 2 * wl + 1, mid = wl;
    const int ldx = (nt << 4) + ((nt & 1) ? 0 : 16);  // ≡ 16 (mod 32): MFMA operand reads hit disjoint bank halves
    const int ntiles = nt * (nt + 1) / 2;
    const int is = 2 * Kc - 16 * (nt - 1);  // position of the signal column inside the last tile row (2,6,10,14)
    const double f0 = uni(A.f0_stale);
    const int* mycols = uni(A.cols) + (size_t)f * uni(A.Kmax);
    const int npairs = mid + 1;  // pairs e = 0..mid: (u, v) = (e-1, N-1-e)

    if (dbg && tid == 0) t_prev = __builtin_amdgcn_s_memtime();
    // region U may hold tiles of the previous frame: make the basis chunk finite and its padding zero
    for (int q = tid; q < 2 * TS * ldx_max; q += nt_thr) Xre[q] = 0.0;
    for (int t = tid; t < N; t += nt_thr) {
      win[t] = window_value(false, t, N);
      sig[t] = sA[(size_t)(c - wl) + t];
    }
    __syncthreads();
    int* gappy = (int*)(masks + (size_t)52 * CI_NCH);   // [52] flags
    prepare_slots<CI_STRIDE, TL_WAVES>(A, Qs, Rs, Npad, ci, masks, gappy, mycols, n, N, mid, c, wl, seeds, lane, wave, CI_NCH);
    STAMP(0);

    // system tiles of this wave
    // Three real products per complex one need a third accumulator per tile: the first NM3 tiles of a wave.  Register
    // budget: every tile of the small frames, none beyond 7 slots (tried: 6 of 9 and 4 of 10 slots, 192 accumulator
    // VGPRs like the 12-slot budget — the extra spills cost more than the 17 % / 10 % fewer MFMAs gain: 1.23 M vs
    // 1.27 M frames/s on the 60 s workload).
    constexpr int NM3 = (NS <= 7) ? NS : 0;
    d4 accR[NS], accI[NS], acc3[NM3 ? NM3 : 1];
    int tP[NS], tQ[NS];
    bool live[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      accR[sl] = (d4){0, 0, 0, 0};
      accI[sl] = (d4){0, 0, 0, 0};
      if (sl < NM3) acc3[sl] = (d4){0, 0, 0, 0};
      // Ownership: TL_MAP (eaqhm_ls_tilemap.h, generated by tools/tile_map_search.py) — the wave that owns diagonal tile jb
      // holds few tiles of the trailing matrix of stage jb, so that after diag_D it does not keep the others waiting at the
      // stage barrier with trailing tiles of its own; at most NS tiles per wave, tile counts per SIMD equal to within one
      // (the contraction stays balanced), at most ... panel tiles of a column on one wave.  (Rounds 1-2 dealt the tiles
      // out column by column, tile x on wave x % 8.)
      const int code = TL_MAP[nt][wave][sl];
      live[sl] = code != 0xFF;
      const int P = live[sl] ? (code >> 4) : 0, Q = live[sl] ? (code & 15) : 0;
      tP[sl] = __builtin_amdgcn_readfirstlane(P);   // wave-uniform: scalar registers, not spill slots
      tQ[sl] = __builtin_amdgcn_readfirstlane(Q);
    }

    // ================= Gramian =================
    // sample pairs (u, v) = (mid-d-1, mid+d), d = 0..mid, taken from the centre outwards so that the phase
    // integral of functions.py:508-515 relative to the centre is a running sum
    for (int d0 = 0; d0 < npairs; d0 += PE) {
      // logical column cc of chunk rows (2*el, 2*el+1) lives at XCOL(cc, el): the 16 lanes that write one column
      // of 16 different pairs hit 16 different banks, and MFMA operand reads stay conflict-free
#pragma clang loop unroll(disable)
      for (int idx = tid; idx < 16 * n; idx += nt_thr) {
        const int el = idx & 15, j = idx >> 4, d = d0 + el;
        const bool act = d <= mid;
        const int u = mid - d - 1, v = mid + d;
        double su = 0, cu = 1, sv, cv;
        double pur = 0, pui = 0, nur = 0, nui = 0, pvr, pvi, nvr, nvi;  // positive / negative column values
        {
          double* cj = ci + j * CI_STRIDE;
          const double* fb = ((const double**)cj)[2];   // bridged copy or the track itself (prepare_slots)
          const double* ab = ((const double**)cj)[3];
          // every load of the item up front, indices clamped into the window (results of clamped ones unused)
          const int dc = act ? d : mid;
          const double fu1 = fb[mid - dc], fv = fb[mid + dc];
          const double au = ab[(mid - dc - 1 >= 0) ? (mid - dc - 1) : 0], au1 = ab[mid - dc];
          const double av = ab[mid + dc], av1 = ab[(mid + dc + 1 < N) ? (mid + dc + 1) : (N - 1)];
          // F(v) - F(mid) = sum of fm over (mid, v];  F(u) - F(mid) = -sum over [u+1, mid]
          const double xu = act ? fu1 : 0.0, xv = (act && d >= 1) ? fv : 0.0;
          const double qv = cj[0] + scan16(xv), qu = -(cj[1] + scan16(xu));
          if (el == 15) { cj[0] = qv; cj[1] = -qu; }   // carried to the next chunk (read again after two barriers)
          if (!act) continue;
          const double ainv = cj[17], pr = cj[18], pi = cj[19];
          sincos_cw(qu * w1, &su, &cu);
          sincos_cw(qv * w1, &sv, &cv);
          const double eps = 10e-5;
          // positive column at t uses E1(t); negative column at t uses ratio[mirror+1] * E1(mirror) * rho
          if (u >= 0) {
            const double ru = (eps + au) * ainv, rv1 = (eps + av1) * ainv;
            pur = ru * cu;                    pui = ru * su;
            nur = rv1 * (cv * pr - sv * pi);  nui = rv1 * (cv * pi + sv * pr);
          }
          const double rv = (eps + av) * ainv, ru1 = (eps + au1) * ainv;
          pvr = rv * cv;                      pvi = rv * sv;
          nvr = ru1 * (cu * pr - su * pi);    nvi = ru1 * (cu * pi + su * pr);
        }
        double* xr = Xre + (2 * el) * ldx;
        double* xi = Xim + (2 * el) * ldx;
        const int cpos = XCOL(n + 1 + j, el), cneg = XCOL(j, el);               // amplitude columns
        const int spos = XCOL(Kc + n + 1 + j, el), sneg = XCOL(Kc + j, el);     // slope columns (n_t times)
        const double wv = win[v], nv = (double)(v - mid);
        if (u >= 0) {
          const double wu = win[u], nu = (double)(u - mid);
          pur *= wu; pui *= wu; nur *= wu; nui *= wu;
          xr[cpos] = pur;      xi[cpos] = pui;      xr[cneg] = nur;      xi[cneg] = nui;
          xr[spos] = nu * pur; xi[spos] = nu * pui; xr[sneg] = nu * nur; xi[sneg] = nu * nui;
        }
        pvr *= wv; pvi *= wv; nvr *= wv; nvi *= wv;
        xr += ldx; xi += ldx;
        xr[cpos] = pvr;      xi[cpos] = pvi;      xr[cneg] = nvr;      xi[cneg] = nvi;
        xr[spos] = nv * pvr; xi[spos] = nv * pvi; xr[sneg] = nv * nvr; xi[sneg] = nv * nvi;
      }
      // DC / signal columns: one thread per chunk row, spread over the waves (lanes 0, 16, 32, 48)
      if ((tid & 15) == 0 && (tid >> 4) < TS) {
        const int row = tid >> 4;
        const int d = d0 + (row >> 1);
        const int t = (row & 1) ? (mid + d) : (mid - d - 1);
        const bool ok = (d <= mid) && (t >= 0);
        const double w = ok ? win[t] : 0.0;
        const double sval = ok ? sig[t] : 0.0;
        const double nn = (double)(t - mid);
        const int el = row >> 1;
        double* xr = Xre + row * ldx;
        double* xi = Xim + row * ldx;
        xr[XCOL(n, el)] = w;               xi[XCOL(n, el)] = 0.0;            // DC column
        xr[XCOL(Kc + n, el)] = w * nn;     xi[XCOL(Kc + n, el)] = 0.0;       // its slope copy
        xr[XCOL(2 * Kc, el)] = w * sval;   xi[XCOL(2 * Kc, el)] = 0.0;       // signal column
      }
      // rows beyond the window (the virtual sample u = -1 and the tail of the last chunk): zero
      if (d0 + PE > mid) {
        for (int q = tid; q < TS * 16 * nt; q += nt_thr) {
          const int row = q / (16 * nt), col = q - row * (16 * nt);
          const int d = d0 + (row >> 1);
          const int t = (row & 1) ? (mid + d) : (mid - d - 1);
          if (d > mid || t < 0) { Xre[row * ldx + col] = 0.0; Xim[row * ldx + col] = 0.0; }
        }
      }
      STAMP(12);
      __syncthreads();
      STAMP(1);
      const int pcs = (npairs - d0 < PE) ? (npairs - d0) : PE;   // sample pairs in this chunk
      const int ksl = (pcs + 1) >> 1;                             // k-steps (4 rows each) that hold samples
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        if (!live[sl]) continue;
        const int ca = 16 * tP[sl], cb = 16 * tQ[sl];
        // operands of k-step ks+1 are requested before the MFMAs of k-step ks are issued (LDS latency hidden
        // behind the matrix pipe); row = 4 ks + lq, column swizzle (lcol + row/2) & 15
        int rb = lq * ldx;
        const int sw0 = lcol + (lq >> 1);
        asm volatile("" : "+v"(rb));   // keep the address arithmetic inside the loop (hoisted, it spills)
        const int plane = TS * ldx_max;
        if (ksl < 8) {   // last chunk of the window: only the k-steps that hold samples (the other rows are zero)
#pragma clang loop unroll(disable)
          for (int ks = 0; ks < ksl; ++ks) {
            const int sw = (sw0 + 2 * ks) & 15;
            const double aR = Xre[rb + ca + sw], aI = Xre[rb + ca + sw + plane];
            const double bR = Xre[rb + cb + sw], bI = Xre[rb + cb + sw + plane];
            rb += 4 * ldx;
            if (sl < NM3) {
              accR[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bR, accR[sl], 0, 0, 0);
              acc3[sl < NM3 ? sl : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aI, bI, acc3[sl < NM3 ? sl : 0], 0, 0, 0);
              accI[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR + aI, bI - bR, accI[sl], 0, 0, 0);
            } else {
              accR[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bR, accR[sl], 0, 0, 0);
              accI[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bI, accI[sl], 0, 0, 0);
              accR[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aI, bI, accR[sl], 0, 0, 0);
              accI[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aI, bR, accI[sl], 0, 0, 0);
            }
          }
          continue;
        }
        double aR, aI, bR, bI;
        {
          const int sw = sw0 & 15;
          aR = Xre[rb + ca + sw]; aI = Xre[rb + ca + sw + plane]; bR = Xre[rb + cb + sw]; bI = Xre[rb + cb + sw + plane];
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {   // TS = 32 rows
          double naR = 0, naI = 0, nbR = 0, nbI = 0;
          if (ks < 7) {
            rb += 4 * ldx;
            const int sw = (sw0 + 2 * (ks + 1)) & 15;
            naR = Xre[rb + ca + sw]; naI = Xre[rb + ca + sw + plane]; nbR = Xre[rb + cb + sw]; nbI = Xre[rb + cb + sw + plane];
          }
          __builtin_amdgcn_sched_barrier(0);   // the requests above stay ahead of the MFMAs below
          if (sl < NM3) {   // three real products per complex one: P1 = aR bR, P2 = aI bI, P3 = (aR+aI)(bI-bR)
            accR[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bR, accR[sl], 0, 0, 0);
            acc3[sl < NM3 ? sl : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aI, bI, acc3[sl < NM3 ? sl : 0], 0, 0, 0);
            accI[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR + aI, bI - bR, accI[sl], 0, 0, 0);
          } else {
            accR[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bR, accR[sl], 0, 0, 0);
            accI[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bI, accI[sl], 0, 0, 0);
            accR[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aI, bI, accR[sl], 0, 0, 0);
            accI[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aI, bR, accI[sl], 0, 0, 0);
          }
          aR = naR; aI = naI; bR = nbR; bI = nbI;
        }
      }
      __syncthreads();
      STAMP(2);
    }
    if constexpr (NM3 > 0) {   // Re = P1 + P2,  Im = aR bI - aI bR = P3 + P1 - P2
#pragma unroll
      for (int sl = 0; sl < NM3; ++sl) {
        const d4 p1 = accR[sl], p2 = acc3[sl];
        accR[sl] = p1 + p2;
        accI[sl] = accI[sl] + (p1 - p2);
      }
    }

    {
      // ---- padding positions of the last tile row/column (beyond the signal) get an identity row/column
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        if (!live[sl] || tP[sl] != nt - 1) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = lq + 4 * r, col = lcol;
          const bool pad = (row > is) || (tQ[sl] == nt - 1 && col > is);
          if (pad) {
            accR[sl][r] = (tQ[sl] == nt - 1 && row == col) ? 1.0 : 0.0;
            accI[sl][r] = 0.0;
          }
        }
      }
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {   // original diagonal entries, for the collapsed-pivot check of diag_coop
        if (!live[sl] || tP[sl] != tQ[sl]) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (lq + 4 * r == lcol) dorig[16 * tP[sl] + lcol] = accR[sl][r];
      }
      // Right-looking tile Cholesky with look-ahead.  Stage jb: the trailing update with panel jb-1 — the diagonal tile
      // (jb, jb) first, whose owner then factorises and inverts it on its own (diag_wave: one wave, matrix cores, no
      // workgroup barrier) while the other waves are still in their updates — barrier — panel jb — barrier.
      double* post = Dc;                 // what diag_D posts for diag_Z (Dc and Zc are contiguous: 1088 doubles)
      double* dumpD = Dc + DGP_DOUBLES;  // [128] dump slots of diag_D
      double* zs = dumpD + 128;          // [256] row exchange, dump slots and row factors of diag_Z
      int* dflag = (int*)(zs + 256);     // step counter of the diagonal pipeline
      if (tid == 0) *dflag = 0;          // (region U held the last basis chunk until the barrier that ended the Gramian)
      __syncthreads();
#define TRAILING_UPDATE(sl)                                                                         \
  {                                                                                                 \
    const double* ar = PanR + tP[sl] * TL_TILE;                                                     \
    const double* ai = PanI + tP[sl] * TL_TILE;                                                     \
    const double* br = PanR + tQ[sl] * TL_TILE;                                                     \
    const double* bi = PanI + tQ[sl] * TL_TILE;                                                     \
    /* T -= L_P L_Q^H, three real products: P1 = re re', P2 = im im', P3 = (re+im)(im'-re');   */   \
    /* Re -= P1 + P2,  Im += P3 + P1 - P2  (P3 accumulates straight into the imaginary part)   */   \
    d4 p1 = (d4){0, 0, 0, 0}, p2 = (d4){0, 0, 0, 0};                                                \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                              \
      const int o = (4 * ks + lq) * TL_LD + lcol;                                                   \
      const double aR = ar[o], aI = ai[o], lR = br[o], lI = bi[o];                                  \
      p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, lR, p1, 0, 0, 0);                               \
      p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(aI, lI, p2, 0, 0, 0);                               \
      accI[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR + aI, lI - lR, accI[sl], 0, 0, 0);         \
    }                                                                                               \
    accR[sl] = accR[sl] - (p1 + p2);                                                                \
    accI[sl] = accI[sl] + (p1 - p2);                                                                \
  }
      for (int jb = 0; jb < nt; ++jb) {
        d4 Rt = (d4){0, 0, 0, 0}, It = (d4){0, 0, 0, 0};
        bool mine = false;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
          if (!live[sl] || tP[sl] != jb || tQ[sl] != jb) continue;
          if (jb > 0) TRAILING_UPDATE(sl)
          Rt = accR[sl]; It = accI[sl];
          mine = true;
        }
        // real unknowns: every position but, in the last tile, the signal column `is` and the padding behind it
#ifndef EAQHM_EXPERIMENT_NODIAG   /* (timing experiment: what a frame costs without the diagonal steps; wrong results) */
        if (mine) {
          unsigned long long td0 = 0;
          unsigned long long* tlast = (unsigned long long*)(dflag + 2);   // (diagnostics: end of the previous diag_D)
          const bool tdiag = dbg && uni(A.debug_diag);
          if (tdiag) td0 = __builtin_amdgcn_s_memtime();
          diag_D(Rt, It, post, nullptr, dflag, 16 * jb, dumpD, LdR, LdI, jb == nt - 1);
          if (tdiag && lane == 0) {   // slots 9 / 11 / 13 / 14: cycles inside diag_D, from one diag_D to the next, counts
            const unsigned long long td1 = __builtin_amdgcn_s_memtime();
            atomicAdd(dbg + 9, td1 - td0);
            if (jb > 0) { atomicAdd(dbg + 11, td0 - *tlast); atomicAdd(dbg + 14, 1ull); }
            atomicAdd(dbg + 13, 1ull);
            *tlast = td1;
          }
        }
#else
        if (mine) asm volatile("" ::"v"(Rt[0]), "v"(It[0]));
#endif
        STAMP(10);
        if (jb > 0) {
#pragma unroll
          for (int sl = 0; sl < NS; ++sl) {
            if (!live[sl] || tQ[sl] < jb || (tP[sl] == jb && tQ[sl] == jb)) continue;
            TRAILING_UPDATE(sl)
          }
        }
        STAMP(8);
#ifndef EAQHM_EXPERIMENT_NODIAG
        // The helper wave builds the inverse (eaqhm_ls_tilemap.h) — AFTER its own trailing tiles: diag_D's posts wait for
        // it in LDS and it runs through them without the owner's pace, so its tiles are not what the stage ends on.
        if (!mine && wave == (int)TL_HELP[nt][jb])
          diag_Z(post, dflag, 16 * jb, zs, WtR + jb * TL_TILE, WtI + jb * TL_TILE, dorig + 16 * jb,
                 (jb == nt - 1) ? is : 16, uni(A.fault));
#endif
        __syncthreads();  // (A) inverse of the diagonal tile published; every read of panel jb-1 done
        STAMP(6);
        // ---- panel tiles (P > jb, Q == jb): X = T W^H, published as Pan[P][k][row]
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
          if (!live[sl] || tQ[sl] != jb || tP[sl] == jb) continue;
          double* tr = PanR + tP[sl] * TL_TILE;   // the tile's own panel slot doubles as transposition buffer
          double* ti = PanI + tP[sl] * TL_TILE;
#pragma unroll
          for (int r = 0; r < 4; ++r) {  // T[row = lq+4r][col = lcol] -> tmp[k = col][i = row]
            tr[lcol * TL_LD + lq + 4 * r] = accR[sl][r];
            ti[lcol * TL_LD + lq + 4 * r] = accI[sl][r];
          }
          __builtin_amdgcn_wave_barrier();
          // X = T W^H with three real products per complex one: P1 = re re, P2 = im im, P3 = (re+im)(re'+im')
          d4 p1 = (d4){0, 0, 0, 0}, p2 = (d4){0, 0, 0, 0}, p3 = (d4){0, 0, 0, 0};
          const double* wr = WtR + jb * TL_TILE;
          const double* wi = WtI + jb * TL_TILE;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const int o = (4 * ks + lq) * TL_LD + lcol;
            const double aR = tr[o], aI = ti[o], bR = wr[o], bI = wi[o];
            p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bR, p1, 0, 0, 0);
            p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(aI, bI, p2, 0, 0, 0);
            p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(aR + aI, bR + bI, p3, 0, 0, 0);
          }
          const d4 xr = p1 - p2, xi = p3 - (p1 + p2);
          accR[sl] = xr; accI[sl] = xi;  // the finished L tile stays here for the back substitution
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            tr[lcol * TL_LD + lq + 4 * r] = xr[r];
            ti[lcol * TL_LD + lq + 4 * r] = xi[r];
          }
        }
        STAMP(7);
        __syncthreads();  // (C) panel jb published
      }
#undef TRAILING_UPDATE
      __syncthreads();  // end of factorisation
      STAMP(3);

      // ============ back substitution  L^H x = y,  y = conj(row `is` of the last tile row) ============
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        if (!live[sl] || tP[sl] != nt - 1 || tQ[sl] == nt - 1) continue;
        if (lq == (is & 3)) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r == (is >> 2)) {
              zv[2 * (16 * tQ[sl] + lcol)] = accR[sl][r];
              zv[2 * (16 * tQ[sl] + lcol) + 1] = -accI[sl][r];
            }
        }
      }
      if (tid < 16) {
        const bool ok = tid < is;
        zv[2 * (16 * (nt - 1) + tid)] = ok ? LdR[is * TL_LD + tid] : 0.0;
        zv[2 * (16 * (nt - 1) + tid) + 1] = ok ? -LdI[is * TL_LD + tid] : 0.0;
      }
      for (int q = tid; q < 4 * Kc; q += nt_thr) xs[q] = 0.0;
      __syncthreads();  // y gathered
      for (int P = nt - 1; P >= 0; --P) {
        if (tid < 256) {  // x_P = W_PP^H z_P : thread (i, k) takes one term of row i, 16-lane shuffle reduction
          const int i = tid >> 4, k = tid & 15;
          double xr = 0, xi = 0;
          if (k >= i) {   // W^H is upper triangular
            const double wr = WtR[P * TL_TILE + i * TL_LD + k], wi = WtI[P * TL_TILE + i * TL_LD + k];
            const double zr = zv[2 * (16 * P + k)], zi = zv[2 * (16 * P + k) + 1];
            xr = wr * zr - wi * zi;
            xi = wr * zi + wi * zr;
          }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) { xr += __shfl_xor(xr, o); xi += __shfl_xor(xi, o); }
          if (k == 0) {
            xv[2 * i] = xr; xv[2 * i + 1] = xi;
            const int q = 16 * P + i;   // natural order: amplitudes then slopes
            if (q < 2 * Kc) { xs[2 * q] = xr; xs[2 * q + 1] = xi; }
          }
        }
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
          if (!live[sl] || tP[sl] != P || tQ[sl] == P) continue;
          double sr = 0, si = 0;  // sum_i conj(L[i][j]) x[i] over this lane's rows i = lq + 4r
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double xr = xv[2 * (lq + 4 * r)], xi = xv[2 * (lq + 4 * r) + 1];
            const double lr = accR[sl][r], li = accI[sl][r];
            sr += lr * xr + li * xi;
            si += lr * xi - li * xr;
          }
          sr += __shfl_xor(sr, 16); si += __shfl_xor(si, 16);
          sr += __shfl_xor(sr, 32); si += __shfl_xor(si, 32);
          if (lq == 0) {
            zv[2 * (16 * tQ[sl] + lcol)] -= sr;
            zv[2 * (16 * tQ[sl] + lcol) + 1] -= si;
          }
        }
        __syncthreads();
      }
      STAMP(4);
    }

    write_record(A, xs, sh, mycols, f, n, inst, c, f0, seeds);
    STAMP(5);
  }
}

// Frames bucketed by the number of tile rows of their system (see LsArgs::cls).
__device__ inline int frame_class(int n) {
  const int nt = (2 * (2 * n + 1) + 1 + 15) >> 4;
  return nt <= 8 ? 0 : nt == 9 ? 1 : nt == 10 ? 2 : nt == 11 ? 3 : nt == 12 ? 4 : nt <= TL_NTMAX ? 5 : LS_BIG_CLASS;
}
extern "C" __global__ void __launch_bounds__(256) eaqhm_ls_classify_kernel(LsArgs A) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
  int cl = (f < A.n_frames) ? frame_class((A.mode == 0) ? A.frame_K[f] : A.ncol[f]) : -1;
  if (f < A.n_frames) {   // the frame's window [c - wl, c + wl] inside the signal and (mode 1, with the sample before it) inside the resident tracks
    if (!frame_window_ok(A, A.frame_c[f], A.frame_wl[f])) { cl = -1; atomicAdd(A.fault + 2, 1); }   // dropped: its record row is not written
  }
  if (A.mode == 1 && cl >= 0) {   // chunks of the zero counts this frame's window (and the sample before it) touches
    const long long c = A.frame_c[f], wl = A.frame_wl[f];
    const long long lo = (c - wl - 1 > 0) ? (c - wl - 1) : 0;
    for (long long ch = lo >> 10; ch <= ((c + wl) >> 10); ++ch) A.zflag[ch] = 1;
  }
  for (int c = 0; c < LS_NCLS; ++c) {   // one atomic per wave and class, positions from the ballot
    const unsigned long long m = __ballot(cl == c);
    if (m == 0ull) continue;
    int base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(A.cls + c, __popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (cl == c) A.cls[16 + (size_t)c * A.n_frames + base + __popcll(m & ((1ull << lane) - 1ull))] = f;
  }
}

// Zero counts of every frequency track, in chunks of 1024 samples (LsArgs::zloc / ztot): a frame then knows with two
// look-ups per slot whether its window needs bridging, instead of scanning n windows of N samples.
// fm / zloc: biased by the first resident sample t_lo (LsArgs), rows of Lt samples; samples outside [t_lo, t_lo + Lt) are
// not resident (no frame window of this launch reaches them) and count as nonzero.
// Grid: x = the 1024-sample chunks that overlap the resident window (first one: ch0), y = slots.
extern "C" __global__ void __launch_bounds__(256) eaqhm_ls_zero_prefix_kernel(const double* __restrict__ fm, long long Lt,
                                                                              long long t_lo, int ch0, int zchunks,
                                                                              const unsigned char* __restrict__ zflag,
                                                                              unsigned short* __restrict__ zloc,
                                                                              int* __restrict__ ztot) {
  __shared__ int wsum[4];
  const int k = blockIdx.y, ch = ch0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (!zflag[ch]) return;   // no frame of this launch (this rank) looks here
  const long long base = (long long)ch << 10;
  int run = 0;
  for (int p = 0; p < 4; ++p) {
    const long long t = base + 256 * p + tid;
    const bool in = t >= t_lo && t < t_lo + Lt;
    const bool z = in && (fm[(size_t)k * Lt + t] == 0.0);
    const unsigned long long m = __ballot(z);
    const int incl = __popcll(m & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull)));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int v = wsum[w]; tot += v; if (w < wave) off += v; }
    if (in) zloc[(size_t)k * Lt + t] = (unsigned short)(run + off + incl);
    run += tot;
    __syncthreads();
  }
  if (tid == 0) ztot[(size_t)k * zchunks + ch] = run;
}

// One persistent launch for every frame size: each workgroup works through the size classes, largest first, with
// the register budget (tiles per wave) of the class.
extern "C" __global__ void __launch_bounds__(TL_THREADS) eaqhm_ls_tile_kernel(LsArgs A, int TS, int ldx_max) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ int nxt;
#define RUN_CLASS(NSV, C)                                                                \
  for (;;) {                                                                             \
    if (threadIdx.x == 0) nxt = atomicAdd(A.cls + 8 + C, 1);                             \
    __syncthreads();                                                                     \
    const int item = nxt;                                                                \
    __syncthreads();                                                                     \
    if (item >= A.cls[C]) break;                                                         \
    if (A.mode == 0) a0_frame<A0_NS, A0_M, 2>(A, lds, A.cls[16 + (size_t)C * A.n_frames + item], TZ_TB, TZ_NCH, 520, 64 * CI_NCH);                          \
    else tile_frame<NSV>(A, TS, ldx_max, lds, A.cls[16 + (size_t)C * A.n_frames + item]);               \
  }
  RUN_CLASS(12, 5)   // 91 tiles
  RUN_CLASS(10, 4)   // 78 tiles
  RUN_CLASS(9, 3)    // 66 tiles
  RUN_CLASS(7, 2)    // 55 tiles
  RUN_CLASS(6, 1)    // 45 tiles
  RUN_CLASS(5, 0)    // <= 36 tiles
#undef RUN_CLASS
}

// dynamic LDS the kernel may ask for: it also has a static __shared__ word (the frame cursor), and dynamic + static must
// fit the 160 KiB of a workgroup
#define TL_LDS_LIMIT (159 * 1024)
static size_t tl_usize_c() {
  return (size_t)4 * TL_NTMAX * TL_TILE + 2 * TL_TILE + 2 * DG_TILE + 2 * 16 * TL_NTMAX + 32;
}
static size_t tl_usize_g(int TS, int ldx_max) { return (size_t)2 * TS * ldx_max + (size_t)CI_STRIDE * 52 + 52 * CI_NCH + 32; }
static size_t tl_lds_doubles(int Kcmax, int TS, int ldx_max) {
  const size_t c = tl_usize_c(), g = tl_usize_g(TS, ldx_max);
  return (c > g ? c : g) + 4 * (size_t)Kcmax + 16 + 2 * 64 * CI_NCH + 16 * TL_NTMAX;
}

// the tile variant needs its LDS budget (which grows with Kmax through the solution vector) to fit
bool ls_tile_applicable(int Kcmax, int Nmax) {
  return Nmax <= 64 * CI_NCH && tl_lds_doubles(Kcmax, 32, 16 * TL_NTMAX + 16) * sizeof(double) <= TL_LDS_LIMIT;
}

size_t ls_tile_scratch_stride(int nmax, int Nmax) {
  const size_t Npad = (size_t)((Nmax + 63) >> 6) << 6;
  return (2 * Npad * nmax + 15) & ~(size_t)15;
}

// Frames into their size classes, and (adaptations >= 1) the zero counts of the tracks that the slot set-up of both
// batched kernels looks up.  A.cls (zeroed header) / A.zflag (zeroed) / A.zloc / A.ztot are set by the caller.
int launch_ls_prepass(eaqhm_ctx* ctx, LsArgs A, long long track_t0) {
  hipLaunchKernelGGL(eaqhm_ls_classify_kernel, dim3((A.n_frames + 255) / 256), dim3(256), 0, ctx->stream, A);
  HIP_TRY(ctx, hipGetLastError());
  if (A.mode == 1) {
    // only the chunks the resident track window overlaps (a time block of a long file is a small part of the file)
    const int ch0 = (int)(track_t0 >> 10), ch1 = (int)((track_t0 + A.Lt - 1) >> 10);
    hipLaunchKernelGGL(eaqhm_ls_zero_prefix_kernel, dim3(ch1 - ch0 + 1, A.Kmax), dim3(256), 0, ctx->stream, A.fm_cur, A.Lt,
                       track_t0, ch0, A.zchunks, A.zflag, (unsigned short*)A.zloc, (int*)A.ztot);
    HIP_TRY(ctx, hipGetLastError());
  }
  return EAQHM_OK;
}

// A.scratch / A.scratch_stride / A.zloc / A.ztot / A.cls / A.debug are set by the caller (eaqhm_ls_batch), after launch_ls_prepass.
// Returns the largest number of tile rows handled (frames with more are left to the caller's fallback).
int launch_ls_tile(eaqhm_ctx* ctx, LsArgs A, int grid) {
  const int Kcmax = A.Kcmax;
  const int ldx_max = 16 * TL_NTMAX + 16;
  const int TS = 32;
  const size_t lds_bytes = tl_lds_doubles(Kcmax, TS, ldx_max) * sizeof(double);
  if (lds_bytes > TL_LDS_LIMIT) return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: LDS budget exceeded (tile variant)");
  if (a0_lds_doubles(A0_M, 2, TZ_TB, TZ_NCH, 520, 64 * CI_NCH, Kcmax) * sizeof(double) > lds_bytes)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: LDS budget exceeded (adaptation 0, tile variant)");
  HIP_TRY(ctx, hipFuncSetAttribute((const void*)eaqhm_ls_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(eaqhm_ls_tile_kernel, dim3(grid), dim3(TL_THREADS), lds_bytes, ctx->stream, A, TS, ldx_max);
  HIP_TRY(ctx, hipGetLastError());
  return EAQHM_OK;
}

}  // namespace eaqhm
