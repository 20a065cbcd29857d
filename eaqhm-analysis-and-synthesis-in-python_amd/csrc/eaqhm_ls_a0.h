// eaqhm_ls_a0.h — adaptation 0 as TWO REAL systems, entirely on chip (used by both batched kernels).
//
// At adaptation 0 the signal is real, the columns come in conjugate pairs and the window is symmetric, so the
// solution has a_{-k} = conj(a_k), b_{-k} = conj(b_k), and in the real basis
//     even:  cos(k theta n) (k = 0..K),  n sin(k theta n) (k = 1..K)        odd:  sin(k theta n) (k = 1..K),  n cos(k theta n) (k = 0..K)
// the weighted normal equations fall apart into two REAL symmetric systems of order Kc = 2K+1 (even functions against
// odd ones sum to zero under the symmetric weight).  Their entries come from the same Toeplitz tables
// (cos a cos b = (cos(a-b) + cos(a+b))/2 etc.; toeplitz_tables in eaqhm_ls_common.h), and
//     a_0 = u_0,  a_k = (u_k - j v_k)/2,   b_0 = p_0,  b_k = (p_k - j q_k)/2      (u: cos, q: n sin | v: sin, p: n cos).
// Half the order means half the chain of dependent diagonal steps, a quarter of the tiles, real arithmetic (one MFMA
// where the complex code needs three).  Each system (right-hand side as its last row) is factorised in the register
// file: right-looking tile Cholesky with look-ahead, 16x16 real tiles in MFMA accumulator layout, the diagonal tiles
// by a two-wave pipeline (diag_Dr / diag_Zr, eaqhm_ls_chol.h), panels published in LDS.
//   PAR = 2 (frames of the tile kernel, <= 7 tile rows): the two systems side by side, waves 0-3 the even one, waves
//            4-7 the odd one, sharing the barriers;
//   PAR = 1 (large frames, <= 19 tile rows): all eight waves on one system, the two one after the other.
#pragma once
#include "eaqhm_ls_common.h"
#include "eaqhm_ls_chol.h"

namespace eaqhm {

// Entries (gi0 + 4 r, gj), r = 0..3 — the four values of one lane of a 16x16 tile — of system `sys` (0 even, 1 odd), order
// Kc + 1 with the right-hand side as row / column Kc.  Branch-free: every entry is  0.5 (s1 T[i1] + s2 T[i2])  of one
// table, or one table value (right-hand side), or a constant (padding); the lanes of a tile differ in which, so the case
// distinctions are selects, all the table reads of the four entries are in flight together, and the constants come last.
// (Round 2's version was a branchy function per entry: 28 calls per lane of the 7-tile budget, each with its own exposed
// LDS round trips and divergent paths — a fifth of the adaptation-0 launch.)
__device__ __attribute__((noinline)) d4 a0_entry4(const double* tab, int TB_, double ssq, int sys_, int gi0, int gj, int K_, int Kc_) {
  const int TB = uni(TB_), sys = uni(sys_), K = uni(K_), Kc = uni(Kc_);   // wave-uniform: the case distinctions on them are scalar
  // (integer arithmetic instead of selects: the compiler turns nested selects into exec-mask branches)
  const int split = K + 1 - sys;
  const int tj = (gj >= split) ? 1 : 0;
  const int l = gj - tj * K + sys * (1 - tj);      // harmonic number: even: k | k (n sin k), odd: k + 1 (sin) | k (n cos)
  const int sg = 1 - 2 * sys;
  int i1[4], i2[4];
  double s2[4];
  bool plain[4], rhs[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = gi0 + 4 * r;
    const int ti = (gi >= split) ? 1 : 0;
    const int k = gi - ti * K + sys * (1 - ti);
    const int df = k - l, d = (df < 0) ? -df : df, sm = k + l;
    // mixed products: harmonic of the plain function kp, of the n-times one lp; signed difference dd
    const int kp = k + ti * (l - k), lp = l + ti * (k - l);
    const int dd = sg * (lp - kp), ad = (dd < 0) ? -dd : dd;
    const int same = 1 - (ti ^ tj);
    // table: c0 (both plain), c2 (both n-times), s1 (mixed: s1[sm] + sgn(dd) s1[|dd|], s1[0] = 0)
    const int base = same * ti * 2 * TB + (1 - same) * TB;
    const int e = (gi == Kc) ? gj : gi;                      // the unknown a right-hand-side entry belongs to
    // sum w^2 s f(n):  even: Re r0[k] | Im r1[k],  odd: Im r0[k] | Re r1[k]
    const int hi = (e >= split) ? 1 : 0;                     // even: e > K, odd: e >= K
    const int rq = (3 + sys + hi * (3 - 2 * sys)) * TB + e - hi * K + sys * (1 - hi);
    const bool inside = (gi < Kc) && (gj < Kc);
    const bool isr = !inside && gi <= Kc && gj <= Kc && (gi != gj);
    i1[r] = inside ? (base + sm + same * (d - sm)) : (isr ? rq : 0);
    i2[r] = inside ? (base + ad + same * (sm - ad)) : 0;
    // cos cos | sin sin: c0[d] +- c0[sm];  n sin n sin | n cos n cos: c2[d] -+ c2[sm];  mixed: s1[sm] + sgn(dd) s1[|dd|]
    const int sig = same * (1 - 2 * (sys ^ ti)) + (1 - same) * ((dd < 0) ? -1 : 1);
    s2[r] = (double)sig;
    plain[r] = inside;
    rhs[r] = isr;
  }
  double v1[4], v2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { v1[r] = tab[i1[r]]; v2[r] = tab[i2[r]]; }
  d4 out;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gi = gi0 + 4 * r;
    const double gen = 0.5 * (v1[r] + s2[r] * v2[r]);
    const double pad = (gi > Kc || gj > Kc) ? ((gi == gj) ? 1.0 : 0.0) : ssq;   // identity padding behind the right-hand side | (Kc, Kc)
    out[r] = plain[r] ? gen : (rhs[r] ? v1[r] : pad);
  }
  return out;
}

// LDS doubles a0_frame<., M, PAR> needs (TB: table stride >= 2 K + 2, NCH: chunks of the table sums, WP >= wl + 1,
// NP >= N, Kcmax: for the solution vector)
__host__ __device__ inline size_t a0_lds_doubles(int M, int PAR, int TB, int NCH, int WP, int NP, int Kcmax) {
  const size_t tabs = (size_t)TZ_NQ * TB, parts = (size_t)NCH * TZ_NQ * TB, tiles = (size_t)2 * PAR * M * TL_TILE;
  // PAR = 2: tables and partial sums are dead when the tiles are filled (panels over them); PAR = 1: the tables
  // stay for the second system, only the partial sums are covered
  const size_t head = (PAR == 2) ? ((tabs + parts > tiles) ? tabs + parts : tiles) : tabs + ((parts > tiles) ? parts : tiles);
  return head + 3 * (size_t)WP + 2 * (size_t)NP + (size_t)PAR * (TL_TILE + 256 + 64 + 64 + 16 * M + 16 + 16 * M) +
         2 * 16 * (size_t)M + 16 + 2 + 4 * (size_t)Kcmax;
}

// One frame.  NS: tiles per wave (>= ceil(M (M+1) / 2 / waves per system)), M: tile rows the LDS is laid out for.
template <int NS, int M, int PAR>
__device__ __attribute__((noinline)) void a0_frame(const LsArgs& A, double* lds_, int f_, int TB_, int NCH_, int WP_, int NP_) {
  constexpr int NTHR = 512, WPS = 8 / PAR;   // waves per system
  const int f = uni(f_), TB = uni(TB_), NCH = uni(NCH_), WP = uni(WP_), NP = uni(NP_);
  const int tid = threadIdx.x, lane = tid & 63, lq = lane >> 4, lcol = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wv = wave & (WPS - 1);
  double* lds = uni(lds_);
  const int K = uni(A.frame_K[f]), Kc = 2 * K + 1, m = (Kc + 1 + 15) >> 4, is = Kc - 16 * (m - 1), nts = m * (m + 1) / 2;
  const int c = uni(A.frame_c[f]), wl = uni(A.frame_wl[f]), inst = uni(A.frame_inst[f]);
  const int N = 2 * wl + 1;
  const double f0 = uni(A.frame_f0[f]);
  // ---- LDS
  const size_t tabs = (size_t)TZ_NQ * TB, parts = (size_t)NCH * TZ_NQ * TB, tiles = (size_t)2 * PAR * M * TL_TILE;
  double* tab = lds;                                   // [TZ_NQ][TB]
  double* part = tab + tabs;                           // [NCH][TZ_NQ][TB]
  double* Pan = (PAR == 2) ? lds : part;               // [PAR][M][TL_TILE]  panels (over dead tables / partial sums)
  double* Wt = Pan + (size_t)PAR * M * TL_TILE;        // [PAR][M][TL_TILE]  inverses of the diagonal tiles
  const size_t head = (PAR == 2) ? ((tabs + parts > tiles) ? tabs + parts : tiles) : tabs + ((parts > tiles) ? parts : tiles);
  double* W2 = lds + head;                             // [wl+1] each
  double* PA = W2 + WP;
  double* PB = PA + WP;
  double* win = PB + WP;                               // [N]
  double* sig = win + NP;                              // [N]
  double* Ldl = sig + NP;                              // [PAR][TL_TILE]
  double* post = Ldl + PAR * TL_TILE;                  // [PAR][8 * 32]
  double* dumpD = post + PAR * 256;                    // [PAR][64]
  double* zs = dumpD + PAR * 64;                       // [PAR][64]
  double* zv = zs + PAR * 64;                          // [PAR][16 * M]
  double* xv = zv + PAR * 16 * M;                      // [PAR][16]
  double* dorig = xv + PAR * 16;                       // [PAR][16 * M]
  double* solv = dorig + PAR * 16 * M;                 // [2][16 * M]   solutions of the two systems
  double* sh = solv + 2 * 16 * M;                      // 16
  int* flags = (int*)(sh + 16);                        // [PAR] step counters of the diagonal pipelines
  double* xs = sh + 18;                                // 4 * Kcmax
  unsigned long long* dbg = uni(A.debug);
  unsigned long long t_prev = 0;
#define A0_STAMP(ph)                                                \
  do {                                                              \
    if (dbg && tid == 0) {                                          \
      unsigned long long t_now = __builtin_amdgcn_s_memtime();      \
      atomicAdd(dbg + (ph), t_now - t_prev);                        \
      t_prev = t_now;                                               \
    }                                                               \
  } while (0)
  if (dbg && tid == 0) t_prev = __builtin_amdgcn_s_memtime();

  for (int t = tid; t < N; t += NTHR) {
    win[t] = window_value(1, t, N);
    sig[t] = uni(A.s)[(size_t)(c - wl) + t];
  }
  __syncthreads();
  A0_STAMP(0);
  toeplitz_tables(tab, part, W2, PA, PB, sh, win, sig, K, wl, f0 * (2.0 * M_PI / uni(A.fs)), tid, TB, NCH);
  const double ssq = sh[0];
  A0_STAMP(1);

  // this wave's tiles of its system (numbered column by column, tile y on wave y % WPS, slot y / WPS)
  int tP[NS], tQ[NS];
  bool live[NS];
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) {
    const int y = sl * WPS + wv;
    live[sl] = y < nts;
    int P = 0, Q = 0;
    if (live[sl]) {
      int start = 0;
      while (Q + 1 < m && start + (m - Q) <= y) { start += m - Q; ++Q; }
      P = Q + (y - start);
    }
    tP[sl] = __builtin_amdgcn_readfirstlane(P);
    tQ[sl] = __builtin_amdgcn_readfirstlane(Q);
  }

  for (int round = 0; round < 2 / PAR; ++round) {
    const int sys = (PAR == 2) ? (wave >> 2) : round;   // which system this wave works on
    const int sp = (PAR == 2) ? sys : 0;                // ... and which set of work areas
    d4 acc[NS];
    if (tid < PAR) flags[tid] = 0;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      acc[sl] = live[sl] ? a0_entry4(tab, TB, ssq, sys, 16 * tP[sl] + lq, 16 * tQ[sl] + lcol, K, Kc) : (d4){0, 0, 0, 0};
      if (live[sl] && tP[sl] == tQ[sl]) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (lq + 4 * r == lcol) dorig[sp * 16 * M + 16 * tP[sl] + lcol] = acc[sl][r];
      }
    }
    __syncthreads();   // (PAR = 2: the tables are dead, their space becomes panel / inverse storage)
    A0_STAMP(2);

#define A0_UPDATE(sl)                                                                   \
  {                                                                                     \
    const double* ar = Pan + (sp * M + tP[sl]) * TL_TILE;                               \
    const double* br = Pan + (sp * M + tQ[sl]) * TL_TILE;                               \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                  \
      const int o = (4 * ks + lq) * TL_LD + lcol;                                       \
      acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[o], -br[o], acc[sl], 0, 0, 0);  \
    }                                                                                   \
  }
    for (int jb = 0; jb < m; ++jb) {
      const int yd = jb * m - jb * (jb - 1) / 2;   // the diagonal tile of this stage
      d4 Rt = (d4){0, 0, 0, 0};
      bool mine = false;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        if (!live[sl] || tP[sl] != jb || tQ[sl] != jb) continue;
        if (jb > 0) A0_UPDATE(sl)
        Rt = acc[sl];
        mine = true;
      }
      if (mine)
        diag_Dr(Rt, post + sp * 256, flags + sp, 16 * jb, dumpD + sp * 64, Ldl + sp * TL_TILE, jb == m - 1);
      A0_STAMP(10);
      if (jb > 0) {
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
          if (!live[sl] || tQ[sl] < jb || (tP[sl] == jb && tQ[sl] == jb)) continue;
          A0_UPDATE(sl)
        }
      }
      // the helper wave builds the inverse after its own trailing tiles (as in eaqhm_ls_tile_kernel: diag_Dr's posts wait in LDS)
      if (!mine && wv == ((yd + 1) & (WPS - 1)))
        diag_Zr(post + sp * 256, flags + sp, 16 * jb, zs + sp * 64, Wt + (sp * M + jb) * TL_TILE,
                dorig + sp * 16 * M + 16 * jb, (jb == m - 1) ? is : 16, uni(A.fault));
      A0_STAMP(8);
      __syncthreads();  // (A)
      A0_STAMP(6);
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {   // panel tiles: X = T W^T, published k-major
        if (!live[sl] || tQ[sl] != jb || tP[sl] == jb) continue;
        double* tr = Pan + (sp * M + tP[sl]) * TL_TILE;
#pragma unroll
        for (int r = 0; r < 4; ++r) tr[lcol * TL_LD + lq + 4 * r] = acc[sl][r];
        __builtin_amdgcn_wave_barrier();
        const double* wt = Wt + (sp * M + jb) * TL_TILE;
        d4 x = (d4){0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int o = (4 * ks + lq) * TL_LD + lcol;
          x = __builtin_amdgcn_mfma_f64_16x16x4f64(tr[o], wt[o], x, 0, 0, 0);
        }
        acc[sl] = x;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) tr[lcol * TL_LD + lq + 4 * r] = x[r];
      }
      A0_STAMP(7);
      __syncthreads();  // (C)
    }
#undef A0_UPDATE
    A0_STAMP(3);

    // ---- back substitution  L^T x = y,  y = row `is` of the last tile row
    double* zvs = zv + sp * 16 * M;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      if (!live[sl] || tP[sl] != m - 1 || tQ[sl] == m - 1) continue;
      if (lq == (is & 3)) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r == (is >> 2)) zvs[16 * tQ[sl] + lcol] = acc[sl][r];
      }
    }
    if (wv == 0 && lane < 16) zvs[16 * (m - 1) + lane] = (lane < is) ? Ldl[sp * TL_TILE + is * TL_LD + lane] : 0.0;
    __syncthreads();
    for (int P = m - 1; P >= 0; --P) {
      {   // x_P = W_PP^T z_P: 256 threads of the system's waves, thread (i, k) one term of row i
        const int ts = tid & (64 * WPS - 1), i = (ts >> 4) & 15, k = ts & 15;
        double x = (ts < 256 && k >= i) ? Wt[(sp * M + P) * TL_TILE + i * TL_LD + k] * zvs[16 * P + k] : 0.0;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if (ts < 256 && k == 0) {
          xv[sp * 16 + i] = x;
          solv[sys * 16 * M + 16 * P + i] = x;
        }
      }
      __syncthreads();
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        if (!live[sl] || tP[sl] != P || tQ[sl] == P) continue;
        double sr = 0;   // sum_i L[i][j] x[i] over this lane's rows i = lq + 4r
#pragma unroll
        for (int r = 0; r < 4; ++r) sr += acc[sl][r] * xv[sp * 16 + lq + 4 * r];
        sr += __shfl_xor(sr, 16);
        sr += __shfl_xor(sr, 32);
        if (lq == 0) zvs[16 * tQ[sl] + lcol] -= sr;
      }
      __syncthreads();
    }
    A0_STAMP(4);
  }
  // ---- back to the complex amplitudes and slopes in the order of the complex code: [negative | DC | positive]
  {
    const double* ev = solv;                      // u_0..u_K, q_1..q_K
    const double* od = solv + 16 * M;             // v_1..v_K, p_0..p_K
    for (int col = tid; col < Kc; col += NTHR) {
      const int h = (col < K) ? -(col + 1) : (col - K), k = (h < 0) ? -h : h;
      double ar, ai, br, bi;
      if (k == 0) { ar = ev[0]; ai = 0.0; br = od[K]; bi = 0.0; }
      else {
        ar = 0.5 * ev[k]; ai = -0.5 * od[k - 1];
        br = 0.5 * od[K + k]; bi = -0.5 * ev[K + k];
        if (h < 0) { ai = -ai; bi = -bi; }
      }
      xs[2 * col] = ar; xs[2 * col + 1] = ai;
      xs[2 * (Kc + col)] = br; xs[2 * (Kc + col) + 1] = bi;
    }
  }
  __syncthreads();
  write_record(A, xs, sh, nullptr, f, K, inst, c, f0, false);
  A0_STAMP(5);
#undef A0_STAMP
}

}  // namespace eaqhm
