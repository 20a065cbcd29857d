// eaqhm_ls.hip — per-frame weighted complex least squares of the eaQHM analysis, batched over frames.
// gfx950 (MI355X) only.  FP64 throughout: the reference is float64/complex128 (functions.py:420-535).
//
// One workgroup owns one frame at a time (frames pulled from per-class atomic queues) and runs four phases
// (eaqhm_ls_tile.hip for frames whose system fits the register file, eaqhm_ls_mfma.hip for larger ones):
//   A  basis      window the tracks of the active slots, bridge zero gaps, running-sum phases,
//                 E2 = (am+eps)/(am_mid+eps) * exp(j*2*pi*F/fs); negative/DC/positive column layout
//                 (functions.py:244-292, :508-519; adaptation 0: functions.py:444-455)
//   B  Gramian    G_p = E2^H diag(w^2 n^p) E2, p = 0,1,2, and the right-hand sides, which come for
//                 free by appending the signal window as one more basis column (functions.py:457-464)
//   C  solve      Cholesky of [[G0,G1],[G1,G2]] with the RHS carried as an extra row (forward solve
//                 for free), then back substitution (replaces inv(R) @ arr, functions.py:465 / :530)
//   D  epilogue   frequency mismatch, acceptance, frame-centre records (functions.py:297-324)
//
// Column order of the LS unknowns: [n negative columns | DC | n positive columns], Kc = 2n+1, then the
// same again for the slopes (functions.py:455 / :519: E = [E2, n*E2]).
#include "eaqhm_ls_common.h"

namespace eaqhm {
int launch_ls_mfma(eaqhm_ctx* ctx, LsArgs A, int grid, int min_nb);  // eaqhm_ls_mfma.hip
size_t ls_mfma_scratch_stride(int nmax, int Nmax, int Kcmax);
int launch_ls_tile(eaqhm_ctx* ctx, LsArgs A, int grid);              // eaqhm_ls_tile.hip
int launch_ls_prepass(eaqhm_ctx* ctx, LsArgs A, long long track_t0);                    // eaqhm_ls_tile.hip (classes, zero counts)
size_t ls_tile_scratch_stride(int nmax, int Nmax);
bool ls_tile_applicable(int Kcmax, int Nmax);


// ------------------------------------------------------------------------------------------------
// explicit-matrix seam, phase B: Gramian of the augmented basis -> transposed system matrix Lt (row k holds column k of R,
// entries i = k..M, row stride ldl complex).  One thread per (a, b) pair, a >= b, of the C1 columns.
__device__ void gram_to_system(const double* __restrict__ Xre, const double* __restrict__ Xim, int N, int Kc,
                               int ldx, const double* __restrict__ ww, double mid, double* __restrict__ Lt,
                               int ldl) {
  const int C1 = Kc + 1;
  const int M = 2 * Kc;
  const int npairs = C1 * (C1 + 1) / 2;
  for (int p = threadIdx.x; p < npairs; p += blockDim.x) {
    int a = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
    while ((a + 1) * (a + 2) / 2 <= p) ++a;
    while (a * (a + 1) / 2 > p) --a;
    int b = p - a * (a + 1) / 2;
    if (a == Kc && b == Kc) continue;
    double g0r = 0, g0i = 0, g1r = 0, g1i = 0, g2r = 0, g2i = 0;
    for (int t = 0; t < N; ++t) {
      double ar = Xre[(size_t)t * ldx + a], ai = Xim[(size_t)t * ldx + a];
      double br = Xre[(size_t)t * ldx + b], bi = Xim[(size_t)t * ldx + b];
      double pr = ar * br + ai * bi;  // conj(xa) * xb
      double pi = ar * bi - ai * br;
      double w0 = ww[t], nn = (double)t - mid, w1 = w0 * nn, w2 = w1 * nn;
      g0r += w0 * pr; g0i += w0 * pi;
      g1r += w1 * pr; g1i += w1 * pi;
      g2r += w2 * pr; g2i += w2 * pi;
    }
    // G_p[a][b] = sum conj(x_a) x_b.  R = [[G0, G1], [G1, G2]] (Hermitian); Lt[col][row] = R[row][col].
    auto put = [&](int row, int col, double re, double im) {
      size_t o = ((size_t)col * ldl + row) * 2;
      Lt[o] = re; Lt[o + 1] = im;
    };
    if (a < Kc) {
      put(a, b, g0r, g0i);
      put(Kc + a, Kc + b, g2r, g2i);
      put(Kc + a, b, g1r, g1i);
      if (a != b) put(Kc + b, a, g1r, -g1i);
    } else {
      // signal row: entry = sum ww s x_b = conj(rhs_b); the augmented row M holds conj(rhs)^T
      put(M, b, g0r, g0i);
      put(M, Kc + b, g1r, g1i);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// adaptation >= 1 frame set-up (functions.py:202-213): active slots + empty-row seeding flags
// One wave per frame: lanes test 64 slots at a time, the ballot gives count and compacted positions.
extern "C" __global__ void __launch_bounds__(256) eaqhm_frame_prep_kernel(const double* fm_cur /* biased by -track_t0 */, long long L /* row stride */, int Kmax,
                                                                          const int* frame_c, int n_frames, int* ncol,
                                                                          int* cols, unsigned char* seeded, int* any_seed) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (f >= n_frames) return;
  const int c = frame_c[f];
  int n = 0;
  for (int k0 = 0; k0 < Kmax; k0 += 64) {
    const int k = k0 + lane;
    const bool nz = (k < Kmax) && (fm_cur[(size_t)k * L + c] != 0.0);
    const unsigned long long m = __ballot(nz);
    if (nz) cols[(size_t)f * Kmax + n + __popcll(m & ((1ull << lane) - 1ull))] = k;
    n += __popcll(m);
  }
  if (lane == 0) {
    if (n == 0) {
      seeded[c] = 1;
      cols[(size_t)f * Kmax] = 0;
      n = 1;
      atomicOr(any_seed, 1);
    }
    ncol[f] = n;
  }
}

// ------------------------------------------------------------------------------------------------
// explicit-matrix seam (one frame): eaqhmLS_complexamps / iqhmLS_complexamps as Python-level functions
struct LsExplicitArgs {
  const double* s; int N; const double* am; const double* fm; const double* f0range; int Kc;
  const double* window; double fs; double* out_amp; double* out_slope; double* scratch; int* fault;
};

extern "C" __global__ void __launch_bounds__(256) eaqhm_ls_explicit_kernel(LsExplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int N = A.N, Kc = A.Kc, C1 = Kc + 1, M = 2 * Kc, ldx = C1, ldl = M + 1;
  double* Xre = A.scratch;
  double* Xim = Xre + (size_t)N * C1;
  double* Lt = Xim + (size_t)N * C1;
  double* ww = lds;
  double* rowj = ww + N;
  double* xs = rowj + 2 * M;
  double* sh = xs + 2 * M;
  const double eps = 10e-5;
  const bool eaqhm_mode = (A.fm != nullptr);
  // functions.py:446 / :503: midlen = (len-1)/2 (float) for iqhm, int((len-1)/2) for eaqhm
  const double midf = eaqhm_mode ? (double)((N - 1) / 2) : 0.5 * (double)(N - 1);
  const int midi = (N - 1) / 2;
  for (int u = tid; u < N; u += nt) {
    double w = A.window[u];
    ww[u] = w * w;
    Xre[(size_t)u * ldx + Kc] = A.s[u];
    Xim[(size_t)u * ldx + Kc] = 0.0;
  }
  for (int j = tid; j < Kc; j += nt) {
    if (eaqhm_mode) {
      double fan_mid = 0.0, acc = 0.0;
      for (int u = 0; u <= midi; ++u) acc += A.fm[(size_t)u * Kc + j];
      fan_mid = acc;
      const double amid = A.am[(size_t)midi * Kc + j] + eps;
      acc = 0.0;
      for (int u = 0; u < N; ++u) {
        acc += A.fm[(size_t)u * Kc + j];  // lfilter([1],[1,-1]) == running sum (functions.py:510)
        double sn, cs;
        sincos((2.0 * M_PI * (acc - fan_mid)) / A.fs, &sn, &cs);
        double rr = (eps + A.am[(size_t)u * Kc + j]) / amid;
        Xre[(size_t)u * ldx + j] = rr * cs;
        Xim[(size_t)u * ldx + j] = rr * sn;
      }
    } else {
      const double fk = A.f0range[j];
      for (int u = 0; u < N; ++u) {
        double nn = (double)u - midf;
        double sn, cs;
        sincos((nn * 2.0 * M_PI * fk) / A.fs, &sn, &cs);
        Xre[(size_t)u * ldx + j] = cs;
        Xim[(size_t)u * ldx + j] = sn;
      }
    }
  }
  __syncthreads();
  gram_to_system(Xre, Xim, N, Kc, ldx, ww, midf, Lt, ldl);
  __syncthreads();
  cholesky_solve(Lt, M, ldl, rowj, xs, sh, A.fault);
  for (int q = tid; q < 2 * Kc; q += nt) {
    A.out_amp[q] = xs[q];
    A.out_slope[q] = xs[2 * Kc + q];
  }
}

}  // namespace eaqhm

// ================================================================================================
// C ABI
using namespace eaqhm;

extern "C" int eaqhm_frame_prep(eaqhm_ctx* ctx, const double* fm_cur, int64_t L, int64_t track_t0, int64_t track_len,
                                int32_t Kmax, const int32_t* frame_c, int32_t n_frames, int32_t* ncol, int32_t* cols,
                                uint8_t* seeded, int32_t* any_seed) {
  if (!ctx) return EAQHM_EINVAL;
  if (!fm_cur || !frame_c || !ncol || !cols || !seeded || !any_seed || L <= 0 || Kmax <= 0 || n_frames < 0 ||
      track_t0 < 0 || track_len <= 0 || track_t0 + track_len > L)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_frame_prep: bad argument");
  HIP_TRY(ctx, hipMemsetAsync(seeded, 0, (size_t)L, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(any_seed, 0, sizeof(int32_t), ctx->stream));
  if (n_frames == 0) return EAQHM_OK;
  hipLaunchKernelGGL(eaqhm_frame_prep_kernel, dim3((n_frames + 3) / 4), dim3(256), 0, ctx->stream, fm_cur - track_t0,
                     (long long)track_len, Kmax, frame_c, n_frames, ncol, cols, seeded, any_seed);
  HIP_TRY(ctx, hipGetLastError());
  return EAQHM_OK;
}

extern "C" int eaqhm_ls_batch(eaqhm_ctx* ctx, int32_t mode, const double* s, int64_t L, double fs,
                              const double* am_cur, const double* fm_cur, int64_t track_t0, int64_t track_len,
                              int32_t Kmax, const int32_t* frame_inst,
                              const int32_t* frame_c, const int32_t* frame_wl, const double* frame_f0,
                              const int32_t* frame_K, const int32_t* ncol, const int32_t* cols,
                              const uint8_t* seeded, const int32_t* any_seed, int32_t n_frames, int32_t wl_max,
                              int32_t a_iter, double f0_stale, double f0min, double* records, double* raw_amp,
                              double* raw_slope) {
  if (!ctx) return EAQHM_EINVAL;
  if (mode != 0 && mode != 1) return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: mode must be 0 or 1");
  if (!s || L <= 0 || fs <= 0 || Kmax <= 0 || !frame_inst || !frame_c || !frame_wl || !records ||
      n_frames < 0)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: bad argument");
  if (mode == 0 && (!frame_f0 || !frame_K)) return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: mode 0 needs frame_f0/frame_K");
  if (mode == 1 && (!am_cur || !fm_cur || !ncol || !cols || !seeded || !any_seed))
    return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: mode 1 needs tracks and eaqhm_frame_prep outputs");
  if (mode == 1 && (track_t0 < 0 || track_len <= 0 || track_t0 + track_len > L))
    return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: track window outside the signal");
  if (mode == 0) { track_t0 = 0; track_len = L; }
  if ((raw_amp == nullptr) != (raw_slope == nullptr)) return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: raw_amp/raw_slope");
  if (n_frames == 0) return EAQHM_OK;
  if (wl_max <= 0) return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: wl_max must be positive");
  const int nmax = Kmax, Nmax = 2 * wl_max + 1, Kcmax = 2 * Kmax + 1;
  LsArgs B;
  B.mode = mode; B.s = s; B.L = L; B.fs = fs; B.Kmax = Kmax;
  // tracks (and the zero counts below) biased by the window's first sample: the kernels index with absolute samples.
  // Every frame window must lie inside [track_t0, track_t0 + track_len): the caller's contract; a frame that breaks it is
  // dropped by the classification kernel and counted (fault[2]) instead of reading outside the buffers.
  B.am_cur = am_cur ? am_cur - track_t0 : nullptr; B.fm_cur = fm_cur ? fm_cur - track_t0 : nullptr; B.Lt = track_len; B.trk_t0 = track_t0;
  B.frame_inst = frame_inst; B.frame_c = frame_c; B.frame_wl = frame_wl; B.frame_f0 = frame_f0; B.frame_K = frame_K;
  B.ncol = ncol; B.cols = cols; B.seeded = seeded; B.any_seed = any_seed; B.n_frames = n_frames; B.a_iter = a_iter;
  B.f0_stale = f0_stale; B.f0min = f0min; B.records = records; B.raw_amp = raw_amp; B.raw_slope = raw_slope;
  B.nmax = nmax; B.Nmax = Nmax; B.Kcmax = Kcmax; B.fault = ctx->faults;
  int grid = ctx->n_cu < n_frames ? ctx->n_cu : n_frames;
  const size_t st_m = ls_mfma_scratch_stride(nmax, Nmax, Kcmax), st_t = ls_tile_scratch_stride(nmax, Nmax);
  const size_t frame_bytes = (st_m > st_t ? st_m : st_t) * grid * sizeof(double);
  const int zchunks = (int)((L + 1023) >> 10);
  const size_t zloc_bytes = (mode == 1) ? ((((size_t)Kmax * track_len * sizeof(unsigned short)) + 255) & ~(size_t)255) : 256;
  const size_t ztot_bytes = (((size_t)Kmax * zchunks * sizeof(int)) + 255) & ~(size_t)255;
  const size_t flag_bytes = zloc_bytes + ztot_bytes + (((size_t)zchunks + 255) & ~(size_t)255);
  const size_t cls_bytes = ((16 + (size_t)LS_NCLS * n_frames) * sizeof(int) + 255) & ~(size_t)255;
  int rc = ctx->reserve(frame_bytes + flag_bytes + cls_bytes + 256);
  if (rc) return rc;
  B.zloc = (const unsigned short*)((char*)ctx->scratch + frame_bytes) - track_t0;
  B.ztot = (const int*)((char*)ctx->scratch + frame_bytes + zloc_bytes);
  B.zchunks = zchunks;
  B.zflag = (unsigned char*)ctx->scratch + frame_bytes + zloc_bytes + ztot_bytes;
  {   // chunk flags and the class header right behind them: one clear for both
    const size_t zflag_bytes = flag_bytes - zloc_bytes - ztot_bytes;
    HIP_TRY(ctx, hipMemsetAsync(B.zflag, 0, zflag_bytes + 16 * sizeof(int), ctx->stream));
  }
  B.cls = (int*)((char*)ctx->scratch + frame_bytes + flag_bytes);
  int* counters = (int*)((char*)ctx->scratch + ctx->scratch_bytes - 256);
  const bool tile_path = ctx->ls_variant == 3 && ls_tile_applicable(Kcmax, Nmax);
  if (!tile_path)   // the frame queue of the large-frame kernel when it takes every frame
    HIP_TRY(ctx, hipMemsetAsync(counters, 0, 8 * sizeof(int), ctx->stream));
  B.debug = ctx->dbg_keep ? (unsigned long long*)(counters + 16) : nullptr;
  B.debug_diag = ctx->dbg_keep >= 2;
  B.scratch = (double*)ctx->scratch;
  int min_nb = 0;
  rc = launch_ls_prepass(ctx, B, track_t0);
  if (rc) return rc;
  if (tile_path) {   // small frames: everything in registers/LDS; the rest falls through
    B.scratch_stride = st_t; B.work_counter = counters + 2;
    rc = launch_ls_tile(ctx, B, grid);
    if (rc) return rc;
    min_nb = 1;
    if ((2 * Kcmax + 1 + 15) / 16 <= 13) return EAQHM_OK;   // no frame can be larger: skip the second launch
  }
  B.scratch_stride = st_m; B.work_counter = counters + 1;
  return launch_ls_mfma(ctx, B, grid, min_nb);
}

extern "C" int eaqhm_ls_explicit(eaqhm_ctx* ctx, const double* s, int32_t N, const double* am, const double* fm,
                                 const double* f0range, int32_t Kc, const double* window, double fs, double* out_amp,
                                 double* out_slope) {
  if (!ctx) return EAQHM_EINVAL;
  if (!s || N < 2 || Kc < 1 || !window || fs <= 0 || !out_amp || !out_slope)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_explicit: bad argument");
  if (fm) {
    if (!am) return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_explicit: fm given without am");
    if ((N & 1) == 0) return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_explicit: eaqhm seam needs an odd window length");
  } else if (!f0range) return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_explicit: need fm or f0range");
  const size_t C1 = (size_t)Kc + 1, M = 2 * (size_t)Kc;
  size_t doubles = 2 * (size_t)N * C1 + 2 * M * (M + 1);
  int rc = ctx->reserve(doubles * sizeof(double));
  if (rc) return rc;
  LsExplicitArgs A{s, N, am, fm, f0range, Kc, window, fs, out_amp, out_slope, (double*)ctx->scratch, ctx->faults};
  size_t lds_bytes = ((size_t)N + 4 * M + 8) * sizeof(double);
  if (lds_bytes > 160 * 1024) return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_explicit: problem too large");
  HIP_TRY(ctx, hipFuncSetAttribute((const void*)eaqhm_ls_explicit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(eaqhm_ls_explicit_kernel, dim3(1), dim3(256), lds_bytes, ctx->stream, A);
  HIP_TRY(ctx, hipGetLastError());
  return EAQHM_OK;
}
