// eaqhm_common.h — shared by the HIP translation units of libeaqhm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/eaqhm_hip.h"

#define EAQHM_ABI_VERSION 3

struct eaqhm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int n_cu = 0;
  int lds_bytes = 0;
  int clock_khz = 0;
  int ls_variant = 3;  // 2: MFMA Gramian + tile Cholesky through memory (any size), 3: all-on-chip tiles (+2 for big frames)
  int dbg_keep = 0;    // 1: in-kernel phase stamps on, accumulated across launches (diagnostics only)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  int* faults = nullptr;   // device counters: [0] LS systems whose Cholesky broke down (singular normal matrix),
                           // [1] diagonal pipelines whose hand-shake timed out (eaqhm_ls_chol.h: spin_until),
                           // [2] frames dropped because their window was not inside the resident tracks / the signal
  char err[512] = {0};

  int fail(int code, const char* msg) {
    snprintf(err, sizeof(err), "%s", msg);
    return code;
  }
  // grow-only scratch; growing synchronises the stream first (never happens in steady state)
  int reserve(size_t bytes) {
    if (bytes <= scratch_bytes) return EAQHM_OK;
    if (scratch) {
      if (hipStreamSynchronize(stream) != hipSuccess) return fail(EAQHM_EHIP, "scratch: stream sync failed");
      (void)hipFree(scratch);
      scratch = nullptr;
      scratch_bytes = 0;
    }
    if (hipMalloc(&scratch, bytes) != hipSuccess) {
      snprintf(err, sizeof(err), "scratch: hipMalloc(%zu bytes) failed", bytes);
      return EAQHM_ENOMEM;
    }
    scratch_bytes = bytes;
    return EAQHM_OK;
  }
};

#define HIP_TRY(ctx, call)                                                                        \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
               __FILE__, __LINE__);                                                               \
      return EAQHM_EHIP;                                                                          \
    }                                                                                             \
  } while (0)
