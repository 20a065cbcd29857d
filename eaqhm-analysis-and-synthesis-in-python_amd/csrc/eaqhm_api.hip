// eaqhm_api.hip — context life cycle of libeaqhm_hip.so.
#include "eaqhm_common.h"

extern "C" int eaqhm_ctx_create(eaqhm_ctx** out, int device) {
  if (!out) return EAQHM_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return EAQHM_EHIP;
  if (hipSetDevice(device) != hipSuccess) return EAQHM_EHIP;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) return EAQHM_EHIP;
  eaqhm_ctx* c = new eaqhm_ctx();
  c->device = device;
  c->n_cu = p.multiProcessorCount;
  c->lds_bytes = (int)p.sharedMemPerBlock;
  c->clock_khz = p.clockRate;
  if (hipMalloc((void**)&c->faults, 64) != hipSuccess || hipMemset(c->faults, 0, 64) != hipSuccess) {
    delete c;
    return EAQHM_ENOMEM;
  }
  *out = c;
  return EAQHM_OK;
}

extern "C" int eaqhm_ctx_destroy(eaqhm_ctx* ctx) {
  if (!ctx) return EAQHM_EINVAL;
  if (ctx->scratch) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->scratch);
  }
  if (ctx->faults) (void)hipFree(ctx->faults);
  delete ctx;
  return EAQHM_OK;
}

extern "C" int eaqhm_set_stream(eaqhm_ctx* ctx, void* hip_stream) {
  if (!ctx) return EAQHM_EINVAL;
  ctx->stream = (hipStream_t)hip_stream;
  return EAQHM_OK;
}

extern "C" int eaqhm_sync(eaqhm_ctx* ctx) {
  if (!ctx) return EAQHM_EINVAL;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EAQHM_OK;
}

extern "C" const char* eaqhm_last_error(eaqhm_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int eaqhm_ls_faults(eaqhm_ctx* ctx, int32_t h_count[3]) {
  if (!ctx || !h_count) return EAQHM_EINVAL;
  HIP_TRY(ctx, hipMemcpyAsync(h_count, ctx->faults, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ctx->faults, 0, 3 * sizeof(int32_t), ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return EAQHM_OK;
}

extern "C" int eaqhm_device_info(eaqhm_ctx* ctx, int32_t h_info[4]) {
  if (!ctx || !h_info) return EAQHM_EINVAL;
  h_info[0] = ctx->n_cu;
  h_info[1] = ctx->lds_bytes;
  h_info[2] = ctx->clock_khz;
  h_info[3] = EAQHM_ABI_VERSION;
  return EAQHM_OK;
}

extern "C" int eaqhm_set_option(eaqhm_ctx* ctx, int32_t key, int32_t value) {
  if (!ctx) return EAQHM_EINVAL;
  if (key == EAQHM_OPT_LS_VARIANT && (value == 2 || value == 3)) {
    ctx->ls_variant = value;
    return EAQHM_OK;
  }
  if (key == EAQHM_OPT_DEBUG_KEEP) {
    ctx->dbg_keep = value;
    if (value && ctx->scratch && ctx->scratch_bytes >= 256)
      (void)hipMemsetAsync((char*)ctx->scratch + ctx->scratch_bytes - 256 + 64, 0, 128, ctx->stream);
    return EAQHM_OK;
  }
  return ctx->fail(EAQHM_EINVAL, "eaqhm_set_option: unknown key or value");
}

extern "C" int eaqhm_debug_read(eaqhm_ctx* ctx, uint64_t h_out[16]) {
  if (!ctx || !h_out) return EAQHM_EINVAL;
  if (!ctx->scratch || ctx->scratch_bytes < 256) return ctx->fail(EAQHM_EINVAL, "eaqhm_debug_read: nothing to read");
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(h_out, (char*)ctx->scratch + ctx->scratch_bytes - 256 + 64, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return EAQHM_OK;
}
