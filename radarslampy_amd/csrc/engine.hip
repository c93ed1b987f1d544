// engine: placeholder until the batched lane pipeline lands (next commit)
#include "roam_internal.h"
extern "C" {
int32_t roam_engine_create(roam_ctx *ctx, const roam_engine_cfg *) { if (!ctx) return ROAM_E_ARG; ROAM_SET_ERR(ctx, "engine not built"); return ROAM_E_STATE; }
int32_t roam_engine_destroy(roam_ctx *ctx) { return ctx ? ROAM_OK : ROAM_E_ARG; }
int32_t roam_engine_upload_scan(roam_ctx *ctx, int32_t, const uint8_t *) { return ctx ? ROAM_E_STATE : ROAM_E_ARG; }
int32_t roam_engine_init_lane(roam_ctx *ctx, int32_t, int32_t, const float *, int32_t, const double *) { return ctx ? ROAM_E_STATE : ROAM_E_ARG; }
int32_t roam_engine_step(roam_ctx *ctx, const int32_t *) { return ctx ? ROAM_E_STATE : ROAM_E_ARG; }
int32_t roam_engine_results(roam_ctx *ctx, roam_lane_result *, int32_t) { return ctx ? ROAM_E_STATE : ROAM_E_ARG; }
int32_t roam_engine_lane_features(roam_ctx *ctx, int32_t, float *, int32_t, int32_t *) { return ctx ? ROAM_E_STATE : ROAM_E_ARG; }
int32_t roam_engine_lane_peaks(roam_ctx *ctx, int32_t, int32_t *, int64_t, int64_t *) { return ctx ? ROAM_E_STATE : ROAM_E_ARG; }
int32_t roam_engine_set_features(roam_ctx *ctx, int32_t, const float *, int32_t) { return ctx ? ROAM_E_STATE : ROAM_E_ARG; }
int32_t roam_engine_stage_times(roam_ctx *ctx, float *, const char **, int32_t, int32_t *) { return ctx ? ROAM_E_STATE : ROAM_E_ARG; }
int32_t roam_engine_time_kernel(roam_ctx *ctx, const char *, int32_t, float *, double *) { return ctx ? ROAM_E_STATE : ROAM_E_ARG; }
int32_t roam_doh_blobs(roam_ctx *ctx, const float *, int32_t, int32_t, double, double, int32_t, double, double, double *, int32_t, int32_t *) { if (!ctx) return ROAM_E_ARG; ROAM_SET_ERR(ctx, "doh not built"); return ROAM_E_STATE; }
}
