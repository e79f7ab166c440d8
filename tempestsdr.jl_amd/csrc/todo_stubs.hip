// TEMPORARY: entry points whose kernels are still being written (FFT family).  Each returns
// TSDR_EHIP loudly; this file is deleted as the real implementations land.
#include "common.h"
#define NOTYET(ctx, name) return tsdr::set_err((ctx), TSDR_EHIP, name ": not implemented yet")
extern "C" {
int tsdr_resampler_init(tsdr_ctx *ctx, size_t, int, tsdr_resampler **) { NOTYET(ctx, "resampler_init"); }
int tsdr_resampler_run(tsdr_resampler *, const float *, size_t, float *) { return TSDR_EHIP; }
int tsdr_resampler_run_d(tsdr_resampler *, const float *, size_t, float *) { return TSDR_EHIP; }
int tsdr_resampler_lpf(tsdr_resampler *, float *) { return TSDR_EHIP; }
void tsdr_resampler_free(tsdr_resampler *) {}
int tsdr_autocorr(tsdr_ctx *ctx, const float *, size_t, double, double, double, int, float *, size_t *) { NOTYET(ctx, "autocorr"); }
int tsdr_autocorr_d(tsdr_ctx *ctx, const float *, size_t, double, double, double, int, float *, size_t *) { NOTYET(ctx, "autocorr_d"); }
int tsdr_autocorr_iq_d(tsdr_ctx *ctx, const float *, size_t, double, double, double, int, float *, size_t *) { NOTYET(ctx, "autocorr_iq_d"); }
int tsdr_autocorr_partial_d(tsdr_ctx *ctx, const float *, int, size_t, size_t, size_t, size_t, float *) { NOTYET(ctx, "autocorr_partial_d"); }
int tsdr_autocorr_finish_d(tsdr_ctx *ctx, const float *, size_t, size_t, int, float *) { NOTYET(ctx, "autocorr_finish_d"); }
int tsdr_zoom_bounds(size_t, double, double, double, size_t *, size_t *) { return TSDR_EHIP; }
int tsdr_argmax_d(tsdr_ctx *ctx, const float *, size_t, size_t *, float *) { NOTYET(ctx, "argmax_d"); }
int tsdr_spectrum(tsdr_ctx *ctx, const float *, int, size_t, int, float *) { NOTYET(ctx, "spectrum"); }
int tsdr_spectrum_d(tsdr_ctx *ctx, const float *, int, size_t, int, float *) { NOTYET(ctx, "spectrum_d"); }
int tsdr_welch(tsdr_ctx *ctx, const float *, int, size_t, size_t, int, float *) { NOTYET(ctx, "welch"); }
int tsdr_welch_d(tsdr_ctx *ctx, const float *, int, size_t, size_t, int, float *) { NOTYET(ctx, "welch_d"); }
int tsdr_waterfall(tsdr_ctx *ctx, const float *, int, size_t, size_t, double *) { NOTYET(ctx, "waterfall"); }
int tsdr_waterfall_d(tsdr_ctx *ctx, const float *, int, size_t, size_t, double *) { NOTYET(ctx, "waterfall_d"); }
int tsdr_fft_c2c(tsdr_ctx *ctx, const float *, float *, size_t, size_t, int) { NOTYET(ctx, "fft_c2c"); }
int tsdr_fft_c2c_d(tsdr_ctx *ctx, const float *, float *, size_t, size_t, int) { NOTYET(ctx, "fft_c2c_d"); }
}
