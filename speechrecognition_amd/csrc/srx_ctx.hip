// Context, memory plumbing and host-side constant builders of libsrx.so.
#include <cmath>

#include "srx_common.h"

char g_srx_create_err[512] = {0};

extern "C" const char *srx_version(void) { return "srx 0.1 (gfx950)"; }

extern "C" int srx_ctx_create(int device, void *stream, srx_ctx **out) {
    if (!out) return srx_fail(nullptr, SRX_ERR_INVALID, "srx_ctx_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return srx_fail(nullptr, SRX_ERR_DEVICE, "srx_ctx_create: no HIP device (%s); libsrx has no CPU fallback",
                        hipGetErrorString(e));
    if (device < 0 || device >= n || device >= SRX_MAX_DEVICES) return srx_fail(nullptr, SRX_ERR_INVALID, "srx_ctx_create: device %d of %d", device, n);
    SRX_HIP(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    SRX_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return srx_fail(nullptr, SRX_ERR_UNSUPPORTED, "srx_ctx_create: device is %s; this library is built for gfx950 only",
                        prop.gcnArchName);
    srx_ctx *c = new srx_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    if (stream) {
        c->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
        hipError_t es = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (es != hipSuccess) {
            delete c;
            return srx_fail(nullptr, SRX_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(es));
        }
        c->own_stream = true;
    }
    srx_viterbi_noalloc_setup(device);   // per-device address of the fixed back-pointer scratch (srx_viterbi.hip)
    *out = c;
    return SRX_OK;
}

extern "C" void srx_ctx_destroy(srx_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    for (int i = 0; i < 5; i++)
        if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
    for (auto &r : ctx->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto &t : ctx->tables) (void)hipFree(t.dev);
    delete ctx;
}

extern "C" int srx_profile_enable(srx_ctx *ctx, int on) {
    SRX_ENTER(ctx, "srx_profile_enable");
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &r : ctx->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    ctx->prof.clear();
    ctx->prof_on = on != 0;
    return SRX_OK;
}

extern "C" int srx_profile_query(srx_ctx *ctx, const char *kernel, double *total_ms, int64_t *launches) {
    SRX_ENTER(ctx, "srx_profile_query");
    SRX_REQUIRE(ctx, kernel && total_ms && launches, "srx_profile_query: null argument");
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double ms = 0.0;
    int64_t n = 0;
    for (auto &r : ctx->prof)
        if (strcmp(r.name, kernel) == 0) {
            float t = 0.0f;
            SRX_HIP(ctx, hipEventElapsedTime(&t, r.a, r.b));
            ms += t;
            n++;
        }
    *total_ms = ms;
    *launches = n;
    return SRX_OK;
}

int srx_workspace(srx_ctx *ctx, int slot, size_t bytes, void **out) {
    if (bytes > ctx->ws_bytes[slot]) {
        SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->ws[slot]) (void)hipFree(ctx->ws[slot]);
        ctx->ws[slot] = nullptr;
        ctx->ws_bytes[slot] = 0;
        size_t want = bytes + bytes / 4 + 4096;
        SRX_HIP(ctx, hipMalloc(&ctx->ws[slot], want));
        ctx->ws_bytes[slot] = want;
    }
    *out = ctx->ws[slot];
    return SRX_OK;
}

extern "C" const char *srx_last_error(const srx_ctx *ctx) { return ctx ? ctx->err : g_srx_create_err; }

extern "C" void *srx_ctx_stream(const srx_ctx *ctx) { return ctx ? reinterpret_cast<void *>(ctx->stream) : nullptr; }

extern "C" int srx_sync(srx_ctx *ctx) {
    SRX_ENTER(ctx, "srx_sync");
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}

extern "C" int srx_malloc(srx_ctx *ctx, size_t bytes, void **dev) {
    SRX_REQUIRE(nullptr, ctx && dev, "srx_malloc: null argument");
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    SRX_HIP(ctx, hipMalloc(dev, bytes ? bytes : 16));
    return SRX_OK;
}
extern "C" int srx_free(srx_ctx *ctx, void *dev) {
    SRX_ENTER(ctx, "srx_free");
    if (!dev) return SRX_OK;
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SRX_HIP(ctx, hipFree(dev));
    return SRX_OK;
}
extern "C" int srx_memcpy_h2d(srx_ctx *ctx, void *dev, const void *host, size_t bytes) {
    SRX_ENTER(ctx, "srx_memcpy_h2d");
    if (!bytes) return SRX_OK;
    SRX_HIP(ctx, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}
extern "C" int srx_memcpy_d2h(srx_ctx *ctx, void *host, const void *dev, size_t bytes) {
    SRX_ENTER(ctx, "srx_memcpy_d2h");
    if (!bytes) return SRX_OK;
    SRX_HIP(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}
extern "C" int srx_memset(srx_ctx *ctx, void *dev, int value, size_t bytes) {
    SRX_ENTER(ctx, "srx_memset");
    if (!bytes) return SRX_OK;
    SRX_HIP(ctx, hipMemsetAsync(dev, value, bytes, ctx->stream));
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------
// Mel filterbank constants (host).  Follows mel_filterbank (src/mfcc.cpp:190-201), the filter
// constructor (src/mfcc.cpp:117-128) and OpenFrameworks' ofMap (published formula), all in
// f32 with (int) truncation, so the 3*n constants equal the reference's.
// ------------------------------------------------------------------------------------------
static float srx_of_map(float v, float i0, float i1, float o0, float o1) {
    if (fabsf(i0 - i1) < 1.1920929e-07f) return o0;
    return ((v - i0) / (i1 - i0) * (o1 - o0) + o0);
}

float srx_filter_get(const srx_filter &f, int bin) {
    float diff = f.center - (float)bin;
    if (diff < 0) diff = -diff;
    if (diff > f.width) return 0;
    diff /= f.width;
    diff = 1 - diff;
    diff *= f.height;
    return diff;
}

// filter::filter(c, w, dft_bins, sps)  src/mfcc.cpp:117-128 and filter::get  src/mfcc.cpp:132-140
extern "C" int srx_filter_init(float center_hz, float width_hz, int dft_bins, int sps, srx_filter *out) {
    if (!out || dft_bins <= 0 || sps <= 0) return srx_fail(nullptr, SRX_ERR_INVALID, "srx_filter_init: bad arguments");
    srx_filter f;
    f.center = srx_of_map(center_hz, 0, (float)sps, 0, (float)dft_bins);
    f.width = srx_of_map(width_hz, 0, (float)sps, 0, (float)dft_bins);
    f.height = 1;
    float h = 0;
    for (int b = (int)(f.center - f.width); b <= (int)(f.center + f.width); b++) h += srx_filter_get(f, b);
    f.height = 1 / h;
    *out = f;
    return SRX_OK;
}

extern "C" float srx_filter_value(const srx_filter *f, int bin) { return f ? srx_filter_get(*f, bin) : 0.0f; }

extern "C" int srx_mel_filterbank(float end, int n, int dft_bins, int sps, srx_filter *out, int *bin_max) {
    if (!out || n <= 0 || dft_bins <= 0 || sps <= 0 || !(end > 0))
        return srx_fail(nullptr, SRX_ERR_INVALID, "srx_mel_filterbank: bad arguments");
    float mel_end = 1127 * logf(1 + end / 700.0f);
    float mel_step = mel_end / (float)(n + 1);
    float prev = 0;
    for (int i = 0; i < n; i++) {
        float hz = 700 * (expf(((float)(i + 1) * mel_step) / 1127.0f) - 1);
        srx_filter f;
        f.center = srx_of_map(hz, 0, (float)sps, 0, (float)dft_bins);
        f.width = srx_of_map(hz - prev, 0, (float)sps, 0, (float)dft_bins);
        f.height = 1;
        float h = 0;
        for (int b = (int)(f.center - f.width); b <= (int)(f.center + f.width); b++) h += srx_filter_get(f, b);
        f.height = 1 / h;
        out[i] = f;
        prev = hz;
    }
    if (bin_max) *bin_max = (int)(out[n - 1].center + out[n - 1].width) + 1;
    return SRX_OK;
}

extern "C" int srx_mfcc_frame_offsets(const srx_mfcc_cfg *cfg, const int64_t *sample_off, int n_utt, int64_t *frame_off) {
    if (!cfg || !sample_off || !frame_off || n_utt < 0 || cfg->window_len <= 0 || cfg->frame_shift <= 0)
        return srx_fail(nullptr, SRX_ERR_INVALID, "srx_mfcc_frame_offsets: bad arguments");
    frame_off[0] = 0;
    for (int u = 0; u < n_utt; u++) {
        int64_t L = sample_off[u + 1] - sample_off[u];
        if (L <= cfg->window_len)
            return srx_fail(nullptr, SRX_ERR_INVALID,
                            "srx_mfcc_frame_offsets: utterance %d has %lld samples <= window %d (the reference reads past "
                            "its buffer here)",
                            u, (long long)L, cfg->window_len);
        int64_t n = 1 + (L - cfg->window_len - 1) / cfg->frame_shift;
        frame_off[u + 1] = frame_off[u] + n;
    }
    return SRX_OK;
}

// hmm::hmm(vector<phone>, acoustic_model*)  src/hmm.cpp:56-88, USE_SUBPHONES false.
extern "C" int srx_hmm_build_training(const int32_t *phones, int S, int32_t *ntr, float *trlogp, int32_t *ctx) {
    if (!phones || !ntr || !trlogp || S <= 0) return srx_fail(nullptr, SRX_ERR_INVALID, "srx_hmm_build_training: bad arguments");
    const float LOG_HALF = -0.6931471805f, LOG_THIRD = -1.09861228866f;  // src/hmm.hpp:25-26
    for (int i = 0; i < S; i++) {
        ntr[i] = 0;
        trlogp[3 * i] = trlogp[3 * i + 1] = trlogp[3 * i + 2] = 0.0f;
    }
    for (int i = 0; i + 1 < S; i++) {
        if (phones[i + 1] == 0 && i + 2 < S) {
            ntr[i] = 3;
            trlogp[3 * i] = trlogp[3 * i + 1] = trlogp[3 * i + 2] = LOG_THIRD;
        } else {
            ntr[i] = 2;
            trlogp[3 * i] = trlogp[3 * i + 1] = LOG_HALF;
        }
    }
    ntr[S - 1] = 1;
    if (ctx)
        for (int i = 0; i < S; i++) {
            ctx[3 * i] = i == 0 ? 0 : phones[i - 1];
            ctx[3 * i + 1] = phones[i];
            ctx[3 * i + 2] = i + 1 == S ? 0 : phones[i + 1];
        }
    return SRX_OK;
}
