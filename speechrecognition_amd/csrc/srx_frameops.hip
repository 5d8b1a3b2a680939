// The per-frame operators of the MFCC chain and of the acoustic model as stand-alone calls, in the reference's own
// operation order.
//
// Inside srx_mfcc_batch* these stages are fused into one matrix-core kernel and their intermediates never leave the chip;
// the reference also exposes each of them as a function of its own, and callers written against its headers use them
// one frame at a time:
//   Clip::partial_dft / Clip::dft      src/clip.cpp:166-194  (window through Clip::operator[] const, src/clip.cpp:123-125)
//   filter::operator()                  src/mfcc.cpp:143-149
//   filterbank::operator()              src/mfcc.cpp:161-167
//   log(T*, int)                        src/utils.hpp:55-60
//   dct                                 src/mfcc.cpp:205-214
//   featurevec::compute_deltas / _delta2s  src/mfcc.cpp:73-92 (one vector)
//   gaussian::operator() / logprob      src/gmm.cpp:41-54
// These kernels evaluate them as written -- f64 product and f64 add with the running sum rounded to f32 after every
// add, sequential f32 accumulation of the filter taps -- with the cosines taken from f64 tables the host builds exactly
// as the reference evaluates them, so the results are the reference's bit for bit (log(x+1) to the last place of the
// device's logf).  They are correctness-first (one thread per output, no matrix cores): the batch entry points are
// where the throughput is.  This file is compiled with -ffp-contract=off.
#include <cmath>

#include "srx_common.h"

namespace {

enum { TABLE_DFT = 1, TABLE_DCT = 2 };

void *find_table(srx_ctx *ctx, int kind, int a, int b) {
    for (auto &t : ctx->tables)
        if (t.kind == kind && t.a == a && t.b == b) return t.dev;
    return nullptr;
}

int keep_table(srx_ctx *ctx, int kind, int a, int b, const std::vector<double> &h, void **out) {
    void *d = nullptr;
    SRX_HIP(ctx, hipMalloc(&d, h.size() * sizeof(double)));
    hipError_t e = hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return srx_fail(ctx, SRX_ERR_DEVICE, "table upload failed: %s", hipGetErrorString(e));
    }
    ctx->tables.push_back({kind, a, b, d});
    *out = d;
    return SRX_OK;
}

// cos / sin of -2*M_PI*k*n/N exactly as src/clip.cpp:182-183 evaluates the angle (left to right, in f64):
// table[(n * len + k) * 2 + {0, 1}], n < N/2, k < len -- bin-major inside a sample so that adjacent threads read adjacent pairs
int dft_table(srx_ctx *ctx, int N, int len, const double **out) {
    void *d = find_table(ctx, TABLE_DFT, N, len);
    if (!d) {
        const int half = N / 2;
        std::vector<double> h((size_t)half * len * 2);
        for (int n = 0; n < half; n++)
            for (int k = 0; k < len; k++) {
                const double ang = -2 * M_PI * k * n / N;
                h[((size_t)n * len + k) * 2] = cos(ang);
                h[((size_t)n * len + k) * 2 + 1] = sin(ang);
            }
        if (h.empty()) h.resize(2);
        int rc = keep_table(ctx, TABLE_DFT, N, len, h, &d);
        if (rc != SRX_OK) return rc;
    }
    *out = static_cast<const double *>(d);
    return SRX_OK;
}

// cos(j*(i-0.0f)*M_PI/len) as src/mfcc.cpp:210 evaluates it (the product j*i in f32, the rest in f64): table[j * len + i]
int dct_table(srx_ctx *ctx, int len, int nparams, const double **out) {
    void *d = find_table(ctx, TABLE_DCT, len, nparams);
    if (!d) {
        std::vector<double> h((size_t)len * nparams);
        for (int j = 0; j < nparams; j++)
            for (int i = 0; i < len; i++) {
                const float ji = (float)j * ((float)i - 0.0f);
                h[(size_t)j * len + i] = cos((double)ji * M_PI / len);
            }
        int rc = keep_table(ctx, TABLE_DCT, len, nparams, h, &d);
        if (rc != SRX_OK) return rc;
    }
    *out = static_cast<const double *>(d);
    return SRX_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------ Clip::partial_dft
// One workgroup per frame.  y[n] = x[n] * window[n] (f32, src/clip.cpp:123-125) is staged in LDS once; thread k then runs
// the reference's inner loop for bin k, the magnitudes of all bins are parked in LDS, and output j receives what the
// reference's sequence of writes over k = 0 .. len-1 leaves there (src/clip.cpp:186-192): out[k] and, when len > N/2, the
// mirrored out[N-k-1] -- the later write wins without `stack`, both are added in k order with it.
__global__ __launch_bounds__(256) void k_partial_dft_exact(const float *__restrict__ frames, int64_t ld, const float *__restrict__ window,
                                                           const double *__restrict__ table, int N, int len, int stack,
                                                           float *__restrict__ out, int64_t ldo) {
    extern __shared__ float sh[];
    const int half = N / 2;
    float *y = sh;          // [half]
    float *res = sh + half; // [len]
    const int64_t f = blockIdx.x;
    const float *x = frames + f * ld;
    for (int n = threadIdx.x; n < half; n += blockDim.x) {
        const float w = window ? window[n] : 1.0f;
        y[n] = x[n] * w;
    }
    __syncthreads();
    const float rootn_inv = (float)(1.0f / sqrt((double)N));   // src/clip.cpp:176
    for (int k = threadIdx.x; k < len; k += blockDim.x) {
        float re = 0.0f, im = 0.0f;
        const double *t = table + (size_t)k * 2;
        for (int n = 0; n < half; n++) {
            const double2 cs = *reinterpret_cast<const double2 *>(t + (size_t)n * len * 2);
            const double yn = (double)y[n];
            const double pr = yn * cs.x;
            const double pi = yn * cs.y;
            re = (float)((double)re + pr);
            im = (float)((double)im + pi);
        }
        const float r2 = re * re;
        const float i2 = im * im;
        res[k] = sqrtf(r2 + i2) * rootn_inv;
    }
    __syncthreads();
    const bool mirror = len > N / 2;
    float *o = out + f * ldo;
    for (int j = threadIdx.x; j < len; j += blockDim.x) {
        const int km = N - 1 - j;                              // the bin whose mirrored write lands on out[j]
        const bool has_m = mirror && km >= 0 && km < len;
        if (!stack) {
            o[j] = (has_m && km > j) ? res[km] : res[j];
        } else {
            float v = o[j];
            if (!has_m) {
                v += res[j];
            } else if (km == j) {   // k == N-k-1: `out[k] += res` and `out[N-k-1] += res` of the same iteration
                v += res[j];
                v += res[j];
            } else {
                const int k0 = km < j ? km : j, k1 = km < j ? j : km;
                v += res[k0];
                v += res[k1];
            }
            o[j] = v;
        }
    }
}

extern "C" int srx_partial_dft_frames_dev(srx_ctx *ctx, const float *frames_dev, int64_t ld, const float *window_dev, int N, int len,
                                          int64_t n_frames, int stack, float *out_dev, int64_t ldo) {
    SRX_ENTER(ctx, "srx_partial_dft_frames_dev");
    SRX_REQUIRE(ctx, n_frames >= 0 && N >= 2 && len >= 0, "srx_partial_dft_frames_dev: bad size");
    if (n_frames == 0 || len == 0) return SRX_OK;
    SRX_REQUIRE(ctx, frames_dev && out_dev, "srx_partial_dft_frames_dev: null argument");
    SRX_REQUIRE(ctx, len <= N, "srx_partial_dft_frames_dev: %d bins of a %d-sample window (the reference's mirrored write would index below 0)", len, N);
    SRX_REQUIRE(ctx, ld >= N / 2 && ldo >= len, "srx_partial_dft_frames_dev: row strides shorter than the rows");
    SRX_REQUIRE(ctx, n_frames < (1ll << 31), "srx_partial_dft_frames_dev: too many frames");
    const size_t lds = ((size_t)N / 2 + len) * sizeof(float);
    SRX_REQUIRE(ctx, lds <= 64 * 1024, "srx_partial_dft_frames_dev: window %d with %d bins needs %zu bytes of LDS (limit 64 KiB)", N, len, lds);
    const double *table = nullptr;
    int rc = dft_table(ctx, N, len, &table);
    if (rc != SRX_OK) return rc;
    {
        SrxProf prof_(ctx, "k_partial_dft_exact");
        hipLaunchKernelGGL(k_partial_dft_exact, dim3((unsigned)n_frames), dim3(256), lds, ctx->stream, frames_dev, ld, window_dev, table, N, len,
                           stack, out_dev, ldo);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_partial_dft_frames(srx_ctx *ctx, const float *frames, const float *window, int N, int len, int64_t n_frames, int stack,
                                      float *out) {
    SRX_ENTER(ctx, "srx_partial_dft_frames");
    SRX_REQUIRE(ctx, n_frames >= 0 && N >= 2 && len >= 0, "srx_partial_dft_frames: bad size");
    if (n_frames == 0 || len == 0) return SRX_OK;
    SRX_REQUIRE(ctx, frames && out, "srx_partial_dft_frames: null argument");
    const int half = N / 2;
    DevBuf dx, dw, dout;
    SRX_HIP(ctx, dx.alloc((size_t)n_frames * half * sizeof(float)));
    SRX_HIP(ctx, dout.alloc((size_t)n_frames * len * sizeof(float)));
    // only the first half of every window enters the transform (src/clip.cpp:181): rows of N samples in, half of each copied
    SRX_HIP(ctx, hipMemcpy2DAsync(dx.p, half * sizeof(float), frames, (size_t)N * sizeof(float), half * sizeof(float), (size_t)n_frames,
                                  hipMemcpyHostToDevice, ctx->stream));
    if (window) {
        SRX_HIP(ctx, dw.alloc(half * sizeof(float)));
        SRX_HIP(ctx, hipMemcpyAsync(dw.p, window, half * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    }
    if (stack) SRX_HIP(ctx, hipMemcpyAsync(dout.p, out, (size_t)n_frames * len * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    int rc = srx_partial_dft_frames_dev(ctx, dx.as<float>(), half, window ? dw.as<float>() : nullptr, N, len, n_frames, stack, dout.as<float>(), len);
    if (rc != SRX_OK) return rc;
    SRX_HIP(ctx, hipMemcpyAsync(out, dout.p, (size_t)n_frames * len * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------ filterbank::operator()
static constexpr int SRX_FB_MAX = 64;
struct FbArgs {
    float center[SRX_FB_MAX], width[SRX_FB_MAX], height[SRX_FB_MAX];
};

// filter::get  src/mfcc.cpp:132-140
__device__ __forceinline__ float filter_get(float center, float width, float height, int bin) {
    float diff = center - (float)bin;
    if (diff < 0) diff = -diff;
    if (diff > width) return 0.0f;
    diff /= width;
    diff = 1 - diff;
    diff *= height;
    return diff;
}

// thread = (row, filter): res += get(i) * spectrum[i] over the filter's own bins, f32, in order (src/mfcc.cpp:143-149)
__global__ __launch_bounds__(256) void k_filterbank_exact(const float *__restrict__ spec, int64_t ld, int64_t n_rows, int nfilt, FbArgs fb,
                                                          float *__restrict__ out, int64_t ldo) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * nfilt) return;
    const int64_t r = idx / nfilt;
    const int i = (int)(idx - r * nfilt);
    const float c = fb.center[i], w = fb.width[i], h = fb.height[i];
    const float *s = spec + r * ld;
    float res = 0.0f;
    for (int b = (int)(c - w); b <= (int)(c + w); b++) {
        const float tap = filter_get(c, w, h, b) * s[b];
        res += tap;
    }
    out[r * ldo + i] = res;
}

extern "C" int srx_filterbank_apply_dev(srx_ctx *ctx, const srx_filter *fb, int n_filters, const float *spec_dev, int64_t ld, int n_bins,
                                        int64_t n_rows, float *out_dev, int64_t ldo) {
    SRX_ENTER(ctx, "srx_filterbank_apply_dev");
    SRX_REQUIRE(ctx, n_rows >= 0 && n_filters >= 0, "srx_filterbank_apply_dev: negative size");
    if (n_rows == 0 || n_filters == 0) return SRX_OK;
    SRX_REQUIRE(ctx, fb && spec_dev && out_dev, "srx_filterbank_apply_dev: null argument");
    SRX_REQUIRE(ctx, n_filters <= SRX_FB_MAX, "srx_filterbank_apply_dev: at most %d filters", SRX_FB_MAX);
    SRX_REQUIRE(ctx, ld >= n_bins && ldo >= n_filters, "srx_filterbank_apply_dev: row strides shorter than the rows");
    FbArgs a;
    for (int i = 0; i < n_filters; i++) {
        // the reference indexes spectrum[(int)(center-width) .. (int)(center+width)] unchecked: refuse what would leave the row
        SRX_REQUIRE(ctx, (int)(fb[i].center - fb[i].width) >= 0 && (int)(fb[i].center + fb[i].width) < n_bins,
                    "srx_filterbank_apply_dev: filter %d reaches outside [0, %d)", i, n_bins);
        a.center[i] = fb[i].center;
        a.width[i] = fb[i].width;
        a.height[i] = fb[i].height;
    }
    const int64_t total = n_rows * n_filters;
    SRX_REQUIRE(ctx, ceil_div64(total, 256) < (1ll << 31), "srx_filterbank_apply_dev: too many rows");
    {
        SrxProf prof_(ctx, "k_filterbank_exact");
        hipLaunchKernelGGL(k_filterbank_exact, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, ctx->stream, spec_dev, ld, n_rows, n_filters, a,
                           out_dev, ldo);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_filterbank_apply(srx_ctx *ctx, const srx_filter *fb, int n_filters, const float *spec, int n_bins, int64_t n_rows,
                                    float *out) {
    SRX_ENTER(ctx, "srx_filterbank_apply");
    SRX_REQUIRE(ctx, n_rows >= 0 && n_filters >= 0 && n_bins >= 0, "srx_filterbank_apply: negative size");
    if (n_rows == 0 || n_filters == 0) return SRX_OK;
    SRX_REQUIRE(ctx, fb && spec && out, "srx_filterbank_apply: null argument");
    DevBuf ds, dout;
    SRX_HIP(ctx, ds.alloc((size_t)n_rows * n_bins * sizeof(float)));
    SRX_HIP(ctx, dout.alloc((size_t)n_rows * n_filters * sizeof(float)));
    SRX_HIP(ctx, hipMemcpyAsync(ds.p, spec, (size_t)n_rows * n_bins * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    int rc = srx_filterbank_apply_dev(ctx, fb, n_filters, ds.as<float>(), n_bins, n_bins, n_rows, dout.as<float>(), n_filters);
    if (rc != SRX_OK) return rc;
    SRX_HIP(ctx, hipMemcpyAsync(out, dout.p, (size_t)n_rows * n_filters * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------ log(T*, int)
__global__ __launch_bounds__(256) void k_log1p(float *__restrict__ buf, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the f32 sum, then the logarithm: through the f64 routine and rounded once (the device's logf is two units off
    // glibc's for arguments just above 1, which is where normalised filterbank outputs put it)
    if (i < n) {
        const float a = buf[i] + 1;
        buf[i] = (float)log((double)a);
    }
}

extern "C" int srx_log1p_dev(srx_ctx *ctx, float *buf_dev, int64_t n) {
    SRX_ENTER(ctx, "srx_log1p_dev");
    SRX_REQUIRE(ctx, n >= 0, "srx_log1p_dev: negative size");
    if (n == 0) return SRX_OK;
    SRX_REQUIRE(ctx, buf_dev, "srx_log1p_dev: null argument");
    SRX_REQUIRE(ctx, ceil_div64(n, 256) < (1ll << 31), "srx_log1p_dev: too many values");
    {
        SrxProf prof_(ctx, "k_log1p");
        hipLaunchKernelGGL(k_log1p, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, buf_dev, n);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_log1p(srx_ctx *ctx, float *buf, int64_t n) {
    SRX_ENTER(ctx, "srx_log1p");
    SRX_REQUIRE(ctx, n >= 0, "srx_log1p: negative size");
    if (n == 0) return SRX_OK;
    SRX_REQUIRE(ctx, buf, "srx_log1p: null argument");
    DevBuf d;
    SRX_HIP(ctx, d.alloc((size_t)n * sizeof(float)));
    SRX_HIP(ctx, hipMemcpyAsync(d.p, buf, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    int rc = srx_log1p_dev(ctx, d.as<float>(), n);
    if (rc != SRX_OK) return rc;
    SRX_HIP(ctx, hipMemcpyAsync(buf, d.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------ dct
// thread = (row, j): res[j] += filtered[i] * cos(...), the product and the add in f64, res rounded to f32 per add
__global__ __launch_bounds__(256) void k_dct_exact(const float *__restrict__ in, int64_t ld, int64_t n_rows, int len, int nparams,
                                                   const double *__restrict__ table, float *__restrict__ out, int64_t ldo) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * nparams) return;
    const int64_t r = idx / nparams;
    const int j = (int)(idx - r * nparams);
    const float *x = in + r * ld;
    const double *c = table + (size_t)j * len;
    float res = 0.0f;
    for (int i = 0; i < len; i++) {
        const double p = (double)x[i] * c[i];
        res = (float)((double)res + p);
    }
    out[r * ldo + j] = res;
}

extern "C" int srx_dct_dev(srx_ctx *ctx, const float *in_dev, int64_t ld, int len, int nparams, int64_t n_rows, float *out_dev, int64_t ldo) {
    SRX_ENTER(ctx, "srx_dct_dev");
    SRX_REQUIRE(ctx, n_rows >= 0 && len >= 1 && nparams >= 0, "srx_dct_dev: bad size");
    if (n_rows == 0 || nparams == 0) return SRX_OK;
    SRX_REQUIRE(ctx, in_dev && out_dev, "srx_dct_dev: null argument");
    SRX_REQUIRE(ctx, ld >= len && ldo >= nparams, "srx_dct_dev: row strides shorter than the rows");
    SRX_REQUIRE(ctx, (int64_t)len * nparams <= (1 << 24), "srx_dct_dev: %d x %d cosine table too large", len, nparams);
    const double *table = nullptr;
    int rc = dct_table(ctx, len, nparams, &table);
    if (rc != SRX_OK) return rc;
    const int64_t total = n_rows * nparams;
    SRX_REQUIRE(ctx, ceil_div64(total, 256) < (1ll << 31), "srx_dct_dev: too many rows");
    {
        SrxProf prof_(ctx, "k_dct_exact");
        hipLaunchKernelGGL(k_dct_exact, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, ctx->stream, in_dev, ld, n_rows, len, nparams, table,
                           out_dev, ldo);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_dct(srx_ctx *ctx, const float *in, int len, int nparams, int64_t n_rows, float *out) {
    SRX_ENTER(ctx, "srx_dct");
    SRX_REQUIRE(ctx, n_rows >= 0 && len >= 1 && nparams >= 0, "srx_dct: bad size");
    if (n_rows == 0 || nparams == 0) return SRX_OK;
    SRX_REQUIRE(ctx, in && out, "srx_dct: null argument");
    DevBuf di, dout;
    SRX_HIP(ctx, di.alloc((size_t)n_rows * len * sizeof(float)));
    SRX_HIP(ctx, dout.alloc((size_t)n_rows * nparams * sizeof(float)));
    SRX_HIP(ctx, hipMemcpyAsync(di.p, in, (size_t)n_rows * len * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    int rc = srx_dct_dev(ctx, di.as<float>(), len, len, nparams, n_rows, dout.as<float>(), nparams);
    if (rc != SRX_OK) return rc;
    SRX_HIP(ctx, hipMemcpyAsync(out, dout.p, (size_t)n_rows * nparams * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------ featurevec::compute_deltas (one vector)
// out[q] = (hi[q] - lo[q]) / (k - j), f32 (src/mfcc.cpp:79, 90); span = k - j is 0 (a one-frame utterance: 0/0), 1 or 2
__global__ __launch_bounds__(256) void k_delta(const float *__restrict__ hi, const float *__restrict__ lo, int span, float *__restrict__ out,
                                               int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float d = hi[i] - lo[i];
        out[i] = d / (float)span;
    }
}

extern "C" int srx_delta(srx_ctx *ctx, const float *hi, const float *lo, int span, float *out, int64_t n) {
    SRX_ENTER(ctx, "srx_delta");
    SRX_REQUIRE(ctx, n >= 0, "srx_delta: negative size");
    if (n == 0) return SRX_OK;
    SRX_REQUIRE(ctx, hi && lo && out, "srx_delta: null argument");
    SRX_REQUIRE(ctx, ceil_div64(n, 256) < (1ll << 31), "srx_delta: too many values");
    DevBuf d;
    SRX_HIP(ctx, d.alloc((size_t)n * 3 * sizeof(float)));
    float *p = d.as<float>();
    SRX_HIP(ctx, hipMemcpyAsync(p, hi, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(p + n, lo, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    {
        SrxProf prof_(ctx, "k_delta");
        hipLaunchKernelGGL(k_delta, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, p, p + n, span, p + 2 * n, n);
    }
    SRX_LAUNCH_CHECK(ctx);
    SRX_HIP(ctx, hipMemcpyAsync(out, p + 2 * n, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------ gaussian::operator()
// One Gaussian that is not part of a model: thread = frame, srx_logprob_exact (srx_common.h) = src/gmm.cpp:41-49 term by term
__global__ __launch_bounds__(256) void k_gaussian_exact(const float *__restrict__ feats, int64_t ld, int64_t T, int D,
                                                        const double *__restrict__ mean, const double *__restrict__ var, float logwt,
                                                        float *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) out[t] = srx_logprob_exact(feats + t * ld, mean, var, logwt, D);
}

extern "C" int srx_gaussian_logprob(srx_ctx *ctx, const float *feats, int ld, int64_t T, int D, const double *mean, const double *var,
                                    float logwt, float *out) {
    SRX_ENTER(ctx, "srx_gaussian_logprob");
    SRX_REQUIRE(ctx, T >= 0 && D >= 0 && ld >= D, "srx_gaussian_logprob: bad size");
    if (T == 0) return SRX_OK;
    SRX_REQUIRE(ctx, feats && mean && var && out, "srx_gaussian_logprob: null argument");
    SRX_REQUIRE(ctx, ceil_div64(T, 256) < (1ll << 31), "srx_gaussian_logprob: too many frames");
    DevBuf dx, dp, dout;
    SRX_HIP(ctx, dx.alloc((size_t)T * ld * sizeof(float)));
    SRX_HIP(ctx, dp.alloc((size_t)(2 * D + 2) * sizeof(double)));
    SRX_HIP(ctx, dout.alloc((size_t)T * sizeof(float)));
    SRX_HIP(ctx, hipMemcpyAsync(dx.p, feats, (size_t)T * ld * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (D > 0) {
        SRX_HIP(ctx, hipMemcpyAsync(dp.p, mean, (size_t)D * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        SRX_HIP(ctx, hipMemcpyAsync(dp.as<double>() + D, var, (size_t)D * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    {
        SrxProf prof_(ctx, "k_gaussian_exact");
        hipLaunchKernelGGL(k_gaussian_exact, dim3((unsigned)ceil_div64(T, 256)), dim3(256), 0, ctx->stream, dx.as<float>(), (int64_t)ld, T, D,
                           dp.as<double>(), dp.as<double>() + D, logwt, dout.as<float>());
    }
    SRX_LAUNCH_CHECK(ctx);
    SRX_HIP(ctx, hipMemcpyAsync(out, dout.p, (size_t)T * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}
