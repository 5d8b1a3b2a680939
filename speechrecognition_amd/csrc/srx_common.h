// Internal shared definitions for libsrx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/srx.h"

static constexpr int SRX_MAX_DEVICES = 64;   // srx_ctx_create rejects a device index beyond this

struct srx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    char err[512] = {0};
    // grow-only device workspace (Viterbi back-pointer words, ...)
    // slot 0: GMM operand rows, 1: Viterbi back-pointers, 2: frames sorted by state (EM statistics), 3: EM partial sums,
    // 4: utterance of each frame block (FFT path of the MFCC)
    void *ws[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t ws_bytes[5] = {0, 0, 0, 0, 0};
    // what slot 2 currently holds (srx_train_seg.hip): the alignment it was sorted from
    const int32_t *em_gid = nullptr;
    int64_t em_T = -1;
    int em_G = 0;
    // per-kernel timing (srx_profile_enable / srx_profile_query): event pairs recorded on `stream`
    struct prof_rec {
        const char *name;
        hipEvent_t a, b;
    };
    bool prof_on = false;
    std::vector<prof_rec> prof;
    // small constant tables of the per-frame operators (srx_frameops.hip): f64 twiddles keyed by (window, bins), f64 DCT
    // cosines keyed by (len, nparams).  Built on first use, freed with the context.
    struct table_rec {
        int kind, a, b;
        void *dev;
    };
    std::vector<table_rec> tables;
};

// Brackets one kernel launch with events on the context's stream while profiling is enabled.
struct SrxProf {
    srx_ctx *c;
    hipEvent_t b = nullptr;
    SrxProf(srx_ctx *ctx, const char *name) : c(ctx) {
        if (!c->prof_on) return;
        hipEvent_t a = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { b = nullptr; return; }
        (void)hipEventRecord(a, c->stream);
        c->prof.push_back({name, a, b});
    }
    ~SrxProf() {
        if (b) (void)hipEventRecord(b, c->stream);
    }
};

// Returns a device workspace of at least `bytes` (contents undefined).  Grow-only, owned by ctx.
int srx_workspace(srx_ctx *ctx, int slot, size_t bytes, void **out);

// srx_viterbi.hip: looks up this device's copy of the fixed back-pointer array of srx_viterbi_batch_noalloc_dev
// (the device must be current).  Called by srx_ctx_create.
void srx_viterbi_noalloc_setup(int device);

extern char g_srx_create_err[512];

static inline int srx_fail(srx_ctx *ctx, int code, const char *fmt, ...) {
    char *dst = ctx ? ctx->err : g_srx_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define SRX_HIP(ctx, call)                                                                                   \
    do {                                                                                                     \
        hipError_t e__ = (call);                                                                             \
        if (e__ != hipSuccess)                                                                               \
            return srx_fail((ctx), e__ == hipErrorOutOfMemory ? SRX_ERR_NOMEM : SRX_ERR_DEVICE,              \
                            "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__);     \
    } while (0)

#define SRX_REQUIRE(ctx, cond, ...)                                    \
    do {                                                               \
        if (!(cond)) return srx_fail((ctx), SRX_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// First line of every entry point that takes a context: rejects a null context and makes the context's device the
// calling thread's current one, so that allocations, function attributes and launches land on that GPU whatever the
// caller's current device is (a process may hold contexts on several devices; each context is still used by one thread
// at a time).
#define SRX_ENTER(ctx, name)                                                                                 \
    do {                                                                                                     \
        if (!(ctx)) return srx_fail(nullptr, SRX_ERR_INVALID, "%s: null ctx", name);                         \
        SRX_HIP(ctx, hipSetDevice((ctx)->device));                                                           \
    } while (0)

#define SRX_LAUNCH_CHECK(ctx) SRX_HIP(ctx, hipGetLastError())

// RAII device buffer for the host-pointer convenience entry points.
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T>
    T *as() {
        return reinterpret_cast<T *>(p);
    }
};

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Model storage (device-resident, double precision master copy + packed f32 MFMA operand).
struct srx_model {
    srx_ctx *ctx = nullptr;
    int G = 0, M = 0, D = 0;
    double *mean = nullptr;    // [G][M][D]
    double *var = nullptr;     // [G][M][D]
    float *weight = nullptr;   // [G][M]
    float *logwt = nullptr;    // [G][M]  logf(weight)
    // packed operand for the MFMA scoring kernel (rebuilt lazily when dirty)
    float *packed = nullptr;   // see srx_gmm.hip
    float *center = nullptr;   // [D] centring vector (f32)
    int packed_mode = -1;      // mix mode the packing was built for
    // every change of the parameters bumps `version`; each packed form remembers the version it was built from
    uint64_t version = 1;
    uint64_t packed_version = 0;
    size_t packed_bytes = 0;
    // split-f16 operand of the SRX_SCORE_MFMA_F16X3 path (srx_gmm_f16.hip)
    void *packed16 = nullptr;      // [row tile][k-step][lane][8 halves], hi segments then lo segments
    size_t packed16_bytes = 0;
    uint64_t packed16_version = 0;
    int packed16_mode = -1;
    // frame-independent part of gaussian::operator(): lconst[o][d] = (double)(float)log(2/pi) + log(var[o][d])  (srx_train.hip)
    double *lconst = nullptr;   // [G*M*D] lconst followed by [G*M*D] 1/var
    uint64_t lconst_version = 0;
    void *packed16s = nullptr;     // per-state form for alignment-mode scoring (srx_gmm_f16_align.hip)
    size_t packed16s_bytes = 0;
    uint64_t packed16s_version = 0;
    int packed16s_mode = -1;
    uint64_t packed16s_epoch = 0;  // colstats_epoch the per-state table was built with
    float *colstats = nullptr;     // [3][D]: centre, column scale s_d, 1/s_d in use; followed by [3][D] of the current model (candidate)
    uint64_t colstats_epoch = 0;   // bumped whenever the values in use change (frame operands built with them are then stale)
    int *flag16 = nullptr;         // device flag: some coefficient does not fit f16
    bool f16_ok = false;
};

// gaussian::operator()  src/gmm.cpp:41-49 in the reference's operation order (f64 stands in for long double)
__device__ __forceinline__ float srx_logprob_exact(const float *__restrict__ fv, const double *__restrict__ mean,
                                                   const double *__restrict__ var, float logwt, int D) {
    float res = 0.0f;
    const float l2pi = (float)-0.4515827052894549;  // (float)log(M_2_PI), src/gmm.cpp:43
    for (int i = 0; i < D; i++) {
        float t = (float)((double)fv[i] - mean[i]);
        float tt = t * t;
        double term = ((double)l2pi + log(var[i])) + (double)tt / var[i];
        res = (float)((double)res + term);
    }
    float h = -0.5f * res;
    return h + logwt;
}

// gmm::operator() as the commented-out intent of src/gmm.cpp:103-107: log(sum_m exp(g_m)) in double, for the M components
// of one tied state (mean/var/logwt point at its component 0).  Written as the reference has it wherever that form is
// well inside the f64 range; where every exp() underflows (a frame tens of standard deviations from every mean: the
// reference's form gives log(0) = -inf there, and one such frame removes its whole utterance from training through the
// pruning test of src/hmm.cpp:165) or overflows, the sum is taken relative to the largest component, which is the same
// quantity without the range limit.  DESIGN.md section 2 lists this as a divergence from the dead code.
__device__ __forceinline__ float srx_lse_exact(const float *__restrict__ fv, const double *__restrict__ mean,
                                               const double *__restrict__ var, const float *__restrict__ logwt, int M, int D) {
    double s = 0.0;
    float mxv = -__builtin_huge_valf();
    for (int m = 0; m < M; m++) {
        const float lp = srx_logprob_exact(fv, mean + (size_t)m * D, var + (size_t)m * D, logwt[m], D);
        s += exp((double)lp);
        mxv = fmaxf(mxv, lp);
    }
    if (s > 1e-290 && s < 1e290) return (float)log(s);
    if (s != s) return (float)s;                            // NaN feature or parameter: NaN, as log(NaN) in the reference's form
    if (mxv == -__builtin_huge_valf()) return mxv;          // every component at -inf (all weights zero)
    double s2 = 0.0;
    for (int m = 0; m < M; m++) {
        const float lp = srx_logprob_exact(fv, mean + (size_t)m * D, var + (size_t)m * D, logwt[m], D);
        s2 += exp((double)lp - (double)mxv);
    }
    return (float)(log(s2) + (double)mxv);
}

// The statistics form: the frame-independent sum (l2pi + log(var[i])) and the reciprocal of the variance come from
// per-model tables, so the inner loop has no division and no log.  tt * (1 / var) differs from tt / var by at most one
// ulp of the f64 quotient, 2^-29 of the f32 rounding applied to `res` on the next line (the reference divides in
// 80-bit long double; neither f64 form reproduces that bit for bit).  The multiply and the add of the constant are
// one fused operation (one f64 rounding instead of two, before the f32 rounding of `res`).
__device__ __forceinline__ float srx_logprob_exact_lc(const float *__restrict__ fv, const double *__restrict__ mean,
                                                      const double *__restrict__ rvar, const double *__restrict__ lconst, float logwt,
                                                      int D) {
    float res = 0.0f;
    for (int i = 0; i < D; i++) {
        float t = (float)((double)fv[i] - mean[i]);
        float tt = t * t;
        double term = fma((double)tt, rvar[i], lconst[i]);
        res = (float)((double)res + term);
    }
    float h = -0.5f * res;
    return h + logwt;
}
