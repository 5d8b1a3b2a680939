// MFCC extraction on the f16 matrix cores of gfx950 with split operands (a = a_hi + a_lo, each an f16;
// products a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, f32 accumulate: ~22 significant bits per product).
//
// Replaces the same reference code as srx_mfcc.hip (paths relative to /root/reference):
//   Clip::partial_dft  src/clip.cpp:174-194   -> GEMM1  [bins x half] . [half x frames]   (cos and sin rows)
//   filterbank::operator()  src/mfcc.cpp:143-149,161-167 -> GEMM2  [filters x bins] . [bins x frames]
//   log(T*,int)  src/utils.hpp:55-60 -> logf(x + 1)
//   dct  src/mfcc.cpp:205-214 -> GEMM3  [ceps x filters] . [filters x frames]
//   normalizeVolume / preemphasis fused into the frame load as in srx_mfcc.hip; hamming_window folded into the twiddles
//
// Frames are the MFMA columns throughout: a lane owns one frame (lane % 32) and, in every accumulator, 16 of the 32 rows
// of a tile.  The accumulators of one product are therefore, converted to f16 pairs, exactly the B fragments of the next
// product (the k index of an MFMA may be any permutation as long as both operands use it; the table side is packed on
// the host in the matching order), so magnitudes, filterbank outputs and log values never leave the lane.
//
// f16 range: each frame is scaled by its own power of two (exact) so that its largest windowed sample lies in [16, 32);
// magnitudes are then <= 200 * 32 * sqrt(2) < 65504.  The filterbank outputs are unscaled before log(x+1); the log
// values get a second per-frame power of two for the DCT.  The scale of a column is a scale of the output column.
//
// Layout: a workgroup = 8 waves = 256 consecutive frames of the ragged batch; a wave keeps the fragments of its 32
// frames in registers (KS k-steps x hi/lo) and walks the 32-bin tiles; the table fragments of a bin tile (twiddles
// cos/sin x hi/lo per k-step, then the filterbank columns of those bins) arrive as one chunk by LDS-DMA, double buffered.
#include <cmath>

#include "srx_mfcc_plan.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 128-bit load from a 4-byte aligned address

static constexpr int NW = 8;                 // waves per workgroup
static constexpr int FRAMES_WG = NW * 32;
static constexpr int NRT = 2;                // filter row tiles (filters padded to 64)

// accumulator register r of lane half hh holds tile row (r % 4) + 8 * (r / 4) + 4 * hh
static inline int acc_row(int r, int hh) { return (r % 4) + 8 * (r / 4) + 4 * hh; }

int srx_mfcc16_supported(int half, int n_filters, int n_ceps, int bin_max) {
    const int ks = (half + 15) / 16;
    return (ks == 10 || ks == 13) && n_filters <= 32 * NRT && n_ceps <= 16 && bin_max <= 256;
}

static void split16(float v, _Float16 *hi, _Float16 *lo) {
    const _Float16 h = (_Float16)v;
    *hi = h;
    *lo = (_Float16)(v - (float)h);
}

// Packs the tables of the f32 path (same values: f64 trigonometry rounded to f32 once) into MFMA fragments.
int srx_mfcc16_build(srx_mfcc_plan *p, const std::vector<float> &win, const std::vector<float> &tw, const std::vector<float> &fbm,
                     const std::vector<float> &dm) {
    const int KS = (p->half + 15) / 16;
    const int NBT = (p->bin_max + 31) / 32;
    const size_t frag = 64 * 8;  // halves per fragment
    const size_t chunk_frags = (size_t)KS * 4 + NRT * 2 * 2;
    std::vector<_Float16> pk(NBT * chunk_frags * frag);
    for (int t = 0; t < NBT; t++) {
        _Float16 *ck = pk.data() + (size_t)t * chunk_frags * frag;
        for (int s = 0; s < KS; s++)
            for (int part = 0; part < 2; part++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 8; e++) {
                        const int b = 32 * t + lane % 32, n = 16 * s + 8 * (lane / 32) + e;
                        // the window is folded into the table: (x w) cos = x (w cos)
                        const float v = (b < p->bin_max && n < p->half) ? (float)((double)win[n] * (double)tw[((size_t)n * 2 + part) * p->NB + b]) : 0.0f;
                        _Float16 hi, lo;
                        split16(v, &hi, &lo);
                        ck[(((size_t)s * 2 + part) * 2 + 0) * frag + lane * 8 + e] = hi;
                        ck[(((size_t)s * 2 + part) * 2 + 1) * frag + lane * 8 + e] = lo;
                    }
        for (int rt = 0; rt < NRT; rt++)
            for (int q = 0; q < 2; q++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 8; e++) {
                        const int i = 32 * rt + lane % 32;
                        const int b = 32 * t + acc_row(8 * q + e, lane / 32);
                        const float v = (i < p->nfilt && b < p->bin_max) ? fbm[(size_t)b * p->NFP + i] : 0.0f;
                        _Float16 hi, lo;
                        split16(v, &hi, &lo);
                        const size_t fi = (size_t)KS * 4 + ((size_t)rt * 2 + q) * 2;
                        ck[(fi + 0) * frag + lane * 8 + e] = hi;
                        ck[(fi + 1) * frag + lane * 8 + e] = lo;
                    }
    }
    std::vector<_Float16> dk((size_t)NRT * 2 * 2 * frag);
    for (int rt = 0; rt < NRT; rt++)
        for (int q = 0; q < 2; q++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 8; e++) {
                    const int c = lane % 32;
                    const int i = 32 * rt + acc_row(8 * q + e, lane / 32);
                    const float v = (c < p->cfg.n_ceps && i < p->nfilt) ? dm[(size_t)i * 16 + c] : 0.0f;
                    _Float16 hi, lo;
                    split16(v, &hi, &lo);
                    const size_t fi = ((size_t)rt * 2 + q) * 2;
                    dk[(fi + 0) * frag + lane * 8 + e] = hi;
                    dk[(fi + 1) * frag + lane * 8 + e] = lo;
                }
    hipError_t e;
    if ((e = hipMalloc(&p->pk16, pk.size() * 2)) != hipSuccess || (e = hipMalloc(&p->dct16, dk.size() * 2)) != hipSuccess ||
        (e = hipMemcpy(p->pk16, pk.data(), pk.size() * 2, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->dct16, dk.data(), dk.size() * 2, hipMemcpyHostToDevice)) != hipSuccess)
        return srx_fail(p->ctx, SRX_ERR_DEVICE, "srx_mfcc_plan_create: f16 table upload failed: %s", hipGetErrorString(e));
    p->ks16 = KS;
    p->nbt16 = NBT;
    return SRX_OK;
}

struct Mfcc16Args {
    const float *x;
    const int64_t *soff;
    const int64_t *foff;
    const float *scale;  // per-utterance gain, may be null
    const f16x8 *pk16;
    const f16x8 *dct16;
    float *ceps;
    int64_t total_frames;
    int n_utt, hop, half, n_ceps, nbt;
    int preemph;
    float alpha, rootn_inv;
};

__device__ __forceinline__ int find_utt16(const int64_t *foff, int n_utt, int64_t f) {
    int lo = 0, hi = n_utt;  // foff[lo] <= f < foff[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (foff[mid] <= f) lo = mid; else hi = mid;
    }
    return lo;
}

// v -> (hi, lo) halves
__device__ __forceinline__ void split_dev(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// power of two e such that m * 2^e lies in [16, 32)  (m > 0 finite); 0 for m == 0 or not finite
__device__ __forceinline__ int scale_exp(float m) {
    if (!(m > 0.0f) || m > 3.0e38f) return 0;
    int ex;
    (void)frexpf(m, &ex);  // m = fr * 2^ex, fr in [0.5, 1)
    int e = 5 - ex;
    return e > 120 ? 120 : (e < -120 ? -120 : e);
}

template <int KS>
__global__ __launch_bounds__(64 * NW, 2) void k_mfcc_f16(Mfcc16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_m[];
    constexpr int CHUNK_FRAGS = KS * 4 + NRT * 4;
    constexpr int CHUNK_BYTES = CHUNK_FRAGS * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hh = lane >> 5;
    typedef __attribute__((address_space(3))) unsigned char *lds_ptr;
    const lds_ptr lds3 = (lds_ptr)lds_m;

    auto dma = [&](int chunk, int buf) __attribute__((always_inline)) {
        const char *g = reinterpret_cast<const char *>(a.pk16) + (size_t)chunk * CHUNK_BYTES + lane * 16;
        unsigned char *dst = lds_m + buf * CHUNK_BYTES;
        for (int p = w; p < CHUNK_FRAGS; p += NW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)p * 1024),
                                             (__attribute__((address_space(3))) void *)(dst + p * 1024), 16, 0, 0);
    };
    dma(0, 0);

    // ---- this lane's frame: windowed half-frame, scaled, split into B fragments ----
    const int64_t f = (int64_t)blockIdx.x * FRAMES_WG + 32 * w + j;
    const bool valid = f < a.total_frames;
    const float *xu = a.x;
    int64_t p0 = 0;
    float sc = 1.0f;
    if (valid) {
        // the binary search is a chain of dependent loads: it runs once per wave, on the wave's first frame (uniform, so
        // the loads are scalar ones), and a lane walks forward from there (its frame is at most 31 further on)
        const int64_t fw = (int64_t)blockIdx.x * FRAMES_WG + 32 * w;
        int u = find_utt16(a.foff, a.n_utt, fw < a.total_frames ? fw : a.total_frames - 1);
        u = __builtin_amdgcn_readfirstlane(u);
        while (u + 1 < a.n_utt && a.foff[u + 1] <= f) u++;
        xu = a.x + a.soff[u];
        p0 = (f - a.foff[u]) * (int64_t)a.hop;
        if (a.scale) sc = a.scale[u];
    }
    f16x8 Yh[KS], Yl[KS];
    int E1;   // the frame's scale exponent
    {
        float ymax = 0.0f;
        float y[KS][8];
        // All sample loads first, unconditionally: an invalid lane reads frame 0 of the batch (xu = x, p0 = 0) and a
        // k-step past the half window reads on inside the frame's own window (16 KS <= window), so every address is
        // inside the buffer; validity is applied to the values.  Inside `if` bodies the loads of the 13 k-steps became
        // 13 dependent load-wait chains.
        f4u A[KS], B[KS];
        float C[KS];
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int64_t base = p0 + 16 * s + 8 * hh;
            // nine consecutive samples x[base-1 .. base+7] as two 128-bit loads and one 32-bit load (4-byte alignment is
            // enough for global_load_dwordx4); the first frame of an utterance has no x[-1]: it loads x[0 .. 8] and shifts
            const bool at0 = s == 0 && base == 0;
            const float *src = xu + (at0 ? 0 : base - 1);
            A[s] = *reinterpret_cast<const f4u *>(src);
            B[s] = *reinterpret_cast<const f4u *>(src + 4);
            C[s] = src[8];
        }
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int n0 = 16 * s + 8 * hh;
            const bool at0 = s == 0 && p0 + n0 == 0;
            float W[9] = {A[s][0], A[s][1], A[s][2], A[s][3], B[s][0], B[s][1], B[s][2], B[s][3], C[s]};
            float L[9];
#pragma unroll
            for (int i = 0; i < 9; i++) {
                float v = (s == 0 && at0) ? W[i ? i - 1 : 0] : W[i];
                L[i] = v * sc;   // normalizeVolume first (src/main.cpp:82; sc = 1 without a gain), then pre-emphasis
            }
            if (a.preemph) {
                // v(p) = x[q] - alpha * x[q-1], q = max(p, 1)  (src/clip.cpp:199-205: x[0] takes the new x[1])
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float m = a.alpha * L[e];
                    y[s][e] = L[e + 1] - m;
                }
                if (s == 0 && at0) {
                    const float m = a.alpha * L[1];
                    y[s][0] = L[2] - m;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) y[s][e] = L[e + 1];
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (!valid || n0 + e >= a.half) y[s][e] = 0.0f;   // (the window itself is part of the table)
                ymax = fmaxf(ymax, fabsf(y[s][e]));
            }
        }
        ymax = fmaxf(ymax, __shfl_xor(ymax, 32, 64));
        const int E = E1 = scale_exp(ymax);
#pragma unroll
        for (int s = 0; s < KS; s++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                _Float16 hi, lo;
                split_dev(ldexpf(y[s][e], E), hi, lo);
                Yh[s][e] = hi;
                Yl[s][e] = lo;
            }
    }

    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 fb[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) fb[rt] = z;

    auto frag = [&](unsigned off) __attribute__((always_inline)) {
        return *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>(lds3 + off);
    };

    for (int t = 0; t < a.nbt; t++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk t have landed
        __syncthreads();                                   // everybody's; and everybody is done with the other buffer
        if (t + 1 < a.nbt) dma(t + 1, (t + 1) & 1);
        const unsigned cb = (t & 1) * CHUNK_BYTES + lane * 16;
        // GEMM1: re/im of 32 bins for the wave's 32 frames
        f32x16 re = z, im = z;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const f16x8 ch = frag(cb + (s * 4 + 0) * 1024), cl = frag(cb + (s * 4 + 1) * 1024);
            const f16x8 sh = frag(cb + (s * 4 + 2) * 1024), sl = frag(cb + (s * 4 + 3) * 1024);
            re = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, Yh[s], re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl, Yh[s], im, 0, 0, 0);
            re = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, Yl[s], re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh, Yl[s], im, 0, 0, 0);
            re = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, Yh[s], re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh, Yh[s], im, 0, 0, 0);
        }
        // magnitudes (scaled domain) -> B fragments of GEMM2: k-step q takes accumulator registers 8q .. 8q+7
        f16x8 Mh[2], Ml[2];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float s2 = re[r] * re[r];
            const float i2 = im[r] * im[r];
            const float mg = __builtin_amdgcn_sqrtf(s2 + i2);
            _Float16 hi, lo;
            split_dev(mg, hi, lo);
            Mh[r / 8][r % 8] = hi;
            Ml[r / 8][r % 8] = lo;
        }
        // GEMM2: the filterbank columns of these 32 bins
#pragma unroll
        for (int rt = 0; rt < NRT; rt++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const f16x8 fh = frag(cb + (KS * 4 + (rt * 2 + q) * 2 + 0) * 1024);
                const f16x8 fl = frag(cb + (KS * 4 + (rt * 2 + q) * 2 + 1) * 1024);
                fb[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, Mh[q], fb[rt], 0, 0, 0);
                fb[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, Ml[q], fb[rt], 0, 0, 0);
                fb[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, Mh[q], fb[rt], 0, 0, 0);
            }
    }

    // ---- unscale, * 1/sqrt(N) (src/clip.cpp:176), log(x + 1), second per-frame scale ----
    float lf[NRT][16];
    float lmax = 0.0f;
#pragma unroll
    for (int rt = 0; rt < NRT; rt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float v = ldexpf(fb[rt][r] * a.rootn_inv, -E1);
            lf[rt][r] = logf(v + 1.0f);
            lmax = fmaxf(lmax, fabsf(lf[rt][r]));
        }
    lmax = fmaxf(lmax, __shfl_xor(lmax, 32, 64));
    const int E2 = scale_exp(lmax);
    // ---- GEMM3: DCT ----
    f32x16 cp = z;
#pragma unroll
    for (int rt = 0; rt < NRT; rt++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
            f16x8 Lh, Ll;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                _Float16 hi, lo;
                split_dev(ldexpf(lf[rt][8 * q + e], E2), hi, lo);
                Lh[e] = hi;
                Ll[e] = lo;
            }
            const f16x8 dh = a.dct16[((rt * 2 + q) * 2 + 0) * 64 + lane];
            const f16x8 dl = a.dct16[((rt * 2 + q) * 2 + 1) * 64 + lane];
            cp = __builtin_amdgcn_mfma_f32_32x32x16_f16(dl, Lh, cp, 0, 0, 0);
            cp = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, Ll, cp, 0, 0, 0);
            cp = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, Lh, cp, 0, 0, 0);
        }
    if (valid) {
#pragma unroll
        for (int r = 0; r < 8; r++) {   // rows 0..15 of the tile live in registers 0..7
            const int c = (r % 4) + 8 * (r / 4) + 4 * hh;
            if (c < a.n_ceps) a.ceps[f * a.n_ceps + c] = ldexpf(cp[r], -E2);
        }
    }
}

int srx_mfcc16_launch(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples_dev, const int64_t *sample_off_dev, int n_utt,
                      const int64_t *frame_off_dev, int64_t total_frames, const float *scale_dev, float *ceps_dev) {
    const srx_mfcc_cfg &c = plan->cfg;
    Mfcc16Args a;
    a.x = samples_dev; a.soff = sample_off_dev; a.foff = frame_off_dev; a.scale = scale_dev;
    a.pk16 = (const f16x8 *)plan->pk16; a.dct16 = (const f16x8 *)plan->dct16;
    a.ceps = ceps_dev; a.total_frames = total_frames; a.n_utt = n_utt; a.hop = c.frame_shift; a.half = plan->half;
    a.n_ceps = c.n_ceps; a.nbt = plan->nbt16; a.preemph = c.preemphasis; a.alpha = c.alpha; a.rootn_inv = plan->rootn_inv;
    const int KS = plan->ks16;
    const size_t lds_bytes = (size_t)2 * (KS * 4 + NRT * 4) * 1024;
    const int64_t nblk = ceil_div64(total_frames, FRAMES_WG);
    SRX_REQUIRE(ctx, nblk < (1ll << 31), "srx_mfcc_batch_dev: too many frames");
    static bool attr_set[SRX_MAX_DEVICES] = {};   // the attribute is per device
    if (!attr_set[ctx->device]) {
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_mfcc_f16<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_mfcc_f16<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[ctx->device] = true;
    }
    {
        SrxProf prof_(ctx, "k_mfcc_f16");
        if (KS == 10)
            hipLaunchKernelGGL(k_mfcc_f16<10>, dim3((unsigned)nblk), dim3(64 * NW), lds_bytes, ctx->stream, a);
        else if (KS == 13)
            hipLaunchKernelGGL(k_mfcc_f16<13>, dim3((unsigned)nblk), dim3(64 * NW), lds_bytes, ctx->stream, a);
        else
            return srx_fail(ctx, SRX_ERR_UNSUPPORTED, "srx_mfcc_batch_dev: split-f16 path has no kernel for %d k-steps", KS);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}
