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
// Layout: a workgroup = NW waves = 32 NW consecutive frames of the ragged batch; a wave keeps the fragments of its 32
// frames in registers (KS k-steps x hi/lo) and walks the 32-bin tiles; the table fragments of a bin tile (twiddles
// cos/sin x hi/lo per k-step, then the filterbank columns of those bins) arrive by LDS-DMA, double buffered, as one
// chunk or (SPLIT) as two.
#include <cmath>
#include <type_traits>

#include "srx_mfcc_plan.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 128-bit load from a 4-byte aligned address

#ifndef SRX_MFCC_NW
#define SRX_MFCC_NW 4
#endif
#ifndef SRX_MFCC_SPLIT
#define SRX_MFCC_SPLIT 1
#endif
static constexpr int NW = SRX_MFCC_NW;       // waves per workgroup
// SPLIT: the table of a bin tile arrives as two chunks (first half of the k-steps | the rest + the filterbank columns), so a
// chunk buffer is ~32 KiB, the workgroup's LDS 64 KiB and TWO workgroups share a CU: one's sample loads, scaling and
// log / DCT epilogue (a third of a wave's instructions, no matrix work) run under the other's MFMAs.  With one
// 8-wave workgroup per CU every wave was in the same phase at the same time (MFMA-busy 0.35).
static constexpr bool MSPLIT = SRX_MFCC_SPLIT != 0;
static constexpr int FRAMES_WG = NW * 32;
static constexpr int NRT = 2;                // filter row tiles (filters padded to 64)

// accumulator register r of lane half hh holds tile row (r % 4) + 8 * (r / 4) + 4 * hh
static inline int acc_row(int r, int hh) { return (r % 4) + 8 * (r / 4) + 4 * hh; }

int srx_mfcc16_supported(int half, int n_filters, int n_ceps, int bin_max) {
    const int ks = (half + 15) / 16;
    return (ks == 8 || ks == 10 || ks == 13 || ks == 16) && n_filters <= 32 * NRT && n_ceps <= 16 && bin_max <= 256;
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

#ifdef SRX_MFCC_EXP_STAMP
// diagnostic build: cycles (s_memtime) of wave 0 of every workgroup, summed per phase; printed by the launcher
static __device__ unsigned long long g_mstamp[8];
#define SRX_MSTAMP(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_mstamp[k], now_ - t_prev_); t_prev_ = now_; } } while (0)
#else
#define SRX_MSTAMP(k) do { } while (0)
#endif

template <int KS>
__global__ __launch_bounds__(64 * NW, 2) void k_mfcc_f16(Mfcc16Args a) {
#ifdef SRX_MFCC_EXP_STAMP
    unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) atomicAdd(&g_mstamp[7], 1ull);
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_m[];
    constexpr int CHUNK_FRAGS = KS * 4 + NRT * 4;          // fragments of one bin tile's table
    constexpr int S0 = MSPLIT ? (KS + 1) / 2 : KS;         // k-steps in the tile's first chunk
    constexpr int FR0 = S0 * 4, FR1 = CHUNK_FRAGS - FR0;   // fragments of the first / second chunk
    constexpr int BUF_FRAGS = MSPLIT ? (FR0 > FR1 ? FR0 : FR1) : CHUNK_FRAGS;
    constexpr int BUF_BYTES = BUF_FRAGS * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hh = lane >> 5;
    typedef __attribute__((address_space(3))) unsigned char *lds_ptr;
    const lds_ptr lds3 = (lds_ptr)lds_m;
    const int nchunks = MSPLIT ? 2 * a.nbt : a.nbt;

    auto dma = [&](int chunk, int buf) __attribute__((always_inline)) {
        const int t = MSPLIT ? chunk >> 1 : chunk, part = MSPLIT ? chunk & 1 : 0;
        const int first = part ? FR0 : 0, cnt = MSPLIT ? (part ? FR1 : FR0) : CHUNK_FRAGS;
        const char *g = reinterpret_cast<const char *>(a.pk16) + ((size_t)t * CHUNK_FRAGS + first) * 1024 + lane * 16;
        unsigned char *dst = lds_m + buf * BUF_BYTES;
        for (int p = w; p < cnt; p += NW)
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
        // the loads are scalar ones), and a lane walks forward from there (its frame is at most 31 further on).  (A table
        // of every frame's utterance filled by a small kernel beforehand was measured: no gain.)
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
        // A lane needs the nine consecutive samples x[base-1 .. base+7] of each k-step (pre-emphasis looks one sample
        // back).  It loads the eight from base on as two 128-bit loads (4-byte alignment is enough for
        // global_load_dwordx4); the ninth, x[base-1], is the LAST sample its partner lane (the other half of the frame's
        // k-step, lane ^ 32) loaded -- of the same k-step for the upper half, of the previous one for the lower half -- and
        // comes over by one cross-lane move instead of a third load per k-step: the prologue is bound by the number of
        // cache-line requests of these 64-lines-per-instruction loads (39 -> 27 per lane).  Only the lower half's first
        // k-step has no partner sample: one scalar-width load of x[p0-1] (an utterance's first frame takes x[0], below).
        f4u A[KS], B[KS];
        float C0;
#pragma unroll
        for (int s = 0; s < KS; s++) {
#ifdef SRX_MFCC_EXP_COAL
            // diagnostic build (results are wrong): the same number of loads and bytes, but each instruction reads 1 KiB of
            // consecutive samples instead of 64 pieces 640 bytes apart -- what the per-lane line requests cost
            const float *src = a.x + ((int64_t)w * 2 * KS + 2 * s) * 256 + lane * 4;
            A[s] = *reinterpret_cast<const f4u *>(src);
            B[s] = *reinterpret_cast<const f4u *>(src + 256);
#else
            const float *src = xu + p0 + 16 * s + 8 * hh;
            A[s] = *reinterpret_cast<const f4u *>(src);
            B[s] = *reinterpret_cast<const f4u *>(src + 4);
#endif
        }
        C0 = xu[p0 > 0 ? p0 - 1 : 0];
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int n0 = 16 * s + 8 * hh;
            const bool at0 = s == 0 && p0 + n0 == 0;     // the first sample of an utterance: no x[-1]
            // partner's last sample: of this k-step (I am the upper half) or of the previous one (I am the lower half)
            const float plast_same = __shfl_xor(B[s][3], 32, 64);
            const float plast_prev = s > 0 ? __shfl_xor(B[s > 0 ? s - 1 : 0][3], 32, 64) : C0;
            const float before = hh ? plast_same : plast_prev;
            float W[9] = {before, A[s][0], A[s][1], A[s][2], A[s][3], B[s][0], B[s][1], B[s][2], B[s][3]};
            float L[9];
#pragma unroll
            for (int i = 0; i < 9; i++) L[i] = W[i] * sc;   // normalizeVolume first (src/main.cpp:82; sc = 1 without a gain), then pre-emphasis
            if (a.preemph) {
                // v(p) = x[q] - alpha * x[q-1], q = max(p, 1)  (src/clip.cpp:199-205: x[0] takes the new x[1])
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float m = a.alpha * L[e];
                    y[s][e] = L[e + 1] - m;
                }
                if (s == 0 && at0) {   // x[0] takes the NEW x[1] (src/clip.cpp:204): x[1] - alpha * x[0]
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

    SRX_MSTAMP(0);   // utterance lookup, sample loads, scaling, split into fragments
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 fb[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) fb[rt] = z;

    auto frag = [&](unsigned off) __attribute__((always_inline)) {
        return *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>(lds3 + off);
    };

    f32x16 re = z, im = z;
    // k-steps [SA, SB) of the tile whose chunk sits at LDS offset cb (its first fragment is k-step SA's)
    auto dft_steps = [&](auto sa, auto sb, unsigned cb) __attribute__((always_inline)) {
        constexpr int SA = decltype(sa)::value, SB = decltype(sb)::value;
#pragma unroll
        for (int s = SA; s < SB; s++) {
            const int o = (s - SA) * 4;
            const f16x8 ch = frag(cb + (o + 0) * 1024), cl = frag(cb + (o + 1) * 1024);
            const f16x8 sh = frag(cb + (o + 2) * 1024), sl = frag(cb + (o + 3) * 1024);
            re = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, Yh[s], re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl, Yh[s], im, 0, 0, 0);
            re = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, Yl[s], re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh, Yl[s], im, 0, 0, 0);
            re = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, Yh[s], re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh, Yh[s], im, 0, 0, 0);
        }
    };
    // magnitudes of the finished tile -> GEMM2 with the filterbank columns at LDS offset fbo
    auto finish_tile = [&](unsigned fbo) __attribute__((always_inline)) {
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
                const f16x8 fh = frag(fbo + ((rt * 2 + q) * 2 + 0) * 1024);
                const f16x8 fl = frag(fbo + ((rt * 2 + q) * 2 + 1) * 1024);
                fb[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, Mh[q], fb[rt], 0, 0, 0);
                fb[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, Ml[q], fb[rt], 0, 0, 0);
                fb[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, Mh[q], fb[rt], 0, 0, 0);
            }
        re = z;
        im = z;
    };
    (void)nchunks;
    if (!MSPLIT) {
        for (int t = 0; t < a.nbt; t++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk t have landed
            __syncthreads();                                   // everybody's; and everybody is done with the other buffer
            if (t + 1 < a.nbt) dma(t + 1, (t + 1) & 1);
            const unsigned cb = (t & 1) * BUF_BYTES + lane * 16;
            dft_steps(std::integral_constant<int, 0>{}, std::integral_constant<int, KS>{}, cb);
            finish_tile(cb + KS * 4 * 1024);
        }
    } else {
        // chunk 2t (first k-steps) sits in buffer 0, chunk 2t+1 (the rest + the filterbank columns) in buffer 1
        for (int t = 0; t < a.nbt; t++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            dma(2 * t + 1, 1);
            dft_steps(std::integral_constant<int, 0>{}, std::integral_constant<int, S0>{}, lane * 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 1 < a.nbt) dma(2 * t + 2, 0);
            dft_steps(std::integral_constant<int, S0>{}, std::integral_constant<int, KS>{}, BUF_BYTES + lane * 16);
            finish_tile(BUF_BYTES + lane * 16 + (KS - S0) * 4 * 1024);
        }
    }

    SRX_MSTAMP(1);   // the bin-tile loop
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
    SRX_MSTAMP(2);   // log, DCT, store
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
    const int chunk_frags = KS * 4 + NRT * 4, fr0 = ((KS + 1) / 2) * 4;
    const size_t lds_bytes = (size_t)2 * (MSPLIT ? (fr0 > chunk_frags - fr0 ? fr0 : chunk_frags - fr0) : chunk_frags) * 1024;
    const int64_t nblk = ceil_div64(total_frames, FRAMES_WG);
    SRX_REQUIRE(ctx, nblk < (1ll << 31), "srx_mfcc_batch_dev: too many frames");
    static bool attr_set[SRX_MAX_DEVICES] = {};   // the attribute is per device
    if (!attr_set[ctx->device]) {
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_mfcc_f16<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_mfcc_f16<13>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_mfcc_f16<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_mfcc_f16<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[ctx->device] = true;
    }
    {
        SrxProf prof_(ctx, "k_mfcc_f16");
        if (KS == 10)
            hipLaunchKernelGGL(k_mfcc_f16<10>, dim3((unsigned)nblk), dim3(64 * NW), lds_bytes, ctx->stream, a);
        else if (KS == 13)
            hipLaunchKernelGGL(k_mfcc_f16<13>, dim3((unsigned)nblk), dim3(64 * NW), lds_bytes, ctx->stream, a);
        else if (KS == 8)
            hipLaunchKernelGGL(k_mfcc_f16<8>, dim3((unsigned)nblk), dim3(64 * NW), lds_bytes, ctx->stream, a);
        else if (KS == 16)
            hipLaunchKernelGGL(k_mfcc_f16<16>, dim3((unsigned)nblk), dim3(64 * NW), lds_bytes, ctx->stream, a);
        else
            return srx_fail(ctx, SRX_ERR_UNSUPPORTED, "srx_mfcc_batch_dev: split-f16 path has no kernel for %d k-steps", KS);
    }
    SRX_LAUNCH_CHECK(ctx);
#ifdef SRX_MFCC_EXP_STAMP
    {
        unsigned long long h[8] = {0};
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mstamp), sizeof h);
        const double n = h[7] ? (double)h[7] : 1.0;
        fprintf(stderr, "[mfcc stamps] WGs %llu  prologue %.0f  tile-loop %.0f  epilogue %.0f  cycles per WG (wave 0)\n", h[7], h[0] / n, h[1] / n, h[2] / n);
        unsigned long long z8[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mstamp), z8, sizeof z8);
    }
#endif
    return SRX_OK;
}
