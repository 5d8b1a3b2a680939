// Shared by srx_gmm_f16.hip (dense scoring) and srx_gmm_f16_align.hip (scoring of each utterance's own states).
#pragma once
#include <cmath>
#include <utility>

#include "srx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifndef SRX_NW16
#define SRX_NW16 8
#endif
#ifndef SRX_FT_PER_WAVE
#define SRX_FT_PER_WAVE 2
#endif
#ifndef SRX_F16_WAVES_PER_SIMD
#define SRX_F16_WAVES_PER_SIMD 2
#endif
static constexpr int NW16 = SRX_NW16;                  // waves per workgroup
static constexpr int FT_PER_WAVE = SRX_FT_PER_WAVE;    // 32-frame tiles a wave keeps in registers
static constexpr int FRAMES_PER_WG = NW16 * FT_PER_WAVE * 32;
static constexpr int NT16 = 8;          // padding row tiles after the last slice (>= one LDS chunk + one unrolled step)
static constexpr int NSLICE = 8;        // row slices = XCDs
static constexpr float kLn2f = 0.69314718055994530942f;
static constexpr float kPadConst = -30000.0f;  // constant of a padded component: 2^-30000 = 0, representable in f16
static constexpr float kF16Max = 60000.0f;
static constexpr float kFrameLimit = 250.0f;   // |x'/s| beyond this is re-evaluated exactly (250^2 < 65504)

int padded_components16(int M);


int ensure_packed16(srx_ctx *ctx, srx_model *m, int mix_mode, int KS);   // srx_gmm_f16.hip: colstats, dense table, f16_ok

// Coefficient k (0 .. 16*KS) of Gaussian (g, c):  k = 2d -> quadratic, 2d+1 -> linear, 2D -> constant, else 0.
__device__ __forceinline__ float coef16(const double *__restrict__ mean, const double *__restrict__ var, const float *__restrict__ logwt,
                                        const float *__restrict__ cs, int G, int M, int D, int Muse, int g, int c, int k, double scale) {
    if (g >= G || c >= Muse) return k == 2 * D ? kPadConst : 0.0f;
    const size_t o = (size_t)g * M + c;
    if (k < 2 * D) {
        const int d = k >> 1;
        const double iv = 1.0 / var[o * D + d];
        const double s = (double)cs[D + d];
        return (k & 1) ? (float)(scale * iv * (mean[o * D + d] - (double)cs[d]) * s) : (float)(scale * -0.5 * iv * s * s);
    }
    if (k == 2 * D) {
        const double l2pi = (double)(float)-0.4515827052894549;  // (float)log(M_2_PI), src/gmm.cpp:43
        double acc = 0.0;
        for (int d = 0; d < D; d++) {
            const double mu = mean[o * D + d] - (double)cs[d];
            const double s2 = var[o * D + d];
            acc += l2pi + log(s2) + mu * mu / s2;
        }
        return (float)(scale * ((double)logwt[o] - 0.5 * acc));
    }
    return 0.0f;
}


// Eight halves of a frame-side (B') fragment: k = k0 .. k0+7 of frame t; even k = quadratic term of dimension k/2, odd k =
// linear term, k = 2D the constant 1.  `lo`: the low parts.  Sets `bad` when a scaled feature is beyond kFrameLimit or NaN.
__device__ __forceinline__ f16x8 prep16_frag(const float *__restrict__ feats, int ld, int64_t t, bool valid, int D, int k0, bool lo,
                                             const float *__restrict__ cs, bool &bad) {
    f16x8 out;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const int k = k0 + e;  // even: (quadratic, linear) pair of dimension k/2, or (1, 0) at k = 2D
        float q = 0.0f, l = 0.0f;
        if (valid) {
            if (k < 2 * D) {
                const int d = k >> 1;
                l = (feats[t * ld + d] - cs[d]) * cs[2 * D + d];
                if (!(fabsf(l) <= kFrameLimit)) {  // also catches NaN
                    bad = true;
                    l = 0.0f;
                }
                q = l * l;
            } else if (k == 2 * D) {
                q = 1.0f;
            }
        }
        const _Float16 qh = (_Float16)q, lh = (_Float16)l;
        out[e] = lo ? (_Float16)(q - (float)qh) : qh;
        out[e + 1] = lo ? (_Float16)(l - (float)lh) : lh;
    }
    return out;
}

// The high and the low fragment of the same eight k at once, same arithmetic as prep16_frag.  Every load is
// unconditional (indices clamped, validity applied to the values), so a caller that builds many fragments gets all
// their loads in flight together instead of one dependent load-wait-compute chain per fragment.
__device__ __forceinline__ void prep16_frag2(const float *__restrict__ feats, int ld, int64_t t, bool valid, int D, int k0,
                                             const float *__restrict__ cs, bool &bad, f16x8 &hi, f16x8 &lo) {
    float x[4], c0[4], c1[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int d = (k0 >> 1) + e;
        const int dc = d < D ? d : D - 1;
        x[e] = feats[t * ld + dc];
        c0[e] = cs[dc];
        c1[e] = cs[2 * D + dc];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int k = k0 + 2 * e;
        const float lv = (x[e] - c0[e]) * c1[e];
        const bool isdim = valid && k < 2 * D;
        const bool over = isdim && !(fabsf(lv) <= kFrameLimit);   // also catches NaN
        bad = bad || over;
        const float l = (isdim && !over) ? lv : 0.0f;
        const float q = (valid && k == 2 * D) ? 1.0f : l * l;
        const _Float16 qh = (_Float16)q, lh = (_Float16)l;
        hi[2 * e] = qh;
        hi[2 * e + 1] = lh;
        lo[2 * e] = (_Float16)(q - (float)qh);
        lo[2 * e + 1] = (_Float16)(l - (float)lh);
    }
}

// Per-(MP) tiling constants shared by the kernel and its launcher.
template <int MP>
struct F16Cfg {
    static constexpr int NST = 16 / MP;                       // states per lane half per row tile
    static constexpr int SPT = 32 / MP;                       // states per row tile
    static constexpr int GT = SPT >= 8 ? 1 : 8 / SPT;         // row tiles per output group (a group = a multiple of 8 states)
    static constexpr int RG = SPT * GT;                       // states (staging rows) per group
    static constexpr int U = GT > 2 ? GT : 2;                 // row tiles per unrolled step: whole output groups, whole accumulator pairs
    static constexpr int NT = 4;                              // row tiles per LDS chunk (a multiple of U)
    static constexpr int STAGE_FLOATS = FT_PER_WAVE * RG * 32;  // per wave
};

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>), in order
template <int... Is, class F>
__device__ __forceinline__ void static_for16(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}


// The log-sum-exp epilogue of one row tile, cut into NMF slices so that slice i can be issued between MFMA i and MFMA i+1
// of the next tile (shared by k_gmm_f16 and k_gmm_f16_align).  accp = the tile's accumulators: register r of a lane is
// component r % MP of the lane's state r / MP (log2 domain).  Slices [0, N1): per-state maxima; [N1, N1+N2): differences,
// exp2, sums; the last N3: log2, scale to natural log, and store(f, st, value) for frame tile f, lane-local state st.
// pin(): an empty asm the value passes through.  The arithmetic of a slice is pure, so instruction selection would
// otherwise place it wherever it likes (it clustered all 32 v_sub/v_exp between two MFMAs); with its inputs pinned at
// the start of the slice and its results at the end, each slice stays between its two MFMAs.
template <int MP, int KS>
struct LseEpilogue16 {
    static constexpr int NST = 16 / MP;
    static constexpr int NMF = 3 * KS * FT_PER_WAVE;          // MFMAs per row tile = slices
    static constexpr int NEL = 16 * FT_PER_WAVE;              // accumulator registers (elements) per row tile
    static constexpr int N1 = NMF >= 12 ? 2 : 1;
    static constexpr int N3 = NMF >= 12 ? 2 : 1;
    static constexpr int N2 = NMF - N1 - N3;
    float mx[FT_PER_WAVE][NST], sum[FT_PER_WAVE][NST];

    static __device__ __forceinline__ void pin(float &x) { asm volatile("" : "+v"(x)); }

    template <int I, class Store>
    __device__ __forceinline__ void slice(const f32x16 (&accp)[FT_PER_WAVE], Store &&store) {
        if constexpr (I < N1) {
            if constexpr (MP > 1) {
                constexpr int f0 = I * FT_PER_WAVE / N1, f1 = (I + 1) * FT_PER_WAVE / N1;
#pragma unroll
                for (int f = f0; f < f1; f++) {
#pragma unroll
                    for (int st = 0; st < NST; st++) {
                        // v_max3 directly: fmaxf() on MFMA results gets a canonicalising v_max in front of every operand
                        float m_ = accp[f][st * MP];
                        pin(m_);
                        if constexpr (MP == 2) {
                            asm("v_max_f32 %0, %1, %2" : "=v"(m_) : "v"(m_), "v"(accp[f][st * MP + 1]));
                        } else {
#pragma unroll
                            for (int q = 1; q + 1 < MP; q += 2)
                                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m_) : "v"(m_), "v"(accp[f][st * MP + q]), "v"(accp[f][st * MP + q + 1]));
                            asm("v_max_f32 %0, %1, %2" : "=v"(m_) : "v"(m_), "v"(accp[f][st * MP + MP - 1]));
                        }
                        pin(m_);
                        mx[f][st] = m_;
                        sum[f][st] = 0.0f;
                    }
                }
            }
        } else if constexpr (I < N1 + N2) {
            if constexpr (MP > 1) {
                constexpr int e0 = (I - N1) * NEL / N2, e1 = (I - N1 + 1) * NEL / N2;
                float d_[e1 - e0 + 1];
                // all differences, then all exponentials, then the sums: no instruction waits on the one before it
#pragma unroll
                for (int e = e0; e < e1; e++) {
                    float a_ = accp[e / 16][e % 16];
                    pin(a_);
                    d_[e - e0] = a_ - mx[e / 16][(e % 16) / MP];
                }
#pragma unroll
                for (int e = e0; e < e1; e++) d_[e - e0] = __builtin_amdgcn_exp2f(d_[e - e0]);
#pragma unroll
                for (int e = e0; e < e1; e++) sum[e / 16][(e % 16) / MP] += d_[e - e0];
#pragma unroll
                for (int e = e0; e < e1; e++)
                    if (e == e1 - 1 || (e + 1) / MP != e / MP) pin(sum[e / 16][(e % 16) / MP]);
            }
        } else {
            constexpr int k = I - N1 - N2;
            constexpr int f0 = k * FT_PER_WAVE / N3, f1 = (k + 1) * FT_PER_WAVE / N3;
#pragma unroll
            for (int f = f0; f < f1; f++) {
#pragma unroll
                for (int st = 0; st < NST; st++) {
                    float v;
                    if constexpr (MP == 1) {
                        v = accp[f][st];
                    } else {
                        float s_ = sum[f][st];
                        pin(s_);
                        v = (mx[f][st] + __builtin_amdgcn_logf(s_)) * kLn2f;
                    }
                    pin(v);
                    store(f, st, v);
                }
            }
        }
    }
};
