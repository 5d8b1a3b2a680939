// Shared by srx_gmm_f16.hip (dense scoring) and srx_gmm_f16_align.hip (scoring of each utterance's own states).
#pragma once
#include <cmath>
#include <utility>

#include "srx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

static constexpr int NW16 = 8;          // waves per workgroup
static constexpr int FT_PER_WAVE = 2;   // 32-frame tiles a wave keeps in registers
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

