// MFCC with the half-window transform as a mixed-radix FFT in LDS (f32 vector arithmetic, no matrix cores).
//
// Replaces the same reference code as srx_mfcc.hip (Clip::partial_dft src/clip.cpp:174-194 with hamming_window through
// Clip::operator[] const, Clip::preemphasis, filterbank::operator(), log(x+1), dct, the frame loop of write_features).
// The reference's transform sums y[n] e^{-2 pi i k n / N} over the first H = N/2 samples only (src/clip.cpp:181): it is the
// N-point DFT of the REAL sequence [y_0 .. y_{H-1}, 0 .. 0].  The classic packing applies: z[n] = y[2n] + i y[2n+1]
// (n < H, non-zero only for n < H/2), Z = FFT_H(z), and
//     X[k] = 1/2 (Z[k] + conj Z[H-k]) - i/2 e^{-2 pi i k / N} (Z[k] - conj Z[H-k]),   k = 0 .. H;   |X[N-k]| = |X[k]|
// -- ONE complex transform of H = 200 = 8 * 5 * 5 (25 ms windows) or 160 = 8 * 5 * 4 (20 ms) points per frame, three
// Stockham stages: 20x fewer multiply-adds than the dense [H x 2 bins] product the matrix-core paths run.  A 256-thread
// workgroup owns 16 consecutive frames in LDS: stage 1 (radix 8, whose upper four inputs are the zero half) reads the samples
// straight from HBM, applying gain, pre-emphasis and window on the way; stages 2 and 3 and the unpacking are in place (every
// thread reads its share into registers, barrier, writes); the unpacking leaves magnitudes, which the filterbank (sequential
// f32 sums over each filter's own bins, as the reference), log(x+1) and the DCT consume from LDS.  The mirrored bins of
// src/clip.cpp:186-192 are an index map on the magnitudes.  Rows of the transform buffer are padded by one element in eight
// so that the stride-8 writes of stage 1 fall on distinct banks.
#include <cmath>

#include "srx_mfcc_plan.h"

float srx_filter_get(const srx_filter &f, int bin);  // srx_ctx.hip

namespace {

constexpr int FFT_F = 16;        // frames per workgroup
constexpr int FFT_THREADS = 256;

struct MfccFftArgs {
    const float *x;
    const int64_t *soff, *foff;
    const float *scale;      // may be null
    const int *blk_utt;      // [blocks] utterance of each block's first frame
    const float *win;        // [H]
    const float2 *tw;        // [R0*R1] stage-2 table, [H] stage-3 table, [H+1] unpacking table e^{-2 pi i k / N}
    const float *fbw;        // [ntaps] filter::get over each filter's own bins, filter after filter
    const int *fblo;         // [nfilt] first bin, [nfilt] number of bins, [nfilt] offset of the filter's taps in fbw
    const float *dctm;       // [nfilt rounded to 16][16]
    float *ceps;
    int64_t total_frames;
    int n_utt, hop, N, bin_max, nfilt, ntaps, n_ceps, preemph;
    float alpha, rootn_inv;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }   // a * (+i)

// out[q] = sum_t v[t] e^{-2 pi i q t / R}, in place
__device__ __forceinline__ void dft4(float2 &v0, float2 &v1, float2 &v2, float2 &v3) {
    const float2 a0 = cadd(v0, v2), a1 = csub(v0, v2), a2 = cadd(v1, v3), a3 = mul_mi(csub(v1, v3));
    v0 = cadd(a0, a2);
    v1 = cadd(a1, a3);
    v2 = csub(a0, a2);
    v3 = csub(a1, a3);
}
template <int R>
__device__ __forceinline__ void dft(float2 (&v)[R]);
template <>
__device__ __forceinline__ void dft<4>(float2 (&v)[4]) { dft4(v[0], v[1], v[2], v[3]); }
template <>
__device__ __forceinline__ void dft<5>(float2 (&v)[5]) {
    const float c1 = 0.30901699437494745f, c2 = -0.8090169943749473f, s1 = 0.9510565162951535f, s2 = 0.5877852522924732f;
    const float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const float2 m1 = make_float2(fmaf(c2, t2.x, fmaf(c1, t1.x, v[0].x)), fmaf(c2, t2.y, fmaf(c1, t1.y, v[0].y)));
    const float2 m2 = make_float2(fmaf(c1, t2.x, fmaf(c2, t1.x, v[0].x)), fmaf(c1, t2.y, fmaf(c2, t1.y, v[0].y)));
    const float2 n1 = make_float2(fmaf(s2, t4.x, s1 * t3.x), fmaf(s2, t4.y, s1 * t3.y));
    const float2 n2 = make_float2(fmaf(-s1, t4.x, s2 * t3.x), fmaf(-s1, t4.y, s2 * t3.y));
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = make_float2(m1.x + n1.y, m1.y - n1.x);   // m1 - i n1
    v[4] = make_float2(m1.x - n1.y, m1.y + n1.x);   // m1 + i n1
    v[2] = make_float2(m2.x + n2.y, m2.y - n2.x);
    v[3] = make_float2(m2.x - n2.y, m2.y + n2.x);
}
template <>
__device__ __forceinline__ void dft<8>(float2 (&v)[8]) {
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    const float h = 0.70710678118654752f;
    o1 = make_float2(h * (o1.x + o1.y), h * (o1.y - o1.x));     // * (1 - i) / sqrt 2
    o2 = mul_mi(o2);
    o3 = make_float2(h * (o3.y - o3.x), -h * (o3.x + o3.y));    // * (-1 - i) / sqrt 2
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}
// the 8-point transform of [v0, v1, v2, v3, 0, 0, 0, 0]
__device__ __forceinline__ void dft8_lower_half(const float2 v0, const float2 v1, const float2 v2, const float2 v3, float2 (&out)[8]) {
    // even part = DFT4(v0, v2, 0, 0), odd part = DFT4(v1, v3, 0, 0)
    const float2 e0 = cadd(v0, v2), e1 = cadd(v0, mul_mi(v2)), e2 = csub(v0, v2), e3 = cadd(v0, mul_pi(v2));
    float2 o0 = cadd(v1, v3), o1 = cadd(v1, mul_mi(v3)), o2 = csub(v1, v3), o3 = cadd(v1, mul_pi(v3));
    const float h = 0.70710678118654752f;
    o1 = make_float2(h * (o1.x + o1.y), h * (o1.y - o1.x));
    o2 = mul_mi(o2);
    o3 = make_float2(h * (o3.y - o3.x), -h * (o3.x + o3.y));
    out[0] = cadd(e0, o0); out[4] = csub(e0, o0);
    out[1] = cadd(e1, o1); out[5] = csub(e1, o1);
    out[2] = cadd(e2, o2); out[6] = csub(e2, o2);
    out[3] = cadd(e3, o3); out[7] = csub(e3, o3);
}

__device__ __forceinline__ int fft_find_utt(const int64_t *foff, int n_utt, int64_t f) {
    int lo = 0, hi = n_utt;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (foff[mid] <= f) lo = mid; else hi = mid;
    }
    return lo;
}

// utterance of the first frame of every 16-frame block: the binary search (ten dependent loads) runs here, fully parallel,
// instead of at the head of every workgroup of the transform kernel
__global__ __launch_bounds__(256) void k_mfcc_fft_blocks(const int64_t *__restrict__ foff, int n_utt, int64_t nblk, int *__restrict__ blk_utt) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblk) blk_utt[b] = fft_find_utt(foff, n_utt, b * FFT_F);
}

__device__ __forceinline__ int phys(int idx) { return idx + (idx >> 3); }   // one pad element per eight

template <int R1, int R2>
__global__ __launch_bounds__(FFT_THREADS) void k_mfcc_fft(MfccFftArgs a) {
    constexpr int R0 = 8, H = R0 * R1 * R2, T0 = H / R0, T1 = H / R1, T2 = H / R2, L1 = R0 * R1, HP = H + H / 8;
    constexpr int NB1 = (FFT_F * T1 + FFT_THREADS - 1) / FFT_THREADS, NB2 = (FFT_F * T2 + FFT_THREADS - 1) / FFT_THREADS;
    constexpr int NK = H + 1;                                   // bins 0 .. H of the real transform
    constexpr int NBU = (FFT_F * NK + FFT_THREADS - 1) / FFT_THREADS;
    extern __shared__ float2 fsm[];
    float2 *data = fsm;                       // [F][HP]
    float2 *tw1 = data + FFT_F * HP;          // [L1]
    float2 *tw2 = tw1 + L1;                   // [H]
    float2 *twu = tw2 + H;                    // [H + 1]
    int64_t *fub = reinterpret_cast<int64_t *>(twu + NK);     // [F] utterance start
    int64_t *fp0 = fub + FFT_F;               // [F] first sample of the frame inside its utterance (-1: no such frame)
    float *fsc = reinterpret_cast<float *>(fp0 + FFT_F);      // [F] gain
    float *dcs = fsc + FFT_F;                 // [nfilt][16] DCT matrix
    int *flo = reinterpret_cast<int *>(dcs + a.nfilt * 16);   // [3 nfilt] first bin, bins, tap offset of every filter
    float *fws = reinterpret_cast<float *>(flo + 3 * a.nfilt);   // [ntaps] filter taps
    const int tid = threadIdx.x;
    const int64_t F0 = (int64_t)blockIdx.x * FFT_F;
    for (int i = tid; i < a.nfilt * 16; i += FFT_THREADS) dcs[i] = a.dctm[i];
    for (int i = tid; i < 3 * a.nfilt; i += FFT_THREADS) flo[i] = a.fblo[i];
    for (int i = tid; i < a.ntaps; i += FFT_THREADS) fws[i] = a.fbw[i];
    for (int i = tid; i < L1; i += FFT_THREADS) tw1[i] = a.tw[i];
    for (int i = tid; i < H; i += FFT_THREADS) tw2[i] = a.tw[L1 + i];
    for (int i = tid; i < NK; i += FFT_THREADS) twu[i] = a.tw[L1 + H + i];
    if (tid < FFT_F) {
        const int64_t f = F0 + tid;
        int64_t ub = 0, p0 = -1;
        float sc = 1.0f;
        if (f < a.total_frames) {
            int u = a.blk_utt[blockIdx.x];
            while (u + 1 < a.n_utt && a.foff[u + 1] <= f) u++;
            ub = a.soff[u];
            p0 = (f - a.foff[u]) * (int64_t)a.hop;
            if (a.scale) sc = a.scale[u];
        }
        fub[tid] = ub;
        fp0[tid] = p0;
        fsc[tid] = sc;
    }
    __syncthreads();

    // ---- stage 1 (radix 8, Ns = 1: no twiddles): z[n] = y[2n] + i y[2n+1], n = i + t T0; t >= 4 is the zero half ----
    for (int it = tid; it < FFT_F * T0; it += FFT_THREADS) {
        const int f = it / T0, i = it - f * T0;
        const int64_t p0 = fp0[f], ub = fub[f];
        const float sc = fsc[f];
        float2 z[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int n = 2 * (i + t * T0);
            float y0 = 0.0f, y1 = 0.0f;
            if (p0 >= 0) {
                const int64_t p = p0 + n;
                float v0, v1;
                if (a.preemph) {
                    const int64_t q = p == 0 ? 1 : p;      // x[0] = x[1] after the filter (src/clip.cpp:204)
                    float xm = a.x[ub + q - 1], xa = a.x[ub + q], xb = a.x[ub + p + 1];
                    if (a.scale) { xm *= sc; xa *= sc; xb *= sc; }
                    const float m0 = a.alpha * xm;
                    v0 = xa - m0;
                    // the sample after: its left neighbour is x[p] itself (p + 1 >= 1: no special case)
                    const float m1 = a.alpha * (p == 0 ? xm : xa);
                    v1 = xb - m1;
                } else {
                    v0 = a.x[ub + p];
                    v1 = a.x[ub + p + 1];
                    if (a.scale) { v0 *= sc; v1 *= sc; }
                }
                y0 = v0 * a.win[n];
                y1 = v1 * a.win[n + 1];
            }
            z[t] = make_float2(y0, y1);
        }
        float2 o[8];
        dft8_lower_half(z[0], z[1], z[2], z[3], o);
        float2 *d = data + f * HP;
#pragma unroll
        for (int q = 0; q < 8; q++) d[phys(i * 8 + q)] = o[q];
    }
    __syncthreads();

    // ---- stage 2 (radix R1, Ns = 8), in place ----
    {
        float2 v[NB1][R1];
#pragma unroll
        for (int b = 0; b < NB1; b++) {
            const int it = tid + b * FFT_THREADS;
            if (it < FFT_F * T1) {
                const int f = it / T1, i = it - f * T1;
#pragma unroll
                for (int t = 0; t < R1; t++) v[b][t] = data[f * HP + phys(i + t * T1)];
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB1; b++) {
            const int it = tid + b * FFT_THREADS;
            if (it < FFT_F * T1) {
                const int f = it / T1, i = it - f * T1;
                const int k = i % R0;
#pragma unroll
                for (int t = 1; t < R1; t++) v[b][t] = cmul(v[b][t], tw1[k * t]);
                dft<R1>(v[b]);
                const int base = (i - k) * R1 + k;
#pragma unroll
                for (int q = 0; q < R1; q++) data[f * HP + phys(base + q * R0)] = v[b][q];
            }
        }
    }
    __syncthreads();

    // ---- stage 3 (radix R2, Ns = 8 R1 = T2), in place ----
    {
        float2 v[NB2][R2];
#pragma unroll
        for (int b = 0; b < NB2; b++) {
            const int it = tid + b * FFT_THREADS;
            if (it < FFT_F * T2) {
                const int f = it / T2, i = it - f * T2;
#pragma unroll
                for (int t = 0; t < R2; t++) v[b][t] = data[f * HP + phys(i + t * T2)];
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB2; b++) {
            const int it = tid + b * FFT_THREADS;
            if (it < FFT_F * T2) {
                const int f = it / T2, i = it - f * T2;     // k = i: Ns = T2
#pragma unroll
                for (int t = 1; t < R2; t++) v[b][t] = cmul(v[b][t], tw2[i * t]);
                dft<R2>(v[b]);
#pragma unroll
                for (int q = 0; q < R2; q++) data[f * HP + phys(i + q * T2)] = v[b][q];
            }
        }
    }
    __syncthreads();

    // ---- unpack the real transform: |X[k]| / sqrt(N) for k = 0 .. H, in place of Z ----
    float *M = reinterpret_cast<float *>(data);      // [F][2 HP] floats per frame: magnitudes in the first H + 1
    {
        float2 za[NBU], zb[NBU];
#pragma unroll
        for (int b = 0; b < NBU; b++) {
            const int it = tid + b * FFT_THREADS;
            if (it < FFT_F * NK) {
                const int f = it / NK, k = it - f * NK;
                za[b] = data[f * HP + phys(k == H ? 0 : k)];
                zb[b] = data[f * HP + phys(k == 0 ? 0 : H - k)];
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NBU; b++) {
            const int it = tid + b * FFT_THREADS;
            if (it < FFT_F * NK) {
                const int f = it / NK, k = it - f * NK;
                const float2 A = za[b], B = make_float2(zb[b].x, -zb[b].y);
                const float2 s = cadd(A, B), d = cmul(twu[k], csub(A, B));     // X = 1/2 (s - i d)
                const float re = 0.5f * (s.x + d.y), im = 0.5f * (s.y - d.x);
                const float r2 = re * re;
                const float i2 = im * im;
                M[f * (2 * HP) + k] = sqrtf(r2 + i2) * a.rootn_inv;
            }
        }
    }
    __syncthreads();

    // ---- filterbank + log(x+1) (src/mfcc.cpp:143-167, src/utils.hpp:55-60): sequential f32 sums over each filter's bins ----
    float *LF = M + NK + 1;                   // [F] rows of nfilt floats behind each frame's magnitudes (2 HP >= H + 2 + nfilt)
    const int N = a.N;
    const bool mirror = a.bin_max > N / 2;
    // 16 lanes = the 16 frames of one filter: a wave's four filters are neighbours, of about the same width
    for (int it = tid; it < FFT_F * a.nfilt; it += FFT_THREADS) {
        const int f = it & (FFT_F - 1), i = it >> 4;
        const int lo = flo[i], nb = flo[a.nfilt + i];
        const float *w = fws + flo[2 * a.nfilt + i];
        const float *m = M + (size_t)f * (2 * HP);
        float res = 0.0f;
        for (int j = 0; j < nb; j++) {
            const int b = lo + j;
            int kk = b;
            if (mirror && b >= N - a.bin_max) kk = b > N - 1 - b ? b : N - 1 - b;   // the later of the two writes (src/clip.cpp:186-192)
            if (kk > H) kk = N - kk;                                                 // |X[N-k]| = |X[k]|: real input
            const float tap = w[j] * m[kk];
            res += tap;
        }
        LF[f * (2 * HP) + i] = logf(res + 1.0f);
    }
    __syncthreads();
    // ---- DCT (src/mfcc.cpp:205-214) ----
    for (int it = tid; it < FFT_F * 16; it += FFT_THREADS) {
        const int f = it >> 4, j = it & 15;
        const int64_t fr = F0 + f;
        if (fr >= a.total_frames || j >= a.n_ceps) continue;
        float c = 0.0f;
        for (int i = 0; i < a.nfilt; i++) c = fmaf(LF[f * (2 * HP) + i], dcs[i * 16 + j], c);
        a.ceps[fr * a.n_ceps + j] = c;
    }
}

bool fft_radices(int half, int (&r)[3]) {
    switch (half) {
        case 200: r[0] = 8; r[1] = 5; r[2] = 5; return true;
        case 160: r[0] = 8; r[1] = 5; r[2] = 4; return true;
        case 128: r[0] = 8; r[1] = 4; r[2] = 4; return true;
        case 256: r[0] = 8; r[1] = 8; r[2] = 4; return true;
        default: return false;
    }
}

size_t fft_lds_bytes(int H, int L1, int nfilt, int ntaps) {
    return (size_t)(FFT_F * (H + H / 8) + L1 + H + H + 1) * sizeof(float2) + 2 * FFT_F * sizeof(int64_t) + FFT_F * sizeof(float) +
           (size_t)nfilt * 16 * sizeof(float) + (size_t)3 * nfilt * sizeof(int) + (size_t)ntaps * sizeof(float);
}

}  // namespace

bool srx_mfccfft_supported(int window_len, int n_filters, int n_ceps) {
    int r[3];
    return window_len % 2 == 0 && fft_radices(window_len / 2, r) && n_filters <= 64 && n_ceps <= 16;
}

// Tables of the FFT path: twiddles in f64 rounded once, filter taps from filter::get over each filter's own bin range.
int srx_mfccfft_build(srx_mfcc_plan *p, const srx_filter *fb) {
    srx_ctx *ctx = p->ctx;
    int r[3];
    if (!fft_radices(p->half, r)) return SRX_OK;
    const int H = p->half, N = 2 * H, L1 = r[0] * r[1];
    std::vector<float2> tw((size_t)L1 + H + H + 1);
    for (int q = 0; q < L1; q++) tw[q] = make_float2((float)cos(-2 * M_PI * q / L1), (float)sin(-2 * M_PI * q / L1));
    for (int q = 0; q < H; q++) tw[L1 + q] = make_float2((float)cos(-2 * M_PI * q / H), (float)sin(-2 * M_PI * q / H));
    for (int k = 0; k <= H; k++) tw[L1 + H + k] = make_float2((float)cos(-2 * M_PI * k / N), (float)sin(-2 * M_PI * k / N));
    int ntaps = 0;
    std::vector<int> lo((size_t)3 * p->nfilt);
    for (int i = 0; i < p->nfilt; i++) {
        lo[i] = (int)(fb[i].center - fb[i].width);
        lo[p->nfilt + i] = (int)(fb[i].center + fb[i].width) - lo[i] + 1;
        lo[2 * p->nfilt + i] = ntaps;
        ntaps += lo[p->nfilt + i];
    }
    std::vector<float> w((size_t)(ntaps ? ntaps : 1), 0.0f);
    for (int i = 0; i < p->nfilt; i++)
        for (int j = 0; j < lo[p->nfilt + i]; j++) w[(size_t)lo[2 * p->nfilt + i] + j] = srx_filter_get(fb[i], lo[i] + j);
    SRX_HIP(ctx, hipMalloc((void **)&p->fft_tw, tw.size() * sizeof(float2)));
    SRX_HIP(ctx, hipMemcpy(p->fft_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    SRX_HIP(ctx, hipMalloc((void **)&p->fft_fbw, w.size() * sizeof(float)));
    SRX_HIP(ctx, hipMemcpy(p->fft_fbw, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    SRX_HIP(ctx, hipMalloc((void **)&p->fft_fblo, lo.size() * sizeof(int)));
    SRX_HIP(ctx, hipMemcpy(p->fft_fblo, lo.data(), lo.size() * sizeof(int), hipMemcpyHostToDevice));
    p->fft_fbt = ntaps;
    p->fft_r[0] = r[0]; p->fft_r[1] = r[1]; p->fft_r[2] = r[2];
    return SRX_OK;
}

int srx_mfccfft_launch(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples_dev, const int64_t *sample_off_dev, int n_utt,
                       const int64_t *frame_off_dev, int64_t total_frames, const float *scale_dev, float *ceps_dev) {
    const srx_mfcc_cfg &c = plan->cfg;
    MfccFftArgs a;
    a.x = samples_dev; a.soff = sample_off_dev; a.foff = frame_off_dev; a.scale = scale_dev;
    a.win = plan->win; a.tw = (const float2 *)plan->fft_tw; a.fbw = plan->fft_fbw; a.fblo = plan->fft_fblo; a.dctm = plan->dctm;
    a.ceps = ceps_dev; a.total_frames = total_frames; a.n_utt = n_utt; a.hop = c.frame_shift; a.N = c.window_len; a.bin_max = plan->bin_max;
    a.nfilt = plan->nfilt; a.ntaps = plan->fft_fbt; a.n_ceps = c.n_ceps; a.preemph = c.preemphasis; a.alpha = c.alpha; a.rootn_inv = plan->rootn_inv;
    const int H = plan->half, L1 = plan->fft_r[0] * plan->fft_r[1];
    const size_t lds = fft_lds_bytes(H, L1, plan->nfilt, plan->fft_fbt);
    SRX_REQUIRE(ctx, lds <= 160 * 1024, "srx_mfcc_batch_dev: FFT path needs %zu bytes of LDS", lds);
    // a frame's magnitudes [H + 1] and log filterbank outputs [nfilt] reuse its row of the transform buffer
    SRX_REQUIRE(ctx, H + 2 + plan->nfilt <= 2 * (H + H / 8), "srx_mfcc_batch_dev: FFT path: too many filters for the window");
    const int64_t nblk = ceil_div64(total_frames, FFT_F);
    SRX_REQUIRE(ctx, nblk < (1ll << 31), "srx_mfcc_batch_dev: too many frames");
    void *ws = nullptr;
    int rc = srx_workspace(ctx, 4, (size_t)nblk * sizeof(int), &ws);
    if (rc != SRX_OK) return rc;
    a.blk_utt = (const int *)ws;
    static bool attr_set[SRX_MAX_DEVICES] = {};
    if (!attr_set[ctx->device]) {
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_mfcc_fft<5, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_mfcc_fft<5, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_mfcc_fft<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_mfcc_fft<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[ctx->device] = true;
    }
    {
        SrxProf prof_(ctx, "k_mfcc_fft_blocks");
        hipLaunchKernelGGL(k_mfcc_fft_blocks, dim3((unsigned)ceil_div64(nblk, 256)), dim3(256), 0, ctx->stream, frame_off_dev, n_utt, nblk, (int *)ws);
    }
    SRX_LAUNCH_CHECK(ctx);
    {
        SrxProf prof_(ctx, "k_mfcc_fft");
        if (H == 200) hipLaunchKernelGGL((k_mfcc_fft<5, 5>), dim3((unsigned)nblk), dim3(FFT_THREADS), lds, ctx->stream, a);
        else if (H == 160) hipLaunchKernelGGL((k_mfcc_fft<5, 4>), dim3((unsigned)nblk), dim3(FFT_THREADS), lds, ctx->stream, a);
        else if (H == 128) hipLaunchKernelGGL((k_mfcc_fft<4, 4>), dim3((unsigned)nblk), dim3(FFT_THREADS), lds, ctx->stream, a);
        else if (H == 256) hipLaunchKernelGGL((k_mfcc_fft<8, 4>), dim3((unsigned)nblk), dim3(FFT_THREADS), lds, ctx->stream, a);
        else return srx_fail(ctx, SRX_ERR_UNSUPPORTED, "srx_mfcc_batch_dev: FFT path has no kernel for a half window of %d", H);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}
