// Embedded-training statistics as a sort-by-state segmented reduction (tables too large for the LDS-privatised
// accumulation of srx_train.hip, i.e. BASELINE shapes: G*M = 32,000 rows of 1+D doubles = 10 MB).
//
// Replaces the same reference loops as k_em_accum (hmm::train passes, src/hmm.cpp:298-339).  Frames are sorted by
// tied state once per iteration (a stable counting sort written here: within a state they stay in frame order, so
// sums are deterministic -- no atomics anywhere); a workgroup then owns up to CH consecutive frames of ONE state and keeps
// that state's M x D sums in registers while the frames' zetas and feature rows pass through LDS (pass 2: the squared
// f32 difference to the new mean depends on component AND column, so it is no matrix product; an LDS-free form with
// one wave per item and 128-bit row loads was measured at 10.8 ms against this form's 9.1 -- the f64 <-> f32 conversions
// it repeats per component cost more than the LDS reads it saves).  Pass 1 IS a matrix product per state and runs on
// the f64 matrix cores (k_em_accum1_mfma).  States with more than CH frames are split and their partial sums added in
// chunk order by a second small kernel.
// At G = 2000, M = 16, D = 39 the HBM-atomics form spends 157 ms per pass on 5e6 frames; this form is bound by
// reading logp and the feature rows once.
#include "srx_common.h"

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 128-bit load from a 4-byte aligned address

// exp(y) in f64 for the zeta of the accumulation passes (y = logp - shift: at most a little above 0, down to -inf).
// n = rint(y / ln 2), r = y - n ln 2 in two parts, degree-13 Taylor polynomial on |r| <= 0.347 (remainder 4e-18),
// v_ldexp_f64 for the scale (gradual underflow like exp()).  18 f64 instructions, no branches; checked in the
// statistics tests at rtol 1e-9 (tests/test_gpu_train.py).  NaN stays NaN; below -745.2 the result is 0.
__device__ __forceinline__ double srx_exp_f64(double y) {
    const double n = rint(y * 1.4426950408889634);
    double r = fma(-n, 6.93147180369123816490e-01, y);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;               // 1/13!
    p = fma(p, r, 2.08767569878681e-09);             // 1/12!
    p = fma(p, r, 2.505210838544172e-08);            // 1/11!
    p = fma(p, r, 2.755731922398589e-07);            // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);           // 1/9!
    p = fma(p, r, 2.48015873015873e-05);             // 1/8!
    p = fma(p, r, 1.984126984126984e-04);            // 1/7!
    p = fma(p, r, 1.388888888888889e-03);            // 1/6!
    p = fma(p, r, 8.333333333333333e-03);            // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);           // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);           // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double v = ldexp(p, (int)n);
    return y < -745.2 ? 0.0 : v;                     // (also y = -inf, where r is NaN)
}

static constexpr int CH = 2048;     // frames per work item
static constexpr int FB = 32;       // frames staged per step

static constexpr int CS_CHUNK = 4096;   // frames per wave of the counting sort
static constexpr int CS_GROUP = 64;     // chunks per group of the two-level scan

struct EmSortView {
    int32_t *order;       // [T] frame indices sorted by state (skipped frames, gid < 0, last)
    int32_t *hist;        // [n_chunks][G+1] per-chunk key counts, then (in place) each chunk's offset inside its group of chunks
    int32_t *gsum;        // [n_groups][G+1] per-group key counts, then (in place) each group's offset inside its key's segment
    int64_t *total;       // [G+1] frames per key
    int64_t *start;       // [G+2] first sorted position of state g; start[G] = aligned frames; start[G+1] = T
    int32_t *item_start;  // [G+1] exclusive scan of ceil(n_g / CH)
};

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }
static int64_t cs_chunks(int64_t T) { return (T + CS_CHUNK - 1) / CS_CHUNK; }

static void em_sort_view(void *base, int64_t T, int G, EmSortView *v) {
    char *p = reinterpret_cast<char *>(base);
    v->order = reinterpret_cast<int32_t *>(p);    p += align256((size_t)T * 4);
    v->hist = reinterpret_cast<int32_t *>(p);     p += align256((size_t)cs_chunks(T) * (G + 1) * 4);
    v->gsum = reinterpret_cast<int32_t *>(p);     p += align256((size_t)(cs_chunks(T) / 64 + 1) * (G + 1) * 4);
    v->total = reinterpret_cast<int64_t *>(p);    p += align256((size_t)(G + 2) * 8);
    v->start = reinterpret_cast<int64_t *>(p);    p += align256((size_t)(G + 2) * 8);
    v->item_start = reinterpret_cast<int32_t *>(p);
}
static size_t em_sort_bytes(int64_t T, int G) {
    return align256((size_t)T * 4) + align256((size_t)cs_chunks(T) * (G + 1) * 4) + align256((size_t)(cs_chunks(T) / 64 + 1) * (G + 1) * 4) +
           2 * align256((size_t)(G + 2) * 8) + align256((size_t)(G + 2) * 4);
}

// ---- stable counting sort of the frame indices by tied state -------------------------------------------------------
// Keys are 0 .. G (G = "skipped": gid < 0 or out of range); a Viterbi path stays in a state for a dozen frames, so the
// 64 keys a wave looks at hold a handful of distinct values.  A wave owns CS_CHUNK consecutive frames and walks them 64
// at a time; the lanes that share a key are found with one ballot per distinct key (no atomics: the first lane of a
// key group updates the LDS counter for the whole group), which also gives a lane its rank inside the group -- so the
// scatter is stable.  Pass 1 counts per (chunk, key); a scan over the chunks of a key and then over the keys turns the
// counts into positions; pass 2 re-reads the keys and writes the indices.
__device__ __forceinline__ int cs_key(const int32_t *__restrict__ gid, int64_t t, int G) {
    const int g = gid[t];
    return (g < 0 || g >= G) ? G : g;
}

// for the lanes in `valid`: cnt = lanes of the wave with this lane's key, rank = those among them below this lane,
// leader = this lane is the lowest of its group
__device__ __forceinline__ void cs_match(int key, unsigned long long valid, int lane, int &cnt, int &rank, bool &leader) {
    unsigned long long remaining = valid;
    cnt = 0; rank = 0; leader = false;
    while (remaining) {
        const int l0 = __builtin_ctzll(remaining);
        const int k0 = __shfl(key, l0, 64);
        const unsigned long long m = __ballot(key == k0) & valid;
        if (key == k0 && ((valid >> lane) & 1ull)) {
            cnt = __builtin_popcountll(m);
            rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
            leader = lane == l0;
        }
        remaining &= ~m;
    }
}

__global__ __launch_bounds__(64) void k_em_csort_hist(const int32_t *__restrict__ gid, int64_t T, int G, int32_t *__restrict__ hist) {
    extern __shared__ int32_t cs_h[];   // [G+1]
    const int lane = threadIdx.x, Gp = G + 1;
    const int64_t c = blockIdx.x, t0 = c * CS_CHUNK;
    for (int i = lane; i < Gp; i += 64) cs_h[i] = 0;
    __syncthreads();
    const int64_t t1 = t0 + CS_CHUNK < T ? t0 + CS_CHUNK : T;
    for (int64_t tb = t0; tb < t1; tb += 64) {
        const int64_t t = tb + lane;
        const bool v = t < t1;
        const int key = v ? cs_key(gid, t, G) : -1;
        int cnt, rank;
        bool leader;
        cs_match(key, __ballot(v), lane, cnt, rank, leader);
        if (leader) cs_h[key] += cnt;      // distinct keys per leader: no two lanes touch one counter
    }
    __syncthreads();
    for (int i = lane; i < Gp; i += 64) hist[c * Gp + i] = cs_h[i];
}

// Two-level scan over the chunks of a key (a single serial walk over 12,000 chunks per key took 3.6 ms):
//   a) thread (key, group of CS_GROUP chunks): hist[c][g] <- number of key-g frames in the group's chunks before c;
//      gsum[group][g] = the group's total (loads first, then the prefix, then the stores: nothing waits on a store);
//   b) thread (key): gsum[group][g] <- frames in groups before it; total[g] = all of them.
__global__ __launch_bounds__(256) void k_em_csort_scan_a(int32_t *__restrict__ hist, int64_t n_chunks, int Gp, int32_t *__restrict__ gsum) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= Gp) return;
    const int64_t q = blockIdx.y, c0 = q * CS_GROUP;
    int32_t v[CS_GROUP];
#pragma unroll
    for (int i = 0; i < CS_GROUP; i++) v[i] = c0 + i < n_chunks ? hist[(c0 + i) * Gp + g] : 0;
    int32_t run = 0;
#pragma unroll
    for (int i = 0; i < CS_GROUP; i++) {
        const int32_t x = v[i];
        v[i] = run;
        run += x;
    }
#pragma unroll
    for (int i = 0; i < CS_GROUP; i++)
        if (c0 + i < n_chunks) hist[(c0 + i) * Gp + g] = v[i];
    gsum[q * Gp + g] = run;
}
__global__ __launch_bounds__(256) void k_em_csort_scan_b(int32_t *__restrict__ gsum, int64_t n_groups, int Gp, int64_t *__restrict__ total) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= Gp) return;
    int64_t run = 0;
    for (int64_t q0 = 0; q0 < n_groups; q0 += 16) {
        int32_t v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = q0 + i < n_groups ? gsum[(q0 + i) * Gp + g] : 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (q0 + i < n_groups) gsum[(q0 + i) * Gp + g] = (int32_t)run;    // < 2^31: T is
            run += v[i];
        }
    }
    total[g] = run;
}

// start[g] = sum of total[0 .. g), g = 0 .. G+1 (one block; G is at most a few tens of thousands)
__global__ __launch_bounds__(1024) void k_em_csort_starts(const int64_t *__restrict__ total, int Gp, int64_t *__restrict__ start) {
    __shared__ int64_t part[1024];
    const int per = (Gp + 1023) / 1024;
    const int g0 = threadIdx.x * per;
    int64_t s = 0;
    for (int g = g0; g < g0 + per && g < Gp; g++) s += total[g];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int64_t run = part[threadIdx.x] - s;
    for (int g = g0; g < g0 + per && g < Gp; g++) {
        start[g] = run;
        run += total[g];
    }
    if (threadIdx.x == 1023) start[Gp] = part[1023];
}

__global__ __launch_bounds__(64) void k_em_csort_scatter(const int32_t *__restrict__ gid, int64_t T, int G, const int32_t *__restrict__ hist,
                                                         const int32_t *__restrict__ gsum, const int64_t *__restrict__ start,
                                                         int32_t *__restrict__ order) {
    extern __shared__ int32_t cs_h[];   // [G+1] next position of each key (positions are < 2^31)
    const int lane = threadIdx.x, Gp = G + 1;
    const int64_t c = blockIdx.x, t0 = c * CS_CHUNK;
    for (int i = lane; i < Gp; i += 64) cs_h[i] = (int32_t)(start[i] + gsum[(c / CS_GROUP) * Gp + i] + hist[c * Gp + i]);
    __syncthreads();
    const int64_t t1 = t0 + CS_CHUNK < T ? t0 + CS_CHUNK : T;
    for (int64_t tb = t0; tb < t1; tb += 64) {
        const int64_t t = tb + lane;
        const bool v = t < t1;
        const int key = v ? cs_key(gid, t, G) : -1;
        int cnt, rank;
        bool leader;
        cs_match(key, __ballot(v), lane, cnt, rank, leader);
        int base = 0;
        if (leader) {
            base = cs_h[key];
            cs_h[key] = base + cnt;
        }
        // every lane of a group takes the base its leader read: the groups are few, so each leader's (key, base) is
        // broadcast in turn
        int my_base = 0;
        unsigned long long leaders = __ballot(leader);
        while (leaders) {
            const int l0 = __builtin_ctzll(leaders);
            const int k0 = __shfl(key, l0, 64);
            const int b0 = __shfl(base, l0, 64);
            if (key == k0) my_base = b0;
            leaders &= leaders - 1;
        }
        if (v) order[my_base + rank] = (int32_t)t;
    }
}

// item_start = exclusive scan of ceil(n_g / CH) over states (one block; G is at most a few tens of thousands)
__global__ __launch_bounds__(1024) void k_em_item_scan(const int64_t *__restrict__ start, int G, int32_t *__restrict__ item_start) {
    __shared__ int part[1024];
    const int per = (G + 1023) / 1024;
    const int g0 = threadIdx.x * per;
    int s = 0;
    for (int g = g0; g < g0 + per && g < G; g++) s += (int)((start[g + 1] - start[g] + CH - 1) / CH);
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int g = g0; g < g0 + per && g < G; g++) {
        item_start[g] = run;
        run += (int)((start[g + 1] - start[g] + CH - 1) / CH);
    }
    if (threadIdx.x == 1023) item_start[G] = part[1023];
}

// PASS 2: one work item = up to CH frames of one state, sums [zeta * diff_d^2], diff = f32(x_d - meanB[o][d])
// (src/hmm.cpp:331).  partial[item][M*D].
// Thread = (frame parity h = tid / 128, component pair mp = (tid / 16) % 8 -> components mp and mp + 8, column dq = tid % 16)
// with the row elements e = dq, dq + 16, dq + 32, dq + 48 of both components in registers: per frame of its parity a thread
// reads two zetas and NACC staged values from LDS for 2 * NACC sums (the kernel is bound by those LDS reads: one component
// per thread needed 1 + NACC reads for NACC sums).  The staged values are already f64 and the frame loop is unrolled.  Sums
// run in frame order within a parity, one fused multiply-add per term (f64; the reference sums in 80-bit long double,
// src/hmm.cpp:319-334); the two parities are added at the end.
template <int NACC>
__global__ __launch_bounds__(256) void k_em_accum_seg2(const float *__restrict__ feats, int ld, const int32_t *__restrict__ order,
                                                       const int64_t *__restrict__ start, const int32_t *__restrict__ item_start, int G,
                                                       int M, int D, const float *__restrict__ weight, const float *__restrict__ logp,
                                                       const float *__restrict__ shift, const double *__restrict__ meanB,
                                                       double *__restrict__ partial) {
    __shared__ double zeta[FB][16];
    __shared__ double xv[FB][65];
    __shared__ int ft[2][FB];
    const int item = blockIdx.x;
    if (item >= item_start[G]) return;
    int lo = 0, hi = G;   // item_start[lo] <= item < item_start[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (item_start[mid] <= item) lo = mid; else hi = mid;
    }
    const int g = lo;
    const int64_t p0 = start[g] + (int64_t)(item - item_start[g]) * CH;
    const int64_t p1 = p0 + CH < start[g + 1] ? p0 + CH : start[g + 1];
    const int stride = D;
    const int R = M * stride;
    const int h = threadIdx.x >> 7, mp = (threadIdx.x >> 4) & 7, dq = threadIdx.x & 15;
    double acc[2][NACC], mb[2][NACC];
    bool live[2][NACC];   // NACC = ceil(D / 16): only the last accumulator can be past the row's end
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            const int m = mp + 8 * c, e = dq + 16 * a;
            acc[c][a] = 0.0;
            live[c][a] = m < M && e < stride;
            mb[c][a] = live[c][a] ? meanB[((size_t)g * M + m) * D + e] : 0.0;
        }
    for (int i = threadIdx.x; i < FB * 16; i += 256) (&zeta[0][0])[i] = 0.0;   // components >= M and the tail frames stay 0
    // The step's inputs are gathered one step ahead into registers (each thread's share of the FB x M log-densities
    // and the FB x D feature values), so their latency passes under the previous step's sums instead of in
    // front of a barrier; the frame indices run one step further ahead still (register, then LDS).
    constexpr int NZ = 2;   // FB * M / 256 with M <= 16
    constexpr int NX = 8;   // FB * D / 256 with D <= 64
    float zr[NZ], xr[NX];
    double zw[NZ], zs[NZ];  // weight and shift of the components this thread stages
#pragma unroll
    for (int k = 0; k < NZ; k++) {
        const int i = threadIdx.x + 256 * k;
        const int mm = i % M;
        zw[k] = i < FB * M ? (double)weight[(size_t)g * M + mm] : 0.0;
        zs[k] = i < FB * M ? (double)shift[(size_t)g * M + mm] : 0.0;
    }
    auto gather = [&](int buf, int nf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NZ; k++) {
            const int i = threadIdx.x + 256 * k;
            const int f = i / M, mm = i - f * M;
            zr[k] = (i < FB * M && f < nf) ? logp[(int64_t)ft[buf][f] * M + mm] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NX; k++) {
            const int i = threadIdx.x + 256 * k;
            const int f = i / stride, e = i - f * stride;
            xr[k] = (i < FB * stride && f < nf) ? feats[(int64_t)ft[buf][f] * ld + e] : 0.0f;
        }
    };
    const int64_t nsteps = (p1 - p0 + FB - 1) / FB;
    auto frames_of = [&](int64_t s) { const int64_t r = p1 - (p0 + s * FB); return r < FB ? (int)(r < 0 ? 0 : r) : FB; };
    int ord = 0;
    if (threadIdx.x < frames_of(0)) ft[0][threadIdx.x] = order[p0 + threadIdx.x];
    if (threadIdx.x < frames_of(1)) ord = order[p0 + FB + threadIdx.x];
    __syncthreads();
    gather(0, frames_of(0));
    for (int64_t st = 0; st < nsteps; st++) {
        const int nf = frames_of(st);
        __syncthreads();   // the previous step's reads of zeta / xv are done
#pragma unroll
        for (int k = 0; k < NZ; k++) {
            const int i = threadIdx.x + 256 * k;
            const int f = i / M, mm = i - f * M;
            // zeta = weight * prob (src/hmm.cpp:328), shifted; frames past the item's end contribute 0
            if (i < FB * M) zeta[f][mm] = f < nf ? zw[k] * srx_exp_f64((double)zr[k] - zs[k]) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < NX; k++) {
            const int i = threadIdx.x + 256 * k;
            const int f = i / stride, e = i - f * stride;
            if (i < FB * stride) xv[f][e] = f < nf ? (double)xr[k] : 0.0;
        }
        if (threadIdx.x < frames_of(st + 1)) ft[(st + 1) & 1][threadIdx.x] = ord;
        __syncthreads();
        if (st + 1 < nsteps) {
            gather((int)((st + 1) & 1), frames_of(st + 1));
            if (threadIdx.x < frames_of(st + 2)) ord = order[p0 + (st + 2) * FB + threadIdx.x];
        }
#pragma unroll 8
        for (int f2 = 0; f2 < FB / 2; f2++) {
            const int f = 2 * f2 + h;
            const double z0 = zeta[f][mp], z1 = zeta[f][mp + 8];
#pragma unroll
            for (int a = 0; a < NACC; a++) {
                // no test on live[][] here: a dead accumulator sums staged garbage that is never stored
                const double x = xv[f][dq + 16 * a];
                // (rounding the difference to 24 bits with f64 arithmetic -- Veltkamp's split, three operations -- instead of
                //  the two conversions was measured slower: 9.9 ms against 8.6 per 5e7 frames)
                const float d0 = (float)(x - mb[0][a]), d1 = (float)(x - mb[1][a]);
                acc[0][a] = fma(z0 * (double)d0, (double)d0, acc[0][a]);
                acc[1][a] = fma(z1 * (double)d1, (double)d1, acc[1][a]);
            }
        }
    }
    // odd-frame sums onto the even-frame ones (through the staging array: 128 threads x 2 x NACC <= FB x 65 doubles)
    __syncthreads();
    double *comb = &xv[0][0];
    if (h == 1) {
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int a = 0; a < NACC; a++) comb[((threadIdx.x & 127) * 2 + c) * NACC + a] = acc[c][a];
    }
    __syncthreads();
    if (h == 0) {
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int a = 0; a < NACC; a++)
                if (live[c][a])
                    partial[(size_t)item * R + (mp + 8 * c) * stride + dq + 16 * a] = acc[c][a] + comb[(threadIdx.x * 2 + c) * NACC + a];
    }
}

typedef double f64x4 __attribute__((ext_vector_type(4)));

// PASS 1 on the f64 matrix cores: stats1[g][m][e] = sum_t zeta[t][m] * xx[t][e], xx = [1, x_0 .. x_{D-1}], is a
// [M x T_g] . [T_g x (1+D)] product per state.  One wave per work item (<= CH frames of one state); per group of four
// frames one v_mfma_f64_16x16x4_f64 per 16 columns: A = zeta^T (lane l holds zeta of frame l>>4, component l&15 -- each
// (frame, component) exponential is computed exactly once, by the lane that feeds it), B = xx (lane l holds column
// 16c + (l&15) of frame l>>4, straight from the feature row).  No LDS, no barrier: the LDS form spent its time
// re-reading zeta and x from LDS for every fused multiply-add (6.9 ms per 5e7 frames against 1.2 ms of f64 matrix time).
// Order indices run two 32-frame blocks ahead of the sums and the log-densities / feature values one block ahead.
template <int NCT>
__global__ __launch_bounds__(64) void k_em_accum1_mfma(const float *__restrict__ feats, int ld, const int32_t *__restrict__ order,
                                                       const int64_t *__restrict__ start, const int32_t *__restrict__ item_start, int G,
                                                       int M, int D, const float *__restrict__ weight, const float *__restrict__ logp,
                                                       const float *__restrict__ shift, double *__restrict__ partial) {
    const int item = blockIdx.x;
    if (item >= item_start[G]) return;
    int lo = 0, hi = G;   // item_start[lo] <= item < item_start[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (item_start[mid] <= item) lo = mid; else hi = mid;
    }
    const int g = lo;
    const int64_t p0 = start[g] + (int64_t)(item - item_start[g]) * CH;
    const int64_t p1 = p0 + CH < start[g + 1] ? p0 + CH : start[g + 1];
    const int lane = threadIdx.x, m = lane & 15, k = lane >> 4;
    const int stride = 1 + D, R = M * stride;
    const double wm = m < M ? (double)weight[(size_t)g * M + m] : 0.0;
    const double sm = m < M ? (double)shift[(size_t)g * M + m] : 0.0;
    constexpr int GB = 8;                       // groups of four frames per block
    f64x4 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; c++) acc[c] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const int64_t nb = (p1 - p0 + 4 * GB - 1) / (4 * GB);
    auto load_ord = [&](int64_t b, int (&o)[GB]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < GB; j++) {
            const int64_t pos = p0 + (b * GB + j) * 4 + k;
            o[j] = pos < p1 ? order[pos] : -1;
        }
    };
    auto load_data = [&](const int (&o)[GB], float (&lp)[GB], float (&x)[GB][NCT]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < GB; j++) {
            const bool v = o[j] >= 0;
            const int64_t t = v ? o[j] : 0;
            lp[j] = (v && m < M) ? logp[t * M + m] : -__builtin_huge_valf();   // -inf: zeta = 0
#pragma unroll
            for (int c = 0; c < NCT; c++) {
                const int e = c * 16 + m;          // column of xx
                float xv = 0.0f;
                if (v && e >= 1 && e <= D) xv = feats[t * ld + e - 1];
                if (v && e == 0) xv = 1.0f;
                x[j][c] = xv;
            }
        }
    };
    int o_a[GB], o_b[GB];
    float lp_a[GB], lp_b[GB], x_a[GB][NCT], x_b[GB][NCT];
    load_ord(0, o_a);
    load_data(o_a, lp_a, x_a);
    load_ord(1, o_b);
    auto compute = [&](const float (&lp)[GB], const float (&x)[GB][NCT]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < GB; j++) {
            // zeta = weight * prob (src/hmm.cpp:282,307), shifted; a frame past the item's end has logp = -inf -> 0
            const double z = wm * srx_exp_f64((double)lp[j] - sm);
#pragma unroll
            for (int c = 0; c < NCT; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(z, (double)x[j][c], acc[c], 0, 0, 0);
        }
    };
    for (int64_t b = 0; b < nb; b += 2) {
        // block b: data in (lp_a, x_a); order of block b+1 in o_b
        if (b + 1 < nb) {
            load_data(o_b, lp_b, x_b);
            load_ord(b + 2, o_a);
        }
        compute(lp_a, x_a);
        if (b + 1 < nb) {
            if (b + 2 < nb) {
                load_data(o_a, lp_a, x_a);
                load_ord(b + 3, o_b);
            }
            compute(lp_b, x_b);
        }
    }
    // C/D of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int c = 0; c < NCT; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = k + 4 * r, e = c * 16 + m;
            if (row < M && e < stride) partial[(size_t)item * R + row * stride + e] = acc[c][r];
        }
}

// ---- the fused statistics kernel (round 3) --------------------------------------------------------------------------
// Everything hmm::train takes from an aligned frame (src/hmm.cpp:274-339: zeta, zeta * x, zeta * (x - mean)^2) in ONE pass on the
// f64 matrix cores, with the variance pass folded into the first: with x' = x - c (c = the state's centre, below)
//     S0 = sum zeta,   S1 = sum zeta x',   S2 = sum zeta x'^2      ->  mean = c + S1/S0,   var = S2/S0 - (S1/S0)^2
// which is sum zeta (x - mean)^2 / S0 for mean = the NEW mean, as src/hmm.cpp:318-334 has it -- without the reference's rounding
// of (x - mean) to f32 (src/hmm.cpp:331), which moves a variance by parts in 1e8, and without a second sweep over the frames or
// a third exchange between ranks.  Centring on c (the average of the state's component means under A: the same numbers on every
// rank) keeps the subtraction S2/S0 - (S1/S0)^2 away from cancellation: |mean - c| is a few standard deviations, not |mean|.
//
// One wave per work item (<= CH frames of one state), 16 frames per step.  xx[f] = [x'_0 .. x'_{D-1}, 1, 0.. | x'^2_0 .. x'^2_{D-1}, 0..]
// (two halves of H = 8 NCT columns).  LOGP_MFMA: the log-densities themselves come off the matrix cores,
//     logp_m(x) - peak_m = sum_d ( -1/2 rv_d x'^2_d + rv_d mu'_d x'_d - 1/2 rv_d mu'^2_d ),   mu' = mean_m - c,  rv = 1 / var,
// as L^T [16 frames x 16 components] = xx [16 x 2H] . W^T [2H x 16] (4 NCT v_mfma_f64_16x16x4_f64), where peak_m = logp_m(mean_m) is
// the density's maximum: zeta' = w exp(logp - peak) <= w needs no data-dependent shift (no pre-pass over the frames, no MAX
// exchange) and the peak is re-applied where the mixture weights are formed.  The D lane of that product holds L for frames
// kq + 4r (kq = lane >> 4) of component lane & 15 -- exactly the A operand of the accumulation
//     stats[m][col] += sum_f zeta[m][f] xx[f][col]         (4 NCT more MFMAs: K = the 16 frames, four at a time, NCT column groups)
// so zeta never moves between lanes.  The frame rows are read once from HBM (each lane 2 NCT values of frame lane & 15, which are
// its A operand of the first product), centred, and pass through a 16 x H f64 tile in LDS to reach the lanes that need them as
// B operand of the second.  !LOGP_MFMA: zeta from log-densities given (srx_em_shift_dev: the rounding-faithful ones), only
// the accumulation runs here.  Column group c of the accumulation holds x' of dimensions 8c .. 8c+7 in lanes 0-7 and their
// squares in lanes 8-15.  Row layout of the result: [0, D) S1, D: S0, H + [0, D): S2, H + D: the number of frames (for the
// underflow test of k_em_fused_finish).
static constexpr int FUSED_NCT_MAX = 5;   // D + 1 <= 8 * NCT: D <= 39

#ifndef SRX_FUSED_WPE
#define SRX_FUSED_WPE 2   // waves per SIMD the kernel is compiled for
#endif
template <int NCT, bool LOGP_MFMA>
__global__ __launch_bounds__(64, SRX_FUSED_WPE) void k_em_fused(const float *__restrict__ feats, int ld, const int32_t *__restrict__ order,
                                                    const int64_t *__restrict__ start, const int32_t *__restrict__ item_start, int G, int M,
                                                    int D, const double *__restrict__ meanA, const double *__restrict__ rvarA,
                                                    const float *__restrict__ weight, const float *__restrict__ logp_in,
                                                    const float *__restrict__ shift_in, double *__restrict__ partial) {
    constexpr int H = 8 * NCT, K = 2 * H, NV = 2 * NCT;   // NV: values of a frame row per lane: columns NV kq + kk, a contiguous slice
    // (the k index of an MFMA may be permuted freely as long as both operands agree: k-chunk kk, slot kq <-> column NV kq + kk)
    __shared__ double xs[16][K + 1];   // a tile's centred rows: [x - c | 1 | 0.. ] in columns 0 .. H-1, their squares in H .. 2H-1
    __shared__ double cs[H];
    const int item = blockIdx.x;
    if (item >= item_start[G]) return;
    int lo = 0, hi = G;   // item_start[lo] <= item < item_start[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (item_start[mid] <= item) lo = mid; else hi = mid;
    }
    const int g = lo;
    const int64_t p0 = start[g] + (int64_t)(item - item_start[g]) * CH;
    const int64_t p1 = p0 + CH < start[g + 1] ? p0 + CH : start[g + 1];
    const int nfr = (int)(p1 - p0);
    const int lane = threadIdx.x, m = lane & 15, kq = lane >> 4, fl = lane & 15;
    // the state's centre: mean over its components of their means under A, in component order (every rank: same bits)
    if (lane < H) {
        double c = 0.0;
        if (lane < D) {
            for (int mm = 0; mm < M; mm++) c += meanA[((size_t)g * M + mm) * D + lane];
            c = c / (double)M;
        }
        cs[lane] = c;
    }
    __syncthreads();
    double cen[NV];   // centre of the columns this lane loads
#pragma unroll
    for (int kk = 0; kk < NV; kk++) cen[kk] = cs[NV * kq + kk];
    const double wm = m < M ? (double)weight[(size_t)g * M + m] : 0.0;
    const double sm = (!LOGP_MFMA && m < M) ? (double)shift_in[(size_t)g * M + m] : 0.0;
    // W^T as B operand of the first product: lane (k = kq, column m) holds W[m][NV kq + kk] for the 2 NV k-chunks
    double wreg[LOGP_MFMA ? 2 * NV : 1];
    if (LOGP_MFMA) {
        double k0 = 0.0;
        if (m < M)
            for (int d = 0; d < D; d++) {
                const double mu = meanA[((size_t)g * M + m) * D + d] - cs[d];
                k0 = fma(rvarA[((size_t)g * M + m) * D + d] * mu, mu, k0);
            }
#pragma unroll
        for (int kk = 0; kk < NV; kk++) {
            const int e = NV * kq + kk;
            double lin = 0.0, quad = 0.0;
            if (m < M && e < D) {
                const double rv = rvarA[((size_t)g * M + m) * D + e];
                lin = rv * (meanA[((size_t)g * M + m) * D + e] - cs[e]);
                quad = -0.5 * rv;
            }
            if (m < M && e == D) lin = -0.5 * k0;
            wreg[kk] = lin;
            wreg[NV + kk] = quad;
        }
    }
    f64x4 acc[NCT];
    const int bcol = fl < 8 ? fl : H + (fl & 7);
#pragma unroll
    for (int c = 0; c < NCT; c++) acc[c] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const int ntiles = (nfr + 15) >> 4;
    // frame row of tile t for this lane: frame 16 t + fl, columns NV kq .. NV kq + NV - 1: 128-bit loads (4-byte aligned
    // addresses are enough for them), four lanes cover a row's 156 bytes back to back
    auto load_ord = [&](int t) -> int {
        const int f = 16 * t + fl;
        return f < nfr ? order[p0 + f] : -1;
    };
    const int e0 = NV * kq;
    const int nvalid = D - e0 < NV ? (D - e0 < 0 ? 0 : D - e0) : NV;   // columns of this lane's slice that exist in a row
    auto load_row = [&](int o, float (&x)[NV]) __attribute__((always_inline)) {
#ifdef SRX_FUSED_EXP_NOGATHER
        const float *row = feats + (int64_t)(o >= 0 ? (o & 1023) : 0) * ld + e0;   // L2-hot rows
#else
        const float *row = feats + (int64_t)(o >= 0 ? o : 0) * ld + e0;
#endif
#pragma unroll
        for (int q4 = 0; q4 < NV; q4 += 4) {
            if (q4 + 4 <= NV && nvalid >= q4 + 4) {
                const f4u v = *reinterpret_cast<const f4u *>(row + q4);
                x[q4] = v.x; x[q4 + 1] = v.y; x[q4 + 2] = v.z; x[q4 + 3] = v.w;
            } else {
#pragma unroll
                for (int kk = q4; kk < (q4 + 4 < NV ? q4 + 4 : NV); kk++) x[kk] = kk < nvalid ? row[kk] : 0.0f;
            }
        }
        if (o < 0) {
#pragma unroll
            for (int kk = 0; kk < NV; kk++) x[kk] = 0.0f;
        }
    };
    // rows travel two tiles ahead of the arithmetic, frame indices three
    int ord_b = ntiles > 1 ? load_ord(1) : -1, ord_c = ntiles > 2 ? load_ord(2) : -1;
    int ord_cur = load_ord(0);
    float xr[NV], xr2[NV];
    load_row(ord_cur, xr);
    load_row(ord_b, xr2);
    for (int t = 0; t < ntiles; t++) {
        // ---- this lane's slice of the tile: centred values (= its A operand of the first product), also into LDS ----
        double v1[NV];
#pragma unroll
        for (int kk = 0; kk < NV; kk++) {
            const int e = NV * kq + kk;
            double v = (double)xr[kk] - cen[kk];
            if (e == D) v = 1.0;
            if (e > D) v = 0.0;
            v1[kk] = v;
            xs[fl][e] = v;
            xs[fl][H + e] = v * v;
        }
        float lpin[4];
        if (!LOGP_MFMA) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = __shfl(ord_cur, 4 * r + kq, 64);
                lpin[r] = (o >= 0 && m < M) ? logp_in[(int64_t)o * M + m] : -__builtin_huge_valf();
            }
        }
        __syncthreads();
        // ---- the loads of tile t + 2 fly under this tile's and the next tile's arithmetic ----
#pragma unroll
        for (int kk = 0; kk < NV; kk++) xr[kk] = xr2[kk];
        if (!LOGP_MFMA) ord_cur = ord_b;
        ord_b = ord_c;
        load_row(ord_b, xr2);          // (-1 past the item's end: zeros)
        ord_c = t + 3 < ntiles ? load_ord(t + 3) : -1;
        // ---- zeta for frames kq + 4r, component m ----
        double z[4];
        if (LOGP_MFMA) {
            // two independent accumulation chains (linear and quadratic columns) so that consecutive MFMAs do not wait on each other
            f64x4 L = (f64x4){0.0, 0.0, 0.0, 0.0}, Q = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
#ifdef SRX_FUSED_EXP_NOLOGP
            for (int kk = 0; kk < NV; kk++) { L[kk & 3] += v1[kk] * wreg[kk]; Q[kk & 3] += v1[kk] * wreg[NV + kk]; }
#else
            for (int kk = 0; kk < NV; kk++) {
                L = __builtin_amdgcn_mfma_f64_16x16x4f64(v1[kk], wreg[kk], L, 0, 0, 0);
                Q = __builtin_amdgcn_mfma_f64_16x16x4f64(v1[kk] * v1[kk], wreg[NV + kk], Q, 0, 0, 0);
            }
#endif
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool valid = 16 * t + 4 * r + kq < nfr;
                // zeta = weight * prob (src/hmm.cpp:282,307,328) relative to the component's peak density
#ifdef SRX_FUSED_EXP_NOEXP   // diagnostic builds (tools/em_variant.py): wrong results, timing only
                z[r] = valid ? wm * (L[r] + Q[r]) : 0.0;
#else
                z[r] = valid ? wm * srx_exp_f64(L[r] + Q[r]) : 0.0;
#endif
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) z[r] = wm * srx_exp_f64((double)lpin[r] - sm);   // -inf past the item's end: 0
        }
        // ---- accumulate: A = zeta (component m, frame kq + 4r), B = xx[frame kq + 4r][column group c] ----
#pragma unroll
        for (int c = 0; c < NCT; c++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const double b = xs[4 * r + kq][bcol + 8 * c];      // lanes 0-7 of a row: the linear columns; 8-15: the squares
#ifdef SRX_FUSED_EXP_NOACC
                acc[c][r] += z[r] * b;
#else
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(z[r], b, acc[c], 0, 0, 0);
#endif
            }
        }
        __syncthreads();   // the tile is consumed before the next one is written
    }
    // C/D of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int c = 0; c < NCT; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = kq + 4 * r;
            const int dim = 8 * c + (fl & 7);
            const int col = fl < 8 ? dim : H + dim;
            double v = acc[c][r];
            if (fl >= 8 && dim == D) v = (double)nfr;   // the spare column: frames of this item
            if (row < M) partial[((size_t)item * M + row) * K + col] = v;   // columns past D are sums of zeros
        }
}

// peak[o] = logp_o(mean_o) = logwt_o - 1/2 sum_d (log(2/pi) + log var_od): what the matrix-core log-densities are relative to
__global__ __launch_bounds__(256) void k_em_peak(int GM, int D, const double *__restrict__ lconst, const float *__restrict__ lw,
                                                 double *__restrict__ peak) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= GM) return;
    double sl = 0.0;
    for (int d = 0; d < D; d++) sl += lconst[(size_t)o * D + d];
    peak[o] = (double)lw[o] - 0.5 * sl;
}

// After the exchange: B from the fused statistics, then gaussian::blend(A, keep) (src/hmm.cpp:288-296, 317, 339-341; src/gmm.cpp:82-88).
// shift == nullptr: the statistics were taken relative to each component's peak density, recomputed here in f64.  A component
// that saw frames but whose zeta sum is below 1e-290 has lost them to f64 underflow (every frame tens of standard deviations
// away): counted in *underflow, and the caller repeats the iteration's statistics through the data-shifted passes.
__global__ __launch_bounds__(256) void k_em_fused_finish(int G, int M, int D, int H, const double *__restrict__ st,
                                                         const float *__restrict__ shift, const double *__restrict__ peakA, const double *__restrict__ meanA,
                                                         const double *__restrict__ varA, const float *__restrict__ wA,
                                                         const float *__restrict__ lwA, double *__restrict__ meanB,
                                                         double *__restrict__ varB, float *__restrict__ wB, float *__restrict__ lwB, float keep,
                                                         int32_t *__restrict__ underflow) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)G * M * D) return;
    const int K = 2 * H;
    const int e = (int)(idx % D);
    const size_t o = (size_t)(idx / D);
    const int g = (int)(o / M);
    double c = 0.0;
    for (int mm = 0; mm < M; mm++) c += meanA[((size_t)g * M + mm) * D + e];
    c = c / (double)M;
    const double S0 = st[o * K + D], S1 = st[o * K + e], S2 = st[o * K + H + e];
    const double mu = S1 / S0;
    double vnew = S2 / S0 - mu * mu;
    if (vnew < 0.0) vnew = 0.0;   // a component held by one frame: the reference's sum of squared differences is exactly 0
    const float omk = 1 - keep;
    meanB[idx] = (c + mu) * (double)keep + meanA[idx] * (double)omk;
    varB[idx] = vnew * (double)keep + varA[idx] * (double)omk;
    if (e == 0) {
        auto peak = [&](int mm) -> double {
            const size_t om = (size_t)g * M + mm;
            return shift ? (double)shift[om] : peakA[om];
        };
        double pmax = -__builtin_huge_val();
        for (int mm = 0; mm < M; mm++) {
            const double p = peak(mm);
            if (p > pmax) pmax = p;
        }
        double zg = 0.0, zo = 0.0;
        for (int mm = 0; mm < M; mm++) {
            const size_t om = (size_t)g * M + mm;
            const double p = peak(mm);
            const double z = (p == -__builtin_huge_val()) ? 0.0 : st[om * K + D] * exp(p - pmax);
            zg += z;
            if (om == o) zo = z;
        }
        const float w = (float)(zo / zg);
        const float a = w * keep;
        const float b = wA[o] * omk;
        const float wn = a + b;
        wB[o] = wn;
        lwB[o] = logf(wn);
        if (underflow && st[o * K + H + D] > 0.0 && S0 < 1e-290) atomicAdd(underflow, 1);
    }
}

// Log-densities of the aligned frames under A in sorted order: a work item's state parameters sit in LDS and the item's
// frames stream past them (the frame-order kernel k_em_logp gathers 780 bytes of parameters per (frame, component)
// through L2).  Same arithmetic as srx_logprob_exact_lc, statement for statement.  A thread owns one component and
// LP_FPT frames of each 64-frame step, so every parameter triple read from LDS serves LP_FPT frames.
// shift[o] = max over the frames of component o.
static constexpr int LP_FPT = 4;           // frames per thread and step
static constexpr int LP_STEP = 16 * LP_FPT;  // frames per step
__global__ __launch_bounds__(256) void k_em_logp_seg(const float *__restrict__ feats, int ld, const int32_t *__restrict__ order,
                                                     const int64_t *__restrict__ start, const int32_t *__restrict__ item_start, int G,
                                                     int M, int D, const double *__restrict__ mean, const double *__restrict__ rvar,
                                                     const double *__restrict__ lconst, const float *__restrict__ logwt,
                                                     float *__restrict__ logp, float *__restrict__ shift) {
    extern __shared__ double par[];          // mean[M][D], rvar[M][D], lconst[M][D]
    __shared__ double xs[LP_STEP][65];       // one step's feature rows, already widened (D <= 64)
    __shared__ int fts[2][LP_STEP];
    __shared__ float smax[16];
    const int item = blockIdx.x;
    if (item >= item_start[G]) return;
    int lo = 0, hi = G;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (item_start[mid] <= item) lo = mid; else hi = mid;
    }
    const int g = lo;
    const int64_t p0 = start[g] + (int64_t)(item - item_start[g]) * CH;
    const int64_t p1 = p0 + CH < start[g + 1] ? p0 + CH : start[g + 1];
    const int MD = M * D;
    double *pm = par, *pr = par + MD, *pl = par + 2 * MD;
    for (int i = threadIdx.x; i < MD; i += 256) {
        pm[i] = mean[(size_t)g * MD + i];
        pr[i] = rvar[(size_t)g * MD + i];
        pl[i] = lconst[(size_t)g * MD + i];
    }
    if (threadIdx.x < 16) smax[threadIdx.x] = -__builtin_huge_valf();
    // thread = (frame slot f = tid / 16, component m = tid % 16); its frames in a step are f, f + 16, f + 32, f + 48
    const int f = threadIdx.x >> 4, m = threadIdx.x & 15;
    const float lw = m < M ? logwt[(size_t)g * M + m] : 0.0f;
    float mymax = -__builtin_huge_valf();
    // The feature rows of a step are gathered one step ahead into registers and the frame indices one step further
    // (as in k_em_accum_seg): no barrier waits on a gather.
    constexpr int NXR = (LP_STEP * 64 + 255) / 256;   // LP_STEP * D / 256 with D <= 64
    float xr[NXR];
    auto gather = [&](int buf, int nf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NXR; k++) {
            const int i = threadIdx.x + 256 * k;
            const int ff = i / D, d = i - ff * D;
            xr[k] = (i < LP_STEP * D && ff < nf) ? feats[(int64_t)fts[buf][ff] * ld + d] : 0.0f;
        }
    };
    const int64_t nsteps = (p1 - p0 + LP_STEP - 1) / LP_STEP;
    auto frames_of = [&](int64_t s_) { const int64_t r = p1 - (p0 + s_ * LP_STEP); return r < LP_STEP ? (int)(r < 0 ? 0 : r) : LP_STEP; };
    int ord = 0;
    if (threadIdx.x < frames_of(0)) fts[0][threadIdx.x] = order[p0 + threadIdx.x];
    if (threadIdx.x < frames_of(1)) ord = order[p0 + LP_STEP + threadIdx.x];
    __syncthreads();
    gather(0, frames_of(0));
    for (int64_t st = 0; st < nsteps; st++) {
        const int nf = frames_of(st);
        const int cur = (int)(st & 1);
        __syncthreads();   // the previous step's reads of xs are done
#pragma unroll
        for (int k = 0; k < NXR; k++) {
            const int i = threadIdx.x + 256 * k;
            const int ff = i / D, d = i - ff * D;
            if (i < LP_STEP * D) xs[ff][d] = (double)xr[k];
        }
        if (threadIdx.x < frames_of(st + 1)) fts[cur ^ 1][threadIdx.x] = ord;
        __syncthreads();
        if (st + 1 < nsteps) {
            gather(cur ^ 1, frames_of(st + 1));
            if (threadIdx.x < frames_of(st + 2)) ord = order[p0 + (st + 2) * LP_STEP + threadIdx.x];
        }
        if (m < M) {
            float res[LP_FPT];
#pragma unroll
            for (int k = 0; k < LP_FPT; k++) res[k] = 0.0f;
            const double *qm = pm + m * D, *qr = pr + m * D, *ql = pl + m * D;
            for (int i = 0; i < D; i++) {
                const double mu = qm[i], rv = qr[i], lc = ql[i];
#pragma unroll
                for (int k = 0; k < LP_FPT; k++) {
                    const float t = (float)(xs[f + 16 * k][i] - mu);
                    const float tt = t * t;
                    const double term = fma((double)tt, rv, lc);   // one rounding fewer than the reference's two
                    res[k] = (float)((double)res[k] + term);
                }
            }
#pragma unroll
            for (int k = 0; k < LP_FPT; k++) {
                if (f + 16 * k < nf) {
                    const float h = -0.5f * res[k];
                    const float lp = h + lw;
                    logp[(int64_t)fts[cur][f + 16 * k] * M + m] = lp;
                    if (lp == lp) mymax = fmaxf(mymax, lp);
                }
            }
        }
    }
    // per-component maximum of this item -> shift (float max through the ordered-int trick of k_em_logp)
    __syncthreads();
    if (m < M && mymax > -__builtin_huge_valf()) {
        if (mymax >= 0.0f)
            atomicMax(reinterpret_cast<int *>(&smax[m]), __float_as_int(mymax));
        else
            atomicMin(reinterpret_cast<unsigned int *>(&smax[m]), __float_as_uint(mymax));
    }
    __syncthreads();
    if (threadIdx.x < M && smax[threadIdx.x] > -__builtin_huge_valf()) {
        const float v = smax[threadIdx.x];
        float *addr = &shift[(size_t)g * M + threadIdx.x];
        if (v >= 0.0f)
            atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
        else
            atomicMin(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
    }
}

// frames that were skipped (gid < 0) get logp = 0 like k_em_logp
__global__ __launch_bounds__(256) void k_em_logp_skipped(const int32_t *__restrict__ order, const int64_t *__restrict__ start, int G, int64_t T,
                                                         int M, float *__restrict__ logp) {
    const int64_t i = start[G] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const int64_t t = order[i];
    for (int m = 0; m < M; m++) logp[t * M + m] = 0.0f;
}

int srx_em_logp_seg(srx_ctx *ctx, const float *feats, int ld, int64_t T, int G, int M, int D, const double *mean, const double *rvar,
                    const double *lconst, const float *logwt, float *logp, float *shift) {
    EmSortView v;
    em_sort_view(ctx->ws[2], T, G, &v);
    const int64_t max_items = T / CH + G + 1;
    {
        SrxProf prof_(ctx, "k_em_logp_seg");
        hipLaunchKernelGGL(k_em_logp_seg, dim3((unsigned)max_items), dim3(256), (size_t)3 * M * D * sizeof(double), ctx->stream, feats, ld, v.order,
                           v.start, v.item_start, G, M, D, mean, rvar, lconst, logwt, logp, shift);
    }
    SRX_LAUNCH_CHECK(ctx);
    {
        SrxProf prof_(ctx, "k_em_logp_skipped");
        hipLaunchKernelGGL(k_em_logp_skipped, dim3((unsigned)ceil_div64(T, 256)), dim3(256), 0, ctx->stream, v.order, v.start, G, T, M, logp);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

// stats[g][.] = sum over the state's items, in chunk order
__global__ __launch_bounds__(256) void k_em_seg_reduce(const int32_t *__restrict__ item_start, int G, int R, const double *__restrict__ partial,
                                                       double *__restrict__ stats) {
    const int g = blockIdx.x;
    const int i0 = item_start[g], i1 = item_start[g + 1];
    for (int idx = threadIdx.x; idx < R; idx += 256) {
        double s = 0.0;
        int it = i0;
        for (; it + 8 <= i1; it += 8) {   // eight loads in flight, summed in chunk order
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = partial[(size_t)(it + k) * R + idx];
#pragma unroll
            for (int k = 0; k < 8; k++) s += v[k];
        }
        for (; it < i1; it++) s += partial[(size_t)it * R + idx];
        stats[(size_t)g * R + idx] = s;
    }
}

// 16 components x 64 row elements: 256 threads x at most 4 sums; the counting sort keeps G + 1 counters in LDS
bool srx_em_seg_applicable(int G, int M, int D) { return M <= 16 && D <= 63 && G + 1 <= 16384; }

// Sorts the frames of alignment `gid` by state into workspace slot 2 (kept until the next sort).
int srx_em_seg_sort(srx_ctx *ctx, const int32_t *gid, int64_t T, int G) {
    SRX_REQUIRE(ctx, T < (1ll << 31), "srx_em_*: too many frames for one shard (%lld)", (long long)T);
    void *ws = nullptr;
    int rc = srx_workspace(ctx, 2, em_sort_bytes(T, G), &ws);
    if (rc != SRX_OK) return rc;
    EmSortView v;
    em_sort_view(ws, T, G, &v);
    ctx->em_gid = nullptr;
    const int Gp = G + 1;
    const int64_t nch = cs_chunks(T);
    SRX_REQUIRE(ctx, nch < (1ll << 31), "srx_em_*: too many frames");
    const size_t lds = (size_t)Gp * sizeof(int32_t);
    {
        SrxProf prof_(ctx, "k_em_csort_hist");
        hipLaunchKernelGGL(k_em_csort_hist, dim3((unsigned)nch), dim3(64), lds, ctx->stream, gid, T, G, v.hist);
    }
    SRX_LAUNCH_CHECK(ctx);
    {
        SrxProf prof_(ctx, "k_em_csort_scan");
        const int64_t ngr = (nch + CS_GROUP - 1) / CS_GROUP;
        hipLaunchKernelGGL(k_em_csort_scan_a, dim3((Gp + 255) / 256, (unsigned)ngr), dim3(256), 0, ctx->stream, v.hist, nch, Gp, v.gsum);
        hipLaunchKernelGGL(k_em_csort_scan_b, dim3((Gp + 255) / 256), dim3(256), 0, ctx->stream, v.gsum, ngr, Gp, v.total);
        hipLaunchKernelGGL(k_em_csort_starts, dim3(1), dim3(1024), 0, ctx->stream, v.total, Gp, v.start);
    }
    SRX_LAUNCH_CHECK(ctx);
    {
        SrxProf prof_(ctx, "k_em_csort_scatter");
        hipLaunchKernelGGL(k_em_csort_scatter, dim3((unsigned)nch), dim3(64), lds, ctx->stream, gid, T, G, v.hist, v.gsum, v.start, v.order);
    }
    SRX_LAUNCH_CHECK(ctx);
    {
        SrxProf prof_(ctx, "k_em_item_scan");
        hipLaunchKernelGGL(k_em_item_scan, dim3(1), dim3(1024), 0, ctx->stream, v.start, G, v.item_start);
    }
    SRX_LAUNCH_CHECK(ctx);
    ctx->em_gid = gid;
    ctx->em_T = T;
    ctx->em_G = G;
    return SRX_OK;
}

template <int PASS>
int srx_em_accum_seg(srx_ctx *ctx, const float *feats, int ld, const int32_t *gid, int64_t T, int G, int M, int D, const float *weight,
                     const float *logp, const float *shift, const double *meanB, double *stats) {
    if (ctx->em_gid != gid || ctx->em_T != T || ctx->em_G != G) {   // pass called without the shift step of this alignment
        int rc = srx_em_seg_sort(ctx, gid, T, G);
        if (rc != SRX_OK) return rc;
    }
    EmSortView v;
    em_sort_view(ctx->ws[2], T, G, &v);
    const int R = M * (PASS == 1 ? 1 + D : D);
    const int64_t max_items = T / CH + G + 1;
    void *pw = nullptr;
    int rc = srx_workspace(ctx, 3, (size_t)max_items * R * sizeof(double), &pw);
    if (rc != SRX_OK) return rc;
    if (PASS == 1) {
        SrxProf prof_(ctx, "k_em_accum_seg1");
        const int nct = (1 + D + 15) / 16;
        auto kern = nct <= 1 ? k_em_accum1_mfma<1> : nct == 2 ? k_em_accum1_mfma<2> : nct == 3 ? k_em_accum1_mfma<3> : k_em_accum1_mfma<4>;
        hipLaunchKernelGGL(kern, dim3((unsigned)max_items), dim3(64), 0, ctx->stream, feats, ld, v.order, v.start, v.item_start, G, M, D,
                           weight, logp, shift, (double *)pw);
    } else {
        SrxProf prof_(ctx, "k_em_accum_seg2");
        const int nacc = (D + 15) / 16;
        auto kern = nacc <= 1 ? k_em_accum_seg2<1> : nacc == 2 ? k_em_accum_seg2<2> : nacc == 3 ? k_em_accum_seg2<3> : k_em_accum_seg2<4>;
        hipLaunchKernelGGL(kern, dim3((unsigned)max_items), dim3(256), 0, ctx->stream, feats, ld, v.order, v.start,
                           v.item_start, G, M, D, weight, logp, shift, meanB, (double *)pw);
    }
    SRX_LAUNCH_CHECK(ctx);
    {
        SrxProf prof_(ctx, "k_em_seg_reduce");
        hipLaunchKernelGGL(k_em_seg_reduce, dim3(G), dim3(256), 0, ctx->stream, v.item_start, G, R, (const double *)pw, stats);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

bool srx_em_fused_applicable(int G, int M, int D) { return M <= 16 && D + 1 <= 8 * FUSED_NCT_MAX && G + 1 <= 16384; }
int srx_em_fused_row(int D) { return 16 * ((D + 1 + 7) / 8); }   // doubles per (state, component) row of the fused statistics

// logp == nullptr: log-densities on the matrix cores (peak-relative); otherwise the given ones with their shift.
int srx_em_fused_launch(srx_ctx *ctx, const float *feats, int ld, const int32_t *gid, int64_t T, int G, int M, int D, const double *meanA,
                        const double *rvarA, const float *weight, const float *logp, const float *shift, double *stats) {
    const int K = srx_em_fused_row(D);
    const int R = M * K;
    SRX_HIP(ctx, hipMemsetAsync(stats, 0, (size_t)G * R * sizeof(double), ctx->stream));
    if (T == 0) return SRX_OK;
    // a new alignment under the same pointer is the normal case: only the passes that follow srx_em_shift_dev may reuse its sort
    if (!logp || ctx->em_gid != gid || ctx->em_T != T || ctx->em_G != G) {
        int rc = srx_em_seg_sort(ctx, gid, T, G);
        if (rc != SRX_OK) return rc;
    }
    EmSortView v;
    em_sort_view(ctx->ws[2], T, G, &v);
    const int64_t max_items = T / CH + G + 1;
    void *pw = nullptr;
    int rc = srx_workspace(ctx, 3, (size_t)max_items * R * sizeof(double), &pw);
    if (rc != SRX_OK) return rc;
    const int nct = K / 16;
    {
        SrxProf prof_(ctx, "k_em_fused");
#define SRX_FUSED(N)                                                                                                                  \
    do {                                                                                                                              \
        if (logp)                                                                                                                     \
            hipLaunchKernelGGL((k_em_fused<N, false>), dim3((unsigned)max_items), dim3(64), 0, ctx->stream, feats, ld, v.order, v.start, \
                               v.item_start, G, M, D, meanA, rvarA, weight, logp, shift, (double *)pw);                               \
        else                                                                                                                          \
            hipLaunchKernelGGL((k_em_fused<N, true>), dim3((unsigned)max_items), dim3(64), 0, ctx->stream, feats, ld, v.order, v.start,  \
                               v.item_start, G, M, D, meanA, rvarA, weight, logp, shift, (double *)pw);                               \
    } while (0)
        switch (nct) {
            case 1: SRX_FUSED(1); break;
            case 2: SRX_FUSED(2); break;
            case 3: SRX_FUSED(3); break;
            case 4: SRX_FUSED(4); break;
            default: SRX_FUSED(5); break;
        }
#undef SRX_FUSED
    }
    SRX_LAUNCH_CHECK(ctx);
    {
        SrxProf prof_(ctx, "k_em_seg_reduce");
        hipLaunchKernelGGL(k_em_seg_reduce, dim3(G), dim3(256), 0, ctx->stream, v.item_start, G, R, (const double *)pw, stats);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

int srx_em_fused_finish_launch(srx_ctx *ctx, int G, int M, int D, const double *stats, const float *shift, const double *lconstA,
                               double *peak_scratch, const double *meanA, const double *varA, const float *wA, const float *lwA, double *meanB,
                               double *varB, float *wB, float *lwB, float keep, int32_t *underflow) {
    const int64_t n = (int64_t)G * M * D;
    if (underflow) SRX_HIP(ctx, hipMemsetAsync(underflow, 0, sizeof(int32_t), ctx->stream));
    if (!shift) {
        SrxProf prof_(ctx, "k_em_peak");
        hipLaunchKernelGGL(k_em_peak, dim3((G * M + 255) / 256), dim3(256), 0, ctx->stream, G * M, D, lconstA, lwA, peak_scratch);
    }
    SRX_LAUNCH_CHECK(ctx);
    {
        SrxProf prof_(ctx, "k_em_fused_finish");
        hipLaunchKernelGGL(k_em_fused_finish, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, G, M, D, srx_em_fused_row(D) / 2, stats,
                           shift, peak_scratch, meanA, varA, wA, lwA, meanB, varB, wB, lwB, keep, underflow);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

template int srx_em_accum_seg<1>(srx_ctx *, const float *, int, const int32_t *, int64_t, int, int, int, const float *, const float *,
                                 const float *, const double *, double *);
template int srx_em_accum_seg<2>(srx_ctx *, const float *, int, const int32_t *, int64_t, int, int, int, const float *, const float *,
                                 const float *, const double *, double *);
