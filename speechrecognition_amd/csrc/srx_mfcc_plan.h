// MFCC plan shared by srx_mfcc.hip (f32 matrix-core path) and srx_mfcc_f16.hip (split-f16 path).
#pragma once
#include "srx_common.h"

struct srx_mfcc_plan {
    srx_ctx *ctx = nullptr;
    srx_mfcc_cfg cfg;
    int nfilt = 0, bin_max = 0;
    int half = 0;   // N/2 live samples per frame
    int KH = 0;     // half rounded up to a multiple of 4 (f32 MFMA k-step)
    int NB = 0;     // bins rounded up to a multiple of 64 (groups of 4 tiles)
    int NFP = 0;    // filters rounded up to 16
    int NCP = 16;   // cepstra rounded up to 16
    float rootn_inv = 0;
    float *win = nullptr;   // [KH]
    float *twid = nullptr;  // [KH][2][NB]   cos then sin, frequency index already mirrored
    float *fbm = nullptr;   // [NB][NFP]
    float *dctm = nullptr;  // [NFP][NCP]
    // split-f16 path (srx_mfcc_f16.hip); ks16 == 0: not available for this configuration
    int path = SRX_MFCC_AUTO;
    int ks16 = 0;           // 16-sample k-steps per half window
    int nbt16 = 0;          // 32-bin tiles
    void *pk16 = nullptr;     // [nbt16] chunks of MFMA fragments: twiddles then filterbank columns of the tile
    void *dct16 = nullptr;    // DCT fragments
    // LDS FFT path (srx_mfcc_fft.hip); fft_tw == nullptr: not available for this window
    void *fft_tw = nullptr;     // float2 twiddles: [R0*R1] stage 2, [H] stage 3, [H+1] unpacking of the real transform
    float *fft_fbw = nullptr;   // [fft_fbt] filter taps over each filter's own bins, filter after filter
    int *fft_fblo = nullptr;    // [nfilt] first bin, [nfilt] number of bins, [nfilt] offset of the taps
    int fft_fbt = 0;            // taps of all filters together
    int fft_r[3] = {0, 0, 0};
};

// srx_mfcc_f16.hip
int srx_mfcc16_supported(int half, int n_filters, int n_ceps, int bin_max);
int srx_mfcc16_build(srx_mfcc_plan *p, const std::vector<float> &win, const std::vector<float> &tw, const std::vector<float> &fbm,
                     const std::vector<float> &dm);
int srx_mfcc16_launch(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples_dev, const int64_t *sample_off_dev, int n_utt,
                      const int64_t *frame_off_dev, int64_t total_frames, const float *scale_dev, float *ceps_dev);

// srx_mfcc_fft.hip
bool srx_mfccfft_supported(int window_len, int n_filters, int n_ceps);
int srx_mfccfft_build(srx_mfcc_plan *p, const srx_filter *fb);
int srx_mfccfft_launch(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples_dev, const int64_t *sample_off_dev, int n_utt,
                       const int64_t *frame_off_dev, int64_t total_frames, const float *scale_dev, float *ceps_dev);
