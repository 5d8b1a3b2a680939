"""The MFCC stages and the single Gaussian as stand-alone calls (csrc/srx_frameops.hip), each against the oracle's
restatement of the reference function it replaces:
  Clip::partial_dft / dft  src/clip.cpp:166-194     filter / filterbank::operator()  src/mfcc.cpp:143-167
  log(buf, len)  src/utils.hpp:55-60                dct  src/mfcc.cpp:205-214
  featurevec::compute_deltas / compute_delta2s, compute_deltas(vector)  src/mfcc.cpp:73-102
  gaussian::operator()  src/gmm.cpp:41-49
Bar: bit-exact (the kernels follow the reference's operation order with host-built f64 cosine tables), except log(x+1),
which is held to one unit in the last place of the f32 result (device logf against glibc logf), and the Gaussian, where
f64 stands in for the reference's 80-bit long double (<= 1 ulp, as SRX_SCORE_EXACT).  Parity unpinned: the oracle is
the builder's restatement."""
import numpy as np
import pytest

from speechrecognition_amd import host

pytestmark = pytest.mark.gpu


def _ulps(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


@pytest.mark.parametrize("N,len_", [(320, 144), (400, 180), (320, 320), (321, 200), (64, 33), (16, 3)])
def test_partial_dft_frames_bit_exact_all_windows_mirror_and_stack(ctx, oracle, N, len_):
    rng = np.random.default_rng(N * 1000 + len_)
    frames = (rng.standard_normal((5, N)) * 0.3).astype(np.float32)
    ham = np.array([oracle.hamming(n, N) for n in range(N // 2)], np.float32)
    for win in (None, ham):
        got = host.partial_dft_frames(ctx, frames, win, len_)
        for r in range(len(frames)):
            ref = oracle.partial_dft_win(frames[r], win, N, len_)
            assert got[r].tobytes() == ref.tobytes(), (N, len_, r, np.abs(got[r] - ref).max())
        # stack: added to what the caller's buffer holds, mirrored bins twice where k == N-k-1 (src/clip.cpp:186-188)
        acc = rng.standard_normal((len(frames), len_)).astype(np.float32)
        got = host.partial_dft_frames(ctx, frames, win, len_, stack=True, out=acc)
        for r in range(len(frames)):
            ref = oracle.partial_dft_win(frames[r], win, N, len_, stack=True, out=acc[r])
            assert got[r].tobytes() == ref.tobytes(), (N, len_, r, "stack")
    if len_ == 144:   # the Hamming case is what mfcc() runs: equal to the oracle's fixed-window form too
        got = host.partial_dft_frames(ctx, frames, ham, len_)
        assert got[0].tobytes() == oracle.partial_dft(frames[0], N, len_).tobytes()


def test_partial_dft_frames_refuses_more_bins_than_samples(ctx):
    from speechrecognition_amd._capi import SrxError
    with pytest.raises(SrxError):
        host.partial_dft_frames(ctx, np.zeros((1, 32), np.float32), None, 33)
    assert host.partial_dft_frames(ctx, np.zeros((0, 32), np.float32), None, 8).shape == (0, 8)


@pytest.mark.parametrize("bins,sps,end", [(320, 16000, 7200.0), (400, 16000, 7200.0), (768, 44100, 7200.0)])
def test_filterbank_log_dct_chain_bit_exact(ctx, oracle, bins, sps, end):
    fb, bin_max = oracle.mel_filterbank(end, 48, bins, sps)
    rng = np.random.default_rng(bins)
    spec = np.abs(rng.standard_normal((37, bin_max))).astype(np.float32) * np.float32(0.02)
    filt = host.filterbank_apply(ctx, fb, spec)
    for r in range(len(spec)):
        assert filt[r].tobytes() == oracle.filterbank_apply(fb, spec[r]).tobytes(), r
    # one filter alone = filter::operator()
    one = host.filterbank_apply(ctx, fb[7:8], spec[:3])
    assert one[:, 0].tobytes() == filt[:3, 7].tobytes()
    lf = host.log1p(ctx, filt)
    ref_lf = oracle.log1p(filt)
    assert _ulps(lf, ref_lf).max() <= 1
    # the DCT on the ORACLE's logs, so that its own bar is not blurred by the logf difference
    ceps = host.dct(ctx, ref_lf, 13)
    for r in range(len(spec)):
        assert ceps[r].tobytes() == oracle.dct(ref_lf[r], 13).tobytes(), r
    assert host.dct(ctx, ref_lf[:2], 1).shape == (2, 1)
    wide = host.dct(ctx, ref_lf[:2, :5], 9)   # more outputs than inputs
    assert wide[1].tobytes() == oracle.dct(ref_lf[1, :5], 9).tobytes()


def test_filterbank_refuses_a_filter_outside_the_spectrum(ctx, oracle):
    from speechrecognition_amd._capi import SrxError
    fb, bin_max = oracle.mel_filterbank()
    with pytest.raises(SrxError):
        host.filterbank_apply(ctx, fb, np.zeros((1, bin_max - 1), np.float32))


def test_stage_by_stage_equals_mfcc_frame(ctx, oracle):
    """partial_dft -> filterbank -> log -> dct, one call each, against the oracle's mfcc() (src/mfcc.cpp:219-232)."""
    fb, bin_max = oracle.mel_filterbank()
    rng = np.random.default_rng(9)
    x = (rng.standard_normal(4000) * 0.01).astype(np.float32)
    ref_ceps, _ = oracle.write_features(x, 320, 160, fb, 13)
    xp = oracle.preemphasis(x)
    n = ref_ceps.shape[0]
    frames = np.stack([xp[160 * f:160 * f + 320] for f in range(n)])
    ham = np.array([oracle.hamming(k, 320) for k in range(160)], np.float32)
    spec = host.partial_dft_frames(ctx, frames, ham, bin_max)
    lf = host.log1p(ctx, host.filterbank_apply(ctx, fb, spec))
    ceps = host.dct(ctx, lf, 13)
    # every stage but the log is bit-exact; one ulp of a log value (~1e-2) moves a cepstrum by at most 48 * 2^-30
    # (the log differs from glibc's logf in the last place for a few per cent of the values, and 48 of them enter every cepstrum)
    assert np.abs(ceps - ref_ceps).max() <= 48 * 2.0 ** -30


def test_delta_and_compute_deltas(ctx, oracle):
    rng = np.random.default_rng(4)
    c = rng.standard_normal((50, 13)).astype(np.float32)
    c[3, 2] = -0.0
    c[4, 2] = 0.0
    ref = oracle.compute_deltas(c)
    fo = np.array([0, len(c)], np.int64)
    got = host.read_features(ctx, c, fo, -1)          # WL = -1: compute_deltas(vector) alone
    assert got.tobytes() == ref.tobytes()
    for i, (j, k) in ((0, (0, 1)), (7, (6, 8)), (49, (48, 49))):
        d = host.delta(ctx, c[k], c[j], k - j)
        assert d.tobytes() == ref[i, 13:26].tobytes()
        dd = host.delta(ctx, ref[k, 13:26], ref[j, 13:26], k - j)
        assert dd.tobytes() == ref[i, 26:].tobytes()
    one = host.delta(ctx, c[0], c[0], 0)               # a one-frame utterance: 0 / 0 (src/mfcc.cpp:79 with k == j)
    assert np.isnan(one).all()


def test_single_gaussian_logprob(ctx, oracle):
    rng = np.random.default_rng(12)
    D = 13
    x = rng.standard_normal((64, 39)).astype(np.float32) * 2
    mean = rng.standard_normal(D)
    var = rng.uniform(0.2, 3.0, D)
    lw = np.float32(np.log(np.float32(0.37)))
    got = host.gaussian_logprob(ctx, x, mean, var, lw)
    ref = np.array([oracle.gaussian_logprob(x[t, :D], mean, var, lw) for t in range(len(x))], np.float32)
    assert _ulps(got, ref).max() <= 1
    assert (got == ref).mean() > 0.95
    assert host.gaussian_logprob(ctx, x[:0], mean, var, lw).shape == (0,)
