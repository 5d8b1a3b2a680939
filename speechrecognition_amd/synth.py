"""Synthetic Voxforge-shaped inputs.  Host-side data generation only."""
import numpy as np

N_PHONES = 40  # src/pronounce.hpp:24-29, SIL = 0


def utterance_audio(seed, n_samples, sps=16000):
    """Three-formant harmonic source + noise, float32 in [-1, 1]."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / sps
    f0 = rng.uniform(90, 220)
    x = np.zeros(n_samples)
    for fmt, bw in ((rng.uniform(300, 900), 1.0), (rng.uniform(1000, 2300), 0.6), (rng.uniform(2400, 3400), 0.3)):
        k = max(1, int(round(fmt / f0)))
        x += bw * np.sin(2 * np.pi * k * f0 * t + rng.uniform(0, 6.28))
    env = 0.55 + 0.45 * np.sin(2 * np.pi * rng.uniform(1.5, 4.0) * t)
    x = 0.25 * env * x + 0.05 * rng.standard_normal(n_samples)
    return np.clip(x, -1, 1).astype(np.float32)


def corpus_audio(n_utt, seed=0, min_s=2.0, max_s=8.0, sps=16000):
    """n_utt utterances of 2-8 s: returns (samples concatenated, sample_off int64[n_utt+1])."""
    rng = np.random.default_rng(seed)
    lens = (sps * rng.uniform(min_s, max_s, n_utt)).astype(np.int64)
    off = np.zeros(n_utt + 1, np.int64)
    off[1:] = np.cumsum(lens)
    x = np.empty(int(off[-1]), np.float32)
    for u in range(n_utt):
        x[off[u]:off[u + 1]] = utterance_audio(seed * 1000003 + u, int(lens[u]), sps)
    return x, off


def phone_string(rng, n_words=None, max_states=None):
    """SIL w1 SIL w2 ... SIL with 2-6 phones per word (pronouncer::pronounce, src/pronounce.cpp:260-278)."""
    n_words = n_words or int(rng.integers(4, 13))
    ph = [0]
    for _ in range(n_words):
        ph += list(rng.integers(1, N_PHONES, int(rng.integers(2, 7))))
        ph.append(0)
    if max_states and len(ph) > max_states:
        ph = ph[:max_states - 1] + [0]
    return np.array(ph, np.int32)


def random_model(G, M, D, seed=777):
    rng = np.random.default_rng(seed)
    mean = rng.standard_normal((G, M, D))
    var = rng.uniform(0.5, 1.5, (G, M, D))
    w = rng.uniform(0.5, 1.5, (G, M)).astype(np.float32)
    w = (w / w.sum(axis=1, keepdims=True)).astype(np.float32)
    return mean, var, w
