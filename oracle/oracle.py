"""ctypes loader for the CPU oracle (oracle/srx_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under speechrecognition_amd/ may import this module.
Parity status: UNPINNED (see the header of srx_oracle.c and DESIGN.md).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsrx_oracle.so")

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
ldp = np.ctypeslib.ndpointer(np.longdouble, flags="C_CONTIGUOUS")

FILTER_DT = np.dtype([("center", np.float32), ("width", np.float32), ("height", np.float32)])


def build(force=False):
    src = os.path.join(_HERE, "srx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_mel_filterbank.restype = C.c_int
        L.orc_mel_filterbank.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_rms_amplitude.restype = C.c_double
        L.orc_rms_amplitude.argtypes = [f32p, C.c_long]
        L.orc_normalize_volume.argtypes = [f32p, C.c_long, C.c_float]
        L.orc_preemphasis.argtypes = [f32p, C.c_long, C.c_float]
        L.orc_hamming.restype = C.c_float
        L.orc_hamming.argtypes = [C.c_int, C.c_int]
        L.orc_num_frames.restype = C.c_long
        L.orc_num_frames.argtypes = [C.c_long, C.c_int, C.c_int]
        L.orc_partial_dft.argtypes = [f32p, C.c_int, C.c_int, f32p]
        L.orc_partial_dft_win.argtypes = [f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, f32p]
        L.orc_log1p.argtypes = [f32p, C.c_long]
        L.orc_compute_deltas.argtypes = [f32p, C.c_long, C.c_int]
        L.orc_filterbank_apply.argtypes = [C.c_void_p, C.c_int, f32p, f32p]
        L.orc_dct.argtypes = [f32p, C.c_int, C.c_int, f32p]
        L.orc_mfcc_frame.argtypes = [f32p, C.c_int, C.c_void_p, C.c_int, C.c_int, f32p, C.c_void_p]
        L.orc_write_features.restype = C.c_long
        L.orc_write_features.argtypes = [f32p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, f32p]
        L.orc_read_features.argtypes = [f32p, C.c_long, C.c_int, C.c_int, f32p]
        L.orc_corpus_mean_var.argtypes = [f32p, C.c_int, C.c_long, C.c_int, f32p, f32p]
        L.orc_gaussian_logprob.restype = C.c_float
        L.orc_gaussian_logprob.argtypes = [f32p, ldp, ldp, C.c_float, C.c_int]
        L.orc_gmm_score.argtypes = [f32p, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, ldp, ldp, f32p, C.c_int, f32p]
        L.orc_set_wt.argtypes = [f32p, f32p, C.c_long]
        L.orc_build_training_hmm.argtypes = [i32p, C.c_int, i32p, f32p, i32p]
        L.orc_viterbi.restype = C.c_int
        L.orc_viterbi.argtypes = [f32p, C.c_long, C.c_void_p, C.c_long, C.c_int, i32p, f32p, C.c_float, C.c_int,
                                  i32p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_long)]
        L.orc_train_transitions.argtypes = [i32p, C.c_long, i32p, f32p]
        L.orc_em_statistics.argtypes = [f32p, C.c_int, C.c_long, i32p, C.c_int, C.c_int, C.c_int,
                                        ldp, ldp, f32p, f32p, ldp, ldp, f32p, f32p, ldp, C.c_float]
        L.orc_state_model.argtypes = [i32p, f32p, C.c_long, C.c_int, f32p, i32p]
        L.orc_build_lexicon_hmm.restype = C.c_long
        L.orc_build_lexicon_hmm.argtypes = [C.c_int, i32p, i32p, f32p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.POINTER(C.c_long)]
        L.orc_viterbi_graph.restype = C.c_int
        L.orc_viterbi_graph.argtypes = [f32p, C.c_long, i32p, C.c_long, C.c_int, i32p, i32p, f32p, C.c_float, C.c_int, C.c_int,
                                        i32p, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_long)]
        L.orc_update_states.argtypes = [C.c_int, i32p, f32p, i32p, f32p, i32p]
        L.orc_pipeline_utterance.restype = C.c_long
        L.orc_pipeline_utterance.argtypes = [f32p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                             C.c_int, C.c_int, C.c_int, C.c_int, ldp, ldp, f32p, C.c_int, i32p, C.c_int,
                                             i32p, f32p, i32p, C.POINTER(C.c_float)]
        _lib = L
    return _lib


# ---------------------------------------------------------------- filterbank / MFCC
def mel_filterbank(end=7200.0, n=48, dft_bins=320, sps=16000):
    fb = np.zeros(n, dtype=FILTER_DT)
    bin_max = lib().orc_mel_filterbank(end, n, dft_bins, sps, fb.ctypes.data)
    return fb, bin_max


def num_frames(L, wl, hop):
    return int(lib().orc_num_frames(L, wl, hop))


def normalize_volume(x, target=1.0):
    x = np.ascontiguousarray(x, dtype=np.float32).copy()
    lib().orc_normalize_volume(x, len(x), target)
    return x


def preemphasis(x, alpha=0.96):
    x = np.ascontiguousarray(x, dtype=np.float32).copy()
    lib().orc_preemphasis(x, len(x), alpha)
    return x


def partial_dft(frame, N, nbins):
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    out = np.zeros(max(nbins, N), np.float32)
    lib().orc_partial_dft(frame, N, nbins, out)
    return out[:nbins]


def partial_dft_win(frame, win, N, nbins, stack=False, out=None):
    """Clip::partial_dft(out, nbins, stack) with window values win[N/2] (None: rectangular)."""
    frame = np.ascontiguousarray(frame, dtype=np.float32)
    out = np.zeros(nbins, np.float32) if out is None else np.ascontiguousarray(out, np.float32).copy()
    w = None if win is None else np.ascontiguousarray(win, np.float32)
    lib().orc_partial_dft_win(frame, None if w is None else w.ctypes.data, N, nbins, 1 if stack else 0, out)
    return out


def hamming(n, N):
    return np.float32(lib().orc_hamming(n, N))


def filterbank_apply(fb, spec):
    spec = np.ascontiguousarray(spec, dtype=np.float32)
    out = np.zeros(len(fb), np.float32)
    lib().orc_filterbank_apply(fb.ctypes.data, len(fb), spec, out)
    return out


def log1p(buf):
    buf = np.ascontiguousarray(buf, dtype=np.float32).copy()
    lib().orc_log1p(buf.reshape(-1), buf.size)
    return buf


def dct(filtered, nparams):
    filtered = np.ascontiguousarray(filtered, dtype=np.float32)
    out = np.zeros(nparams, np.float32)
    lib().orc_dct(filtered, len(filtered), nparams, out)
    return out


def compute_deltas(coeffs):
    """compute_deltas(vector) on coefficients [n, nparams] -> [n, 3*nparams]."""
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float32)
    n, npar = coeffs.shape
    feats = np.zeros((n, 3 * npar), np.float32)
    feats[:, :npar] = coeffs
    lib().orc_compute_deltas(feats, n, npar)
    return feats


def gaussian_logprob(fv, mean, var, logwt):
    """gaussian::operator() of one Gaussian on one feature vector."""
    fv = np.ascontiguousarray(fv, dtype=np.float32)
    mean = np.ascontiguousarray(mean, dtype=np.longdouble)
    var = np.ascontiguousarray(var, dtype=np.longdouble)
    return np.float32(lib().orc_gaussian_logprob(fv, mean, var, np.float32(logwt), len(mean)))


def write_features(x, wl=320, hop=160, fb=None, nparams=13, alpha=0.96):
    """Returns (ceps [n, nparams], x_after_preemphasis).  Input is not modified."""
    x = np.ascontiguousarray(x, dtype=np.float32).copy()
    n = num_frames(len(x), wl, hop)
    ceps = np.zeros((n, nparams), np.float32)
    got = lib().orc_write_features(x, len(x), wl, hop, fb.ctypes.data, len(fb), nparams, alpha, ceps)
    assert got == n
    return ceps, x


def read_features(raw, WL=2):
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    n, npar = raw.shape
    feats = np.zeros((n, 3 * npar), np.float32)
    lib().orc_read_features(raw, n, npar, WL, feats)
    return feats


def corpus_mean_var(feats, D):
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    mu = np.zeros(D, np.float32)
    var = np.zeros(D, np.float32)
    lib().orc_corpus_mean_var(feats, feats.shape[1], feats.shape[0], D, mu, var)
    return mu, var


# ---------------------------------------------------------------- GMM
def set_wt(w):
    """logwt = logf(weight), gaussian::set_wt src/gmm.cpp:35-38 (glibc logf)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    lw = np.zeros_like(w)
    lib().orc_set_wt(w.reshape(-1), lw.reshape(-1), w.size)
    return lw


def gmm_score(feats, mean, var, logwt, D, mode=0):
    """feats [T, ld] f32; mean/var [G, M, D] (any float dtype, used as long double);
    logwt [G, M] f32 -> [T, G] f32."""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    mean = np.ascontiguousarray(mean, dtype=np.longdouble)
    var = np.ascontiguousarray(var, dtype=np.longdouble)
    logwt = np.ascontiguousarray(logwt, dtype=np.float32)
    G, M, D_ = mean.shape
    assert D_ == D
    T = feats.shape[0]
    out = np.zeros((T, G), np.float32)
    lib().orc_gmm_score(feats, feats.shape[1], T, G, M, D, mean, var, logwt, mode, out)
    return out


def gmm_score_shifted(feats, mean, var, logwt, D):
    """log sum_m exp(g_m) taken relative to the largest component: the oracle's per-component log-densities (mode 0 on
    the [G*M, 1, D] view of the model) combined in f64 numpy.  Equal to gmm_score(..., mode=1) wherever that is finite;
    finite where mode 1 underflows to -inf.  This is what the library returns there (parity unpinned: gmm.cpp:103-107 is
    dead code in the reference, and -inf is what it would give)."""
    mean = np.asarray(mean)
    G, M, _ = mean.shape
    comp = gmm_score(feats, mean.reshape(G * M, 1, D), np.asarray(var).reshape(G * M, 1, D),
                     np.asarray(logwt, np.float32).reshape(G * M, 1), D, 0).astype(np.float64).reshape(-1, G, M)
    mx = comp.max(axis=2, keepdims=True)
    with np.errstate(invalid="ignore"):
        return (np.log(np.exp(comp - mx).sum(axis=2)) + mx[:, :, 0]).astype(np.float32)


# ---------------------------------------------------------------- HMM
def build_training_hmm(phones):
    phones = np.ascontiguousarray(phones, dtype=np.int32)
    S = len(phones)
    ntr = np.zeros(S, np.int32)
    tr = np.zeros((S, 3), np.float32)
    ctx = np.zeros((S, 3), np.int32)
    lib().orc_build_training_hmm(phones, S, ntr, tr, ctx)
    return ntr, tr, ctx


def viterbi(emis, tied, ntr, trlogp, beam=1e10, order=0):
    """Returns dict(status, path, final, prevp, nties)."""
    emis = np.ascontiguousarray(emis, dtype=np.float32)
    T, ldE = emis.shape
    S = len(ntr)
    ntr = np.ascontiguousarray(ntr, dtype=np.int32)
    trlogp = np.ascontiguousarray(trlogp, dtype=np.float32)
    tied_arr = None if tied is None else np.ascontiguousarray(tied, dtype=np.int32)
    path = np.full(T, -1, np.int32)
    final = C.c_float(0)
    prevp = C.c_float(0)
    nties = C.c_long(0)
    rc = lib().orc_viterbi(emis, ldE, None if tied_arr is None else tied_arr.ctypes.data, T, S, ntr, trlogp,
                           beam, order, path, C.byref(final), C.byref(prevp), C.byref(nties))
    return dict(status=rc, path=path, final=np.float32(final.value), prevp=np.float32(prevp.value),
                nties=int(nties.value))


def train_transitions(path, ntr, trlogp):
    trlogp = np.ascontiguousarray(trlogp, dtype=np.float32).copy()
    path = np.ascontiguousarray(path, dtype=np.int32)
    lib().orc_train_transitions(path, len(path), np.ascontiguousarray(ntr, dtype=np.int32), trlogp)
    return trlogp


def state_model(cls, adv, C_):
    """state_model::augment over the frames of the final paths (src/hmm.cpp:46-51, 350-356): cls[t] = tied class of
    the path state at frame t, adv[t] = its get_adv_prob().  Returns (probsum f32 [C], count i32 [C])."""
    cls = np.ascontiguousarray(cls, dtype=np.int32)
    adv = np.ascontiguousarray(adv, dtype=np.float32)
    ps = np.zeros(C_, np.float32)
    cnt = np.zeros(C_, np.int32)
    lib().orc_state_model(cls, adv, len(cls), C_, ps, cnt)
    return ps, cnt


def build_lexicon_hmm(word_phones, uni_logp, lmsf=5.0):
    """hmm::hmm(pronlex&, unigram_model&, ...)  src/hmm.cpp:92-129.  word_phones: list of int arrays in the map's
    (sorted-by-spelling) order.  Returns dict(n_states, tr_off, tr_tgt, tr_logp, ctx [n,3], word_of)."""
    n = len(word_phones)
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum([len(p) for p in word_phones])
    ph = np.ascontiguousarray(np.concatenate(word_phones) if n else np.zeros(0), dtype=np.int32)
    if len(ph) == 0:
        ph = np.zeros(1, np.int32)
    lp = np.ascontiguousarray(uni_logp, dtype=np.float32)
    ntr = C.c_long(0)
    ns = lib().orc_build_lexicon_hmm(n, off, ph, lp, lmsf, None, None, None, None, None, C.byref(ntr))
    tr_off = np.zeros(ns + 1, np.int32)
    tr_tgt = np.zeros(ntr.value, np.int32)
    tr_logp = np.zeros(ntr.value, np.float32)
    ctx = np.zeros((ns, 3), np.int32)
    word_of = np.zeros(ns, np.int32)
    lib().orc_build_lexicon_hmm(n, off, ph, lp, lmsf, tr_off.ctypes.data, tr_tgt.ctypes.data, tr_logp.ctypes.data, ctx.ctypes.data,
                                word_of.ctypes.data, C.byref(ntr))
    return dict(n_states=int(ns), tr_off=tr_off, tr_tgt=tr_tgt, tr_logp=tr_logp, ctx=ctx, word_of=word_of)


def viterbi_graph(emis, tied, tr_off, tr_tgt, tr_logp, beam=1e10, order=0, final_mode=0):
    """hmm::viterbi on an arbitrary transition list.  Returns dict(status, path, final, final_state, nties)."""
    emis = np.ascontiguousarray(emis, dtype=np.float32)
    T, ldE = emis.shape
    n = len(tr_off) - 1
    path = np.full(T, -1, np.int32)
    final, fstate, nties = C.c_float(0), C.c_int(-1), C.c_long(0)
    rc = lib().orc_viterbi_graph(emis, ldE, np.ascontiguousarray(tied, np.int32), T, n, np.ascontiguousarray(tr_off, np.int32),
                                 np.ascontiguousarray(tr_tgt, np.int32), np.ascontiguousarray(tr_logp, np.float32), beam, order,
                                 final_mode, path, C.byref(final), C.byref(fstate), C.byref(nties))
    return dict(status=rc, path=path, final=np.float32(final.value), final_state=int(fstate.value), nties=int(nties.value))


def update_states(tr_off, tr_logp, cls, probsum, count):
    """hmm::update_states  src/hmm.cpp:361-366; returns the new transition log-probs."""
    tr = np.ascontiguousarray(tr_logp, np.float32).copy()
    lib().orc_update_states(len(tr_off) - 1, np.ascontiguousarray(tr_off, np.int32), tr, np.ascontiguousarray(cls, np.int32),
                            np.ascontiguousarray(probsum, np.float32), np.ascontiguousarray(count, np.int32))
    return tr


def em_statistics(feats, gid, D, A_mean, A_var, A_weight, blend_keep=0.2):
    """One statistics+update half-iteration.  Returns (B_mean, B_var, B_weight, B_logwt, zetasum)."""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    gid = np.ascontiguousarray(gid, dtype=np.int32)
    A_mean = np.ascontiguousarray(A_mean, dtype=np.longdouble)
    A_var = np.ascontiguousarray(A_var, dtype=np.longdouble)
    A_weight = np.ascontiguousarray(A_weight, dtype=np.float32)
    A_logwt = set_wt(A_weight)
    G, M, _ = A_mean.shape
    B_mean = np.zeros_like(A_mean)
    B_var = np.zeros_like(A_var)
    B_w = np.zeros_like(A_weight)
    B_lw = np.zeros_like(A_weight)
    zs = np.zeros((G, M), np.longdouble)
    lib().orc_em_statistics(feats, feats.shape[1], feats.shape[0], gid, G, M, D, A_mean, A_var, A_weight, A_logwt,
                            B_mean, B_var, B_w, B_lw, zs, blend_keep)
    return B_mean, B_var, B_w, B_lw, zs


def pipeline_utterance(x, wl, hop, fb, nparams, alpha, WL, mean, var, logwt, D, mode, tied, ntr, trlogp):
    x = np.ascontiguousarray(x, dtype=np.float32).copy()
    mean = np.ascontiguousarray(mean, dtype=np.longdouble)
    var = np.ascontiguousarray(var, dtype=np.longdouble)
    logwt = np.ascontiguousarray(logwt, dtype=np.float32)
    G, M, _ = mean.shape
    n = num_frames(len(x), wl, hop)
    path = np.zeros(n, np.int32)
    final = C.c_float(0)
    tied = np.ascontiguousarray(tied, dtype=np.int32)
    got = lib().orc_pipeline_utterance(x, len(x), wl, hop, fb.ctypes.data, len(fb), nparams, alpha, WL, G, M, D,
                                       mean, var, logwt, mode, tied, len(tied),
                                       np.ascontiguousarray(ntr, dtype=np.int32),
                                       np.ascontiguousarray(trlogp, dtype=np.float32), path, C.byref(final))
    return int(got), path, np.float32(final.value)
