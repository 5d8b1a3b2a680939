"""Python host layer over the C ABI (include/srx.h).

Names follow the reference's (mel_filterbank, write_features, read_features, acoustic_model, hmm
viterbi / train): see SURVEY.md section 8b for the C++ surface they mirror.  The authoritative
drop-in for the C++ callers is speechrecognition_amd/host/srx_host.hpp; this module exists so the
parity tests, bench.py and the multi-GPU driver can call the same entry points from Python.
Every computation here runs in libsrx.so on the GPU; numpy is used for marshalling only.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (SRX_MIX_COMPONENT0, SRX_MIX_LOGSUMEXP, SRX_SCORE_EXACT, SRX_SCORE_MFMA, check, srx_filter,
                    srx_hmm_batch, srx_mfcc_cfg)

FILTER_DT = np.dtype([("center", np.float32), ("width", np.float32), ("height", np.float32)])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """srx_ctx: one per GPU.  `stream` is a raw hipStream_t handle (int) or None."""

    def __init__(self, device=0, stream=None):
        self.lib = _capi.load()
        h = C.c_void_p()
        check(self.lib.srx_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            self.lib.srx_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        check(self.lib.srx_sync(self.h), self.h)

    def stream_handle(self):
        """The hipStream_t this context launches on, as an integer (for torch.cuda.ExternalStream)."""
        return int(self.lib.srx_ctx_stream(self.h) or 0)

    def profile_enable(self, on=True):
        """Start (and clear) / stop per-kernel event timing on this context's stream."""
        check(self.lib.srx_profile_enable(self.h, int(on)), self.h)

    def profile_query(self, kernel):
        """(total_ms, launches) of `kernel` since profile_enable(True); synchronises the stream."""
        ms, n = C.c_double(0), C.c_int64(0)
        check(self.lib.srx_profile_query(self.h, kernel.encode(), C.byref(ms), C.byref(n)), self.h)
        return ms.value, n.value

    # --- device arrays ---
    def empty(self, shape, dtype):
        return DevArray(self, shape, dtype)

    def zeros(self, shape, dtype):
        a = DevArray(self, shape, dtype)
        check(self.lib.srx_memset(self.h, a.ptr, 0, a.nbytes), self.h)
        return a

    def array(self, host):
        host = np.ascontiguousarray(host)
        a = DevArray(self, host.shape, host.dtype)
        a.upload(host)
        return a


class DevArray:
    """A typed device allocation owned through srx_malloc/srx_free."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = (int(shape),) if isinstance(shape, (int, np.integer)) else tuple(int(v) for v in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        check(ctx.lib.srx_malloc(ctx.h, self.nbytes, C.byref(p)), ctx.h)
        self.ptr = p

    @classmethod
    def wrap(cls, ctx, ptr, shape, dtype, owner=None):
        """A non-owning view of device memory allocated elsewhere (e.g. a torch tensor: pass it as `owner`
        to keep it alive)."""
        a = cls.__new__(cls)
        a.ctx = ctx
        a.shape = (int(shape),) if isinstance(shape, (int, np.integer)) else tuple(int(v) for v in shape)
        a.dtype = np.dtype(dtype)
        a.nbytes = int(np.prod(a.shape, dtype=np.int64)) * a.dtype.itemsize
        a.ptr = C.c_void_p(int(ptr))
        a.owner = owner
        a.borrowed = True
        return a

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.nbytes == self.nbytes, (host.shape, self.shape)
        check(self.ctx.lib.srx_memcpy_h2d(self.ctx.h, self.ptr, _ptr(host), self.nbytes), self.ctx.h)

    def get(self):
        out = np.empty(self.shape, self.dtype)
        check(self.ctx.lib.srx_memcpy_d2h(self.ctx.h, _ptr(out), self.ptr, self.nbytes), self.ctx.h)
        return out

    def free(self):
        if getattr(self, "borrowed", False):
            self.ptr = None
            self.owner = None
            return
        if self.ptr and self.ctx.h:
            self.ctx.lib.srx_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ----------------------------------------------------------------------------- filterbank / MFCC
def mel_filterbank(end=7200.0, n=48, dft_bins=320, sps=16000):
    """mel_filterbank(end, n, dft_bins, sps)  src/mfcc.cpp:190-201 -> (filters, bin_max)."""
    lib = _capi.load()
    fb = np.zeros(n, dtype=FILTER_DT)
    bm = C.c_int(0)
    check(lib.srx_mel_filterbank(end, n, dft_bins, sps, fb.ctypes.data_as(C.POINTER(srx_filter)), C.byref(bm)))
    return fb, bm.value


def mfcc_config(window_len=320, frame_shift=160, n_ceps=13, normalize=True, target_rms=1.0, preemphasis=True,
                alpha=0.96):
    return srx_mfcc_cfg(window_len, frame_shift, n_ceps, int(normalize), target_rms, int(preemphasis), alpha)


def frame_offsets(cfg, sample_off):
    sample_off = np.ascontiguousarray(sample_off, dtype=np.int64)
    n = len(sample_off) - 1
    fo = np.zeros(n + 1, np.int64)
    check(_capi.load().srx_mfcc_frame_offsets(C.byref(cfg), _ptr(sample_off), n, _ptr(fo)))
    return fo


MFCC_AUTO, MFCC_F32, MFCC_F16X3, MFCC_FFT = 0, 1, 2, 3
DECODE_AUTO, DECODE_DENSE, DECODE_FRONTIER = 0, 1, 2


class MfccPlan:
    def __init__(self, ctx, cfg, fb, path=MFCC_AUTO):
        self.ctx, self.cfg = ctx, cfg
        fb = np.ascontiguousarray(fb, dtype=FILTER_DT)
        h = C.c_void_p()
        check(ctx.lib.srx_mfcc_plan_create(ctx.h, C.byref(cfg), fb.ctypes.data_as(C.POINTER(srx_filter)), len(fb),
                                           C.byref(h)), ctx.h)
        self.h = h
        self.bin_max = ctx.lib.srx_mfcc_plan_bin_max(h)
        if path != MFCC_AUTO:
            self.set_path(path)

    def set_path(self, path):
        check(self.ctx.lib.srx_mfcc_plan_set_path(self.h, path), self.ctx.h)

    def close(self):
        if self.h:
            self.ctx.lib.srx_mfcc_plan_destroy(self.h)
            self.h = None

    def write_features(self, samples, sample_off):
        """Batch form of write_features (src/mfcc.cpp:235-275) on host arrays.
        Returns (ceps [total_frames, n_ceps], frame_off)."""
        samples = np.ascontiguousarray(samples, dtype=np.float32)
        sample_off = np.ascontiguousarray(sample_off, dtype=np.int64)
        n = len(sample_off) - 1
        fo = frame_offsets(self.cfg, sample_off)
        ceps = np.zeros((int(fo[-1]), self.cfg.n_ceps), np.float32)
        fo2 = np.zeros(n + 1, np.int64)
        check(self.ctx.lib.srx_mfcc_batch(self.ctx.h, self.h, _ptr(samples), _ptr(sample_off), n, _ptr(ceps),
                                          _ptr(fo2)), self.ctx.h)
        assert (fo == fo2).all()
        return ceps, fo

    def partial_dft(self, samples, sample_off):
        """Clip::partial_dft (src/clip.cpp:174-194) of every frame -> (spec [total_frames, bin_max], frame_off)."""
        samples = np.ascontiguousarray(samples, dtype=np.float32)
        sample_off = np.ascontiguousarray(sample_off, dtype=np.int64)
        n = len(sample_off) - 1
        fo = frame_offsets(self.cfg, sample_off)
        spec = np.zeros((int(fo[-1]), self.bin_max), np.float32)
        fo2 = np.zeros(n + 1, np.int64)
        check(self.ctx.lib.srx_partial_dft_batch(self.ctx.h, self.h, _ptr(samples), _ptr(sample_off), n, _ptr(spec), _ptr(fo2)),
              self.ctx.h)
        return spec, fo

    def write_features_dev(self, samples_d, sample_off_d, n_utt, frame_off_d, total_frames, scale_d, ceps_d):
        check(self.ctx.lib.srx_mfcc_batch_dev(self.ctx.h, self.h, samples_d.ptr, sample_off_d.ptr, n_utt,
                                              frame_off_d.ptr, total_frames, scale_d.ptr if scale_d else None,
                                              ceps_d.ptr), self.ctx.h)


def read_features(ctx, ceps, frame_off, WL=2):
    """Smoothing + deltas of read_features (src/mfcc.cpp:296-312) on host arrays -> [T, 3*n_ceps]."""
    ceps = np.ascontiguousarray(ceps, dtype=np.float32)
    frame_off = np.ascontiguousarray(frame_off, dtype=np.int64)
    T, nc = ceps.shape
    feats = np.zeros((T, 3 * nc), np.float32)
    check(ctx.lib.srx_features_batch(ctx.h, WL, nc, _ptr(ceps), _ptr(frame_off), len(frame_off) - 1, _ptr(feats)), ctx.h)
    return feats


def read_features_dev(ctx, WL, n_ceps, ceps_d, frame_off_d, n_utt, total_frames, feats_d):
    check(ctx.lib.srx_features_batch_dev(ctx.h, WL, n_ceps, ceps_d.ptr, frame_off_d.ptr, n_utt, total_frames,
                                         feats_d.ptr), ctx.h)


def normalize_volume(ctx, samples, sample_off, target=1.0):
    """Clip::normalizeVolume (src/clip.cpp:151-154) per utterance; returns the scaled copy."""
    x = ctx.array(np.ascontiguousarray(samples, np.float32))
    so = ctx.array(np.ascontiguousarray(sample_off, np.int64))
    n = len(sample_off) - 1
    sc = ctx.empty((n,), np.float32)
    check(ctx.lib.srx_normalize_volume_dev(ctx.h, x.ptr, so.ptr, n, target, sc.ptr), ctx.h)
    return x.get()


def preemphasis(ctx, samples, sample_off, alpha=0.96):
    """Clip::preemphasis (src/clip.cpp:199-205) per utterance; returns the filtered copy."""
    x = ctx.array(np.ascontiguousarray(samples, np.float32))
    tmp = ctx.empty(x.shape, np.float32)
    so = ctx.array(np.ascontiguousarray(sample_off, np.int64))
    n = len(sample_off) - 1
    check(ctx.lib.srx_preemphasis_dev(ctx.h, x.ptr, tmp.ptr, so.ptr, n, x.shape[0], alpha), ctx.h)
    return x.get()


def corpus_mean_var(ctx, feats, D):
    """mean(ut), variance(ut, mu)  src/utterance.cpp:32-65."""
    feats = np.ascontiguousarray(feats, np.float32)
    T, ld = feats.shape
    f = ctx.array(feats)
    s = ctx.zeros((2 * D,), np.float64)
    check(ctx.lib.srx_corpus_mean_var_dev(ctx.h, f.ptr, ld, T, D, s.ptr), ctx.h)
    r = s.get()
    mu = (r[:D].astype(np.float32) / np.float32(T)).astype(np.float32)
    var = (r[D:] / T).astype(np.float32)
    return mu, var


# ----------------------------------------------------------------------------- acoustic model
class AcousticModel:
    """acoustic_model (src/gmm.hpp:75-94): G tied states x M diagonal Gaussians x D dims, device resident."""

    def __init__(self, ctx, G, M, D):
        self.ctx, self.G, self.M, self.D = ctx, G, M, D
        h = C.c_void_p()
        check(ctx.lib.srx_model_create(ctx.h, G, M, D, C.byref(h)), ctx.h)
        self.h = h

    def close(self):
        if self.h:
            self.ctx.lib.srx_model_destroy(self.h)
            self.h = None

    def set(self, mean, var, weight):
        mean = np.ascontiguousarray(mean, np.float64).reshape(self.G, self.M, self.D)
        var = np.ascontiguousarray(var, np.float64).reshape(self.G, self.M, self.D)
        weight = np.ascontiguousarray(weight, np.float32).reshape(self.G, self.M)
        check(self.ctx.lib.srx_model_set(self.h, _ptr(mean), _ptr(var), _ptr(weight)), self.ctx.h)

    def get(self):
        mean = np.zeros((self.G, self.M, self.D), np.float64)
        var = np.zeros_like(mean)
        w = np.zeros((self.G, self.M), np.float32)
        check(self.ctx.lib.srx_model_get(self.h, _ptr(mean), _ptr(var), _ptr(w)), self.ctx.h)
        return mean, var, w

    def copy_from(self, other):
        check(self.ctx.lib.srx_model_copy(self.h, other.h), self.ctx.h)

    def version(self):
        """srx_model_version: advances whenever the parameters are rewritten."""
        return int(self.ctx.lib.srx_model_version(self.h))

    def score(self, feats, mix_mode=SRX_MIX_COMPONENT0, path=SRX_SCORE_MFMA):
        """acoustic_model::operator() for every frame x tied state, host arrays -> [T, G]."""
        feats = np.ascontiguousarray(feats, np.float32)
        T, ld = feats.shape
        out = np.zeros((T, self.G), np.float32)
        check(self.ctx.lib.srx_gmm_score(self.ctx.h, self.h, mix_mode, path, _ptr(feats), ld, T, _ptr(out)), self.ctx.h)
        return out

    def score_dev(self, feats_d, ld, T, out_d, stride_t, stride_g, mix_mode=SRX_MIX_COMPONENT0, path=SRX_SCORE_MFMA):
        """Score (t, g) -> out[t*stride_t + g*stride_g] (frame-major: (G, 1); state-major: (1, T))."""
        check(self.ctx.lib.srx_gmm_score_dev(self.ctx.h, self.h, mix_mode, path, feats_d.ptr, ld, T, out_d.ptr, stride_t,
                                             stride_g), self.ctx.h)


# ----------------------------------------------------------------------------- HMM batch
def build_training_hmm(phones):
    """hmm::hmm(vector<phone>, acoustic_model*)  src/hmm.cpp:56-88 -> (ntr, trlogp[S,3], ctx[S,3])."""
    phones = np.ascontiguousarray(phones, np.int32)
    S = len(phones)
    ntr = np.zeros(S, np.int32)
    tr = np.zeros((S, 3), np.float32)
    ctx3 = np.zeros((S, 3), np.int32)
    check(_capi.load().srx_hmm_build_training(_ptr(phones), S, _ptr(ntr), _ptr(tr), _ptr(ctx3)))
    return ntr, tr, ctx3


# ---- the MFCC stages and the single Gaussian as calls of their own (csrc/srx_frameops.hip: reference-order kernels) ----
def _as_filters(fb):
    fb = np.ascontiguousarray(fb, dtype=FILTER_DT)
    return fb, fb.ctypes.data_as(C.POINTER(srx_filter))


def partial_dft_frames(ctx, frames, window, len_, stack=False, out=None):
    """Clip::partial_dft(out, len, stack) (src/clip.cpp:174-194) of every row of frames [n, N]; window: [N/2] values of
    the clip's window function or None (rectangular)."""
    frames = np.ascontiguousarray(frames, np.float32)
    n, N = frames.shape
    res = np.zeros((n, len_), np.float32) if out is None else np.ascontiguousarray(out, np.float32).copy()
    w = None if window is None else np.ascontiguousarray(window, np.float32)
    check(ctx.lib.srx_partial_dft_frames(ctx.h, _ptr(frames), None if w is None else _ptr(w), N, len_, n, int(stack), _ptr(res)), ctx.h)
    return res


def filterbank_apply(ctx, fb, spec):
    """filterbank::operator() (src/mfcc.cpp:143-167) of every row of spec [n, n_bins] -> [n, len(fb)]."""
    spec = np.ascontiguousarray(spec, np.float32)
    fb, fbp = _as_filters(fb)
    out = np.zeros((spec.shape[0], len(fb)), np.float32)
    check(ctx.lib.srx_filterbank_apply(ctx.h, fbp, len(fb), _ptr(spec), spec.shape[1], spec.shape[0], _ptr(out)), ctx.h)
    return out


def log1p(ctx, buf):
    """log(buf, len) of src/utils.hpp:55-60 (a copy is returned)."""
    buf = np.ascontiguousarray(buf, np.float32).copy()
    check(ctx.lib.srx_log1p(ctx.h, _ptr(buf), buf.size), ctx.h)
    return buf


def dct(ctx, filtered, nparams):
    """dct (src/mfcc.cpp:205-214) of every row of filtered [n, len] -> [n, nparams]."""
    filtered = np.ascontiguousarray(filtered, np.float32)
    out = np.zeros((filtered.shape[0], nparams), np.float32)
    check(ctx.lib.srx_dct(ctx.h, _ptr(filtered), filtered.shape[1], nparams, filtered.shape[0], _ptr(out)), ctx.h)
    return out


def delta(ctx, hi, lo, span):
    """featurevec::compute_deltas of one vector (src/mfcc.cpp:73-92): (hi - lo) / span."""
    hi, lo = np.ascontiguousarray(hi, np.float32), np.ascontiguousarray(lo, np.float32)
    out = np.zeros_like(hi)
    check(ctx.lib.srx_delta(ctx.h, _ptr(hi), _ptr(lo), int(span), _ptr(out), hi.size), ctx.h)
    return out


def gaussian_logprob(ctx, feats, mean, var, logwt):
    """gaussian::operator() (src/gmm.cpp:41-49) of one free-standing Gaussian on every row of feats [T, ld >= D]."""
    feats = np.ascontiguousarray(feats, np.float32)
    mean, var = np.ascontiguousarray(mean, np.float64), np.ascontiguousarray(var, np.float64)
    out = np.zeros(feats.shape[0], np.float32)
    check(ctx.lib.srx_gaussian_logprob(ctx.h, _ptr(feats), feats.shape[1], feats.shape[0], len(mean), _ptr(mean), _ptr(var),
                                       float(logwt), _ptr(out)), ctx.h)
    return out


class HmmBatch:
    """A ragged batch of per-utterance training HMMs on the device (utterance.hmm of src/utterance.hpp:30)."""

    def __init__(self, ctx, frame_off, ntr_list, trlogp_list, tied_list):
        self.ctx = ctx
        self.n_utt = len(ntr_list)
        so = np.zeros(self.n_utt + 1, np.int32)
        for u, n in enumerate(ntr_list):
            so[u + 1] = so[u] + len(n)
        self.state_off = so
        self.frame_off = np.ascontiguousarray(frame_off, np.int64)
        self.max_states = int(max(len(n) for n in ntr_list)) if self.n_utt else 1
        self.frame_off_d = ctx.array(self.frame_off)
        self.state_off_d = ctx.array(so)
        self.tied_d = ctx.array(np.concatenate(tied_list).astype(np.int32))
        self.ntr_d = ctx.array(np.concatenate(ntr_list).astype(np.int32))
        self.trlogp_d = ctx.array(np.concatenate(trlogp_list).astype(np.float32).reshape(-1, 3))
        self.total_frames = int(self.frame_off[-1])
        self.c = srx_hmm_batch(self.n_utt, self.frame_off_d.ptr, self.state_off_d.ptr, self.tied_d.ptr, self.ntr_d.ptr,
                               self.trlogp_d.ptr, self.max_states, self.total_frames)

    def trlogp(self):
        return self.trlogp_d.get()

    def c_ref(self):
        return C.byref(self.c)

    def viterbi(self, emis_d, stride_t, stride_g=1, beam=1e10, path_d=None, final_d=None, status_d=None, noalloc=False):
        """hmm::viterbi for every utterance (src/hmm.cpp:150-224); emission (t, col) at
        emis[t*stride_t + col*stride_g] (frame-major [T, G]: (G, 1); state-major [G, T]: (1, T)).
        noalloc: srx_viterbi_batch_noalloc_dev (back-pointers in the library's fixed per-GPU array; the caller should
        then pass the three output buffers too, or this wrapper allocates them)."""
        ctx = self.ctx
        path_d = path_d or ctx.zeros((max(self.total_frames, 1),), np.int32)
        final_d = final_d or ctx.zeros((max(self.n_utt, 1),), np.float32)
        status_d = status_d or ctx.zeros((max(self.n_utt, 1),), np.int32)
        fn = ctx.lib.srx_viterbi_batch_noalloc_dev if noalloc else ctx.lib.srx_viterbi_batch_dev
        check(fn(ctx.h, C.byref(self.c), emis_d.ptr, stride_t, stride_g, beam, path_d.ptr, final_d.ptr, status_d.ptr), ctx.h)
        return path_d, final_d, status_d

    def train_transition_probabilities(self, path_d, status_d):
        check(self.ctx.lib.srx_train_transitions_dev(self.ctx.h, C.byref(self.c), path_d.ptr, status_d.ptr), self.ctx.h)

    def state_model(self, cls, path_d, status_d, n_classes):
        """state_model::augment over every frame of every final path (src/hmm.cpp:46-51, 350-356): cls[n_states] = tied
        class of each state (-1: skip) -> (probsum f64 [C], count i32 [C])."""
        ctx = self.ctx
        cls_d = ctx.array(np.ascontiguousarray(cls, np.int32))
        ps, cnt = ctx.empty((n_classes,), np.float64), ctx.empty((n_classes,), np.int32)
        check(ctx.lib.srx_state_model_dev(ctx.h, C.byref(self.c), cls_d.ptr, path_d.ptr, status_d.ptr, n_classes, ps.ptr, cnt.ptr),
              ctx.h)
        ctx.sync()
        return ps.get(), cnt.get()

    # ---- alignment-mode scoring: each utterance against its own states, compact emissions ----
    def align_plan(self):
        """Layout of the compact emission buffer (utterance u: frame-major block [T_u][S_u] at emis_off[u])."""
        if getattr(self, "_plan", None) is None:
            h = C.c_void_p()
            check(self.ctx.lib.srx_align_plan_create(self.ctx.h, _ptr(self.frame_off), _ptr(self.state_off), self.n_utt, C.byref(h)),
                  self.ctx.h)
            self._plan = h
            self.emis_count = int(self.ctx.lib.srx_align_plan_emission_count(h))
            self.emis_off = np.zeros(self.n_utt + 1, np.int64)
            check(self.ctx.lib.srx_align_plan_emission_offsets(h, _ptr(self.emis_off)), self.ctx.h)
        return self._plan

    def cache_frames(self, on=True):
        """srx_align_plan_cache_frames: promise that the feature buffer scored with this batch's plan stays unchanged
        (training iterations), so the frame-side operands are built once."""
        check(self.ctx.lib.srx_align_plan_cache_frames(self.align_plan(), int(on)), self.ctx.h)

    def score_align(self, model, feats_d, ld, emis_d=None, mix_mode=SRX_MIX_COMPONENT0, path=None):
        """(*acm)(fv, ctx) for every frame and every state of the frame's own utterance (src/hmm.cpp:166)."""
        from ._capi import SRX_SCORE_MFMA_F16X3
        plan = self.align_plan()
        emis_d = emis_d or self.ctx.empty((max(self.emis_count, 1),), np.float32)
        check(self.ctx.lib.srx_gmm_score_align_dev(self.ctx.h, model.h, mix_mode, SRX_SCORE_MFMA_F16X3 if path is None else path,
                                                   feats_d.ptr, ld, C.byref(self.c), plan, emis_d.ptr), self.ctx.h)
        return emis_d

    def viterbi_align(self, emis_d, beam=1e10, path_d=None, final_d=None, status_d=None):
        ctx = self.ctx
        path_d = path_d or ctx.zeros((max(self.total_frames, 1),), np.int32)
        final_d = final_d or ctx.zeros((max(self.n_utt, 1),), np.float32)
        status_d = status_d or ctx.zeros((max(self.n_utt, 1),), np.int32)
        check(ctx.lib.srx_viterbi_align_dev(ctx.h, C.byref(self.c), self.align_plan(), emis_d.ptr, beam, path_d.ptr, final_d.ptr,
                                            status_d.ptr), ctx.h)
        return path_d, final_d, status_d

    def close(self):
        if getattr(self, "_plan", None) is not None:
            self.ctx.lib.srx_align_plan_destroy(self._plan)
            self._plan = None


# ----------------------------------------------------------------------------- lexicon HMM / decoding
def build_lexicon_hmm(word_phones, uni_logp, lmsf=5.0):
    """hmm::hmm(pronlex&, unigram_model&, ...)  src/hmm.cpp:92-129 -> dict(n_states, tr_off, tr_tgt, tr_logp, ctx, word_of).
    word_phones: one int array per word, words in the order of the reference's map (sorted by spelling)."""
    lib = _capi.load()
    n = len(word_phones)
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum([len(p) for p in word_phones])
    ph = np.ascontiguousarray(np.concatenate(word_phones) if n else np.zeros(1), np.int32)
    lp = np.ascontiguousarray(uni_logp, np.float32)
    ns, nt = C.c_int64(0), C.c_int64(0)
    check(lib.srx_hmm_build_lexicon(n, _ptr(off), _ptr(ph), _ptr(lp), lmsf, None, None, None, None, None, C.byref(ns), C.byref(nt)))
    tr_off = np.zeros(ns.value + 1, np.int32)
    tr_tgt = np.zeros(nt.value, np.int32)
    tr_logp = np.zeros(nt.value, np.float32)
    ctx3 = np.zeros((ns.value, 3), np.int32)
    word_of = np.zeros(ns.value, np.int32)
    check(lib.srx_hmm_build_lexicon(n, _ptr(off), _ptr(ph), _ptr(lp), lmsf, _ptr(tr_off), _ptr(tr_tgt), _ptr(tr_logp), _ptr(ctx3),
                                    _ptr(word_of), C.byref(ns), C.byref(nt)))
    return dict(n_states=int(ns.value), tr_off=tr_off, tr_tgt=tr_tgt, tr_logp=tr_logp, ctx=ctx3, word_of=word_of)


class GraphHmm:
    """An HMM with an arbitrary transition list on the device (srx_decode_plan) and hmm::viterbi on it."""

    def __init__(self, ctx, tr_off, tr_tgt, tr_logp, tied):
        self.ctx = ctx
        self.n_states = len(tr_off) - 1
        self.tr_off = np.ascontiguousarray(tr_off, np.int32)
        self.tr_tgt = np.ascontiguousarray(tr_tgt, np.int32)
        h = C.c_void_p()
        check(ctx.lib.srx_decode_plan_create(ctx.h, self.n_states, _ptr(self.tr_off), _ptr(self.tr_tgt),
                                             _ptr(np.ascontiguousarray(tr_logp, np.float32)), _ptr(np.ascontiguousarray(tied, np.int32)),
                                             C.byref(h)), ctx.h)
        self.h = h

    def set_transitions(self, tr_logp):
        check(self.ctx.lib.srx_decode_plan_set_transitions(self.h, _ptr(np.ascontiguousarray(tr_logp, np.float32))), self.ctx.h)

    def set_sweep(self, sweep, max_workspace_bytes=0):
        """DECODE_AUTO / DECODE_DENSE / DECODE_FRONTIER (srx_decode_plan_set_sweep): same results, different walk."""
        check(self.ctx.lib.srx_decode_plan_set_sweep(self.h, int(sweep), int(max_workspace_bytes)), self.ctx.h)

    def viterbi(self, emis_d, stride_t, stride_g, frame_off, beam=1e10, final_mode=0):
        """-> (path [frames], final [n_utt], final_state [n_utt], status [n_utt]) as host arrays."""
        ctx = self.ctx
        frame_off = np.ascontiguousarray(frame_off, np.int64)
        n = len(frame_off) - 1
        path_d = ctx.zeros((max(int(frame_off[-1]), 1),), np.int32)
        final_d, fs_d, st_d = ctx.zeros((max(n, 1),), np.float32), ctx.zeros((max(n, 1),), np.int32), ctx.zeros((max(n, 1),), np.int32)
        check(ctx.lib.srx_viterbi_graph_dev(ctx.h, self.h, emis_d.ptr, stride_t, stride_g, _ptr(frame_off), n, beam, final_mode,
                                            path_d.ptr, final_d.ptr, fs_d.ptr, st_d.ptr), ctx.h)
        ctx.sync()
        return path_d.get(), final_d.get(), fs_d.get(), st_d.get()

    def close(self):
        if self.h:
            self.ctx.lib.srx_decode_plan_destroy(self.h)
            self.h = None


# ----------------------------------------------------------------------------- EM statistics
def em_statistics(ctx, A, B, feats_d, ld, gid_d, T, blend_keep=0.2, allreduce_max=None, allreduce_sum=None):
    """The statistics half of one hmm::train iteration (src/hmm.cpp:274-341): rebuilds model B from the
    alignment gid under model A.  `allreduce_*` are optional callables taking a DevArray (multi-GPU
    exchange points); None on a single GPU."""
    lib = ctx.lib
    G, M, D = A.G, A.M, A.D
    logp = ctx.empty((max(T * M, 1),), np.float32)
    shift = ctx.empty((G * M,), np.float32)
    stats1 = ctx.empty((G * M * (1 + D),), np.float64)
    stats2 = ctx.empty((G * M * D,), np.float64)
    check(lib.srx_em_shift_dev(ctx.h, A.h, feats_d.ptr, ld, gid_d.ptr, T, logp.ptr, shift.ptr), ctx.h)
    if allreduce_max:
        allreduce_max(shift)
    check(lib.srx_em_pass1_dev(ctx.h, A.h, feats_d.ptr, ld, gid_d.ptr, T, logp.ptr, shift.ptr, stats1.ptr), ctx.h)
    if allreduce_sum:
        allreduce_sum(stats1)
    check(lib.srx_em_means_dev(ctx.h, B.h, stats1.ptr, shift.ptr), ctx.h)
    check(lib.srx_em_pass2_dev(ctx.h, A.h, B.h, feats_d.ptr, ld, gid_d.ptr, T, logp.ptr, shift.ptr, stats2.ptr), ctx.h)
    if allreduce_sum:
        allreduce_sum(stats2)
    check(lib.srx_em_finish_dev(ctx.h, A.h, B.h, stats1.ptr, stats2.ptr, blend_keep), ctx.h)
    ctx.sync()
    return stats1, stats2, shift


def em_statistics_fused(ctx, A, B, feats_d, ld, gid_d, T, blend_keep=0.2, given_logp=False, allreduce_max=None, allreduce_sum=None):
    """The same statistics in one pass on the f64 matrix cores and one exchange (srx_em_fused_dev / srx_em_fused_finish_dev).
    given_logp: zeta from the rounding-faithful log-densities of srx_em_shift_dev (only the variance pass is folded in);
    otherwise the log-densities come off the matrix cores too, relative to each component's peak.  Returns
    (stats12 DevArray, number of components lost to f64 underflow: non-zero asks for the five-step form)."""
    lib = ctx.lib
    G, M = A.G, A.M
    n = int(lib.srx_em_fused_len(A.h))
    assert n > 0, "shape outside the fused statistics path"
    stats = ctx.empty((n,), np.float64)
    under = ctx.zeros((1,), np.int32)
    logp = shift = None
    if given_logp:
        logp = ctx.empty((max(T * M, 1),), np.float32)
        shift = ctx.empty((G * M,), np.float32)
        check(lib.srx_em_shift_dev(ctx.h, A.h, feats_d.ptr, ld, gid_d.ptr, T, logp.ptr, shift.ptr), ctx.h)
        if allreduce_max:
            allreduce_max(shift)
    check(lib.srx_em_fused_dev(ctx.h, A.h, feats_d.ptr, ld, gid_d.ptr, T, logp.ptr if logp else None, shift.ptr if shift else None,
                               stats.ptr), ctx.h)
    if allreduce_sum:
        allreduce_sum(stats)
    check(lib.srx_em_fused_finish_dev(ctx.h, A.h, B.h, stats.ptr, shift.ptr if shift else None, blend_keep, under.ptr), ctx.h)
    ctx.sync()
    return stats, int(under.get()[0])


def gather_gid(ctx, hmm, path_d, status_d):
    gid = ctx.empty((max(hmm.total_frames, 1),), np.int32)
    check(ctx.lib.srx_em_gather_gid_dev(ctx.h, C.byref(hmm.c), path_d.ptr, status_d.ptr, hmm.total_frames, gid.ptr), ctx.h)
    return gid


# ----------------------------------------------------------------------------- multi-GPU exchange
class Comm:
    """srx_comm: the all-reduces of sufficient statistics behind the C ABI (RCCL; world 1 = no-ops)."""

    def __init__(self, ctx, rank=0, world=1, uid=None, shm_name=None):
        self.ctx = ctx
        h = C.c_void_p()
        if shm_name is not None:
            check(ctx.lib.srx_comm_create_host(ctx.h, rank, world, shm_name.encode(), C.byref(h)), ctx.h)
        else:
            check(ctx.lib.srx_comm_create(ctx.h, rank, world, uid, C.byref(h)), ctx.h)
        self.h = h

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(_capi.load().srx_comm_unique_id(buf))
        return buf

    def allreduce_max_f32(self, a):
        check(self.ctx.lib.srx_allreduce_max_f32_dev(self.h, a.ptr, int(np.prod(a.shape))), self.ctx.h)

    def allreduce_sum_f64(self, a):
        check(self.ctx.lib.srx_allreduce_sum_f64_dev(self.h, a.ptr, int(np.prod(a.shape))), self.ctx.h)

    def allreduce_sum_i32(self, a):
        check(self.ctx.lib.srx_allreduce_sum_i32_dev(self.h, a.ptr, int(np.prod(a.shape))), self.ctx.h)

    def bytes_reduced(self):
        return int(self.ctx.lib.srx_comm_bytes_reduced(self.h))

    def close(self):
        if self.h:
            self.ctx.lib.srx_comm_destroy(self.h)
            self.h = None
