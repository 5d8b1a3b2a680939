"""ctypes binding of libsrx.so (include/srx.h).  There is no fallback: if the library is missing
or does not load, importing a compute entry point raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsrx.so")

SRX_OK = 0
SRX_ERR_INVALID = -1
SRX_ERR_DEVICE = -2
SRX_ERR_NOMEM = -3
SRX_ERR_IO = -4
SRX_ERR_UNSUPPORTED = -5

SRX_MIX_COMPONENT0 = 0
SRX_MIX_LOGSUMEXP = 1
SRX_SCORE_MFMA = 0
SRX_SCORE_EXACT = 1
SRX_SCORE_MFMA_F16X3 = 2
SRX_COMM_ID_BYTES = 128


class SrxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"srx error {code}: {msg}")
        self.code = code


class srx_filter(C.Structure):
    _fields_ = [("center", C.c_float), ("width", C.c_float), ("height", C.c_float)]


class srx_mfcc_cfg(C.Structure):
    _fields_ = [("window_len", C.c_int), ("frame_shift", C.c_int), ("n_ceps", C.c_int), ("normalize", C.c_int),
                ("target_rms", C.c_float), ("preemphasis", C.c_int), ("alpha", C.c_float)]


class srx_hmm_batch(C.Structure):
    _fields_ = [("n_utt", C.c_int), ("frame_off_dev", C.c_void_p), ("state_off_dev", C.c_void_p),
                ("tied_dev", C.c_void_p), ("ntr_dev", C.c_void_p), ("trlogp_dev", C.c_void_p), ("max_states", C.c_int), ("total_frames", C.c_int64)]


vp = C.c_void_p
i64 = C.c_int64

# name -> (restype, argtypes); must list every symbol declared in include/srx.h
SIGNATURES = {
    "srx_ctx_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
    "srx_ctx_destroy": (None, [vp]),
    "srx_last_error": (C.c_char_p, [vp]),
    "srx_sync": (C.c_int, [vp]),
    "srx_ctx_stream": (vp, [vp]),
    "srx_version": (C.c_char_p, []),
    "srx_profile_enable": (C.c_int, [vp, C.c_int]),
    "srx_profile_query": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)]),
    "srx_malloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "srx_free": (C.c_int, [vp, vp]),
    "srx_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "srx_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "srx_memset": (C.c_int, [vp, vp, C.c_int, C.c_size_t]),
    "srx_filter_init": (C.c_int, [C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(srx_filter)]),
    "srx_filter_value": (C.c_float, [C.POINTER(srx_filter), C.c_int]),
    "srx_mel_filterbank": (C.c_int, [C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(srx_filter), C.POINTER(C.c_int)]),
    "srx_mfcc_plan_create": (C.c_int, [vp, C.POINTER(srx_mfcc_cfg), C.POINTER(srx_filter), C.c_int, C.POINTER(vp)]),
    "srx_mfcc_plan_destroy": (None, [vp]),
    "srx_mfcc_plan_bin_max": (C.c_int, [vp]),
    "srx_mfcc_plan_set_path": (C.c_int, [vp, C.c_int]),
    "srx_mfcc_frame_offsets": (C.c_int, [C.POINTER(srx_mfcc_cfg), vp, C.c_int, vp]),
    "srx_mfcc_batch_dev": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, i64, vp, vp]),
    "srx_mfcc_batch": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp]),
    "srx_partial_dft_batch_dev": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, i64, vp, vp]),
    "srx_partial_dft_batch": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp]),
    "srx_partial_dft_frames_dev": (C.c_int, [vp, vp, i64, vp, C.c_int, C.c_int, i64, C.c_int, vp, i64]),
    "srx_partial_dft_frames": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, i64, C.c_int, vp]),
    "srx_filterbank_apply_dev": (C.c_int, [vp, C.POINTER(srx_filter), C.c_int, vp, i64, C.c_int, i64, vp, i64]),
    "srx_filterbank_apply": (C.c_int, [vp, C.POINTER(srx_filter), C.c_int, vp, C.c_int, i64, vp]),
    "srx_log1p_dev": (C.c_int, [vp, vp, i64]),
    "srx_log1p": (C.c_int, [vp, vp, i64]),
    "srx_dct_dev": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, i64, vp, i64]),
    "srx_dct": (C.c_int, [vp, vp, C.c_int, C.c_int, i64, vp]),
    "srx_delta": (C.c_int, [vp, vp, vp, C.c_int, vp, i64]),
    "srx_gaussian_logprob": (C.c_int, [vp, vp, C.c_int, i64, C.c_int, vp, vp, C.c_float, vp]),
    "srx_normalize_volume_dev": (C.c_int, [vp, vp, vp, C.c_int, C.c_float, vp]),
    "srx_scale_dev": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "srx_rms_amplitude_dev": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "srx_preemphasis_dev": (C.c_int, [vp, vp, vp, vp, C.c_int, i64, C.c_float]),
    "srx_features_batch_dev": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, i64, vp]),
    "srx_features_batch": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    "srx_corpus_mean_var_dev": (C.c_int, [vp, vp, C.c_int, i64, C.c_int, vp]),
    "srx_corpus_sums_dev": (C.c_int, [vp, vp, C.c_int, i64, C.c_int, vp, vp]),
    "srx_model_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "srx_model_destroy": (None, [vp]),
    "srx_model_set": (C.c_int, [vp, vp, vp, vp]),
    "srx_model_get": (C.c_int, [vp, vp, vp, vp]),
    "srx_model_copy": (C.c_int, [vp, vp]),
    "srx_model_version": (C.c_uint64, [vp]),
    "srx_gmm_score_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, i64, vp, i64, i64]),
    "srx_gmm_score": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, i64, vp]),
    "srx_hmm_build_training": (C.c_int, [vp, C.c_int, vp, vp, vp]),
    "srx_viterbi_batch_dev": (C.c_int, [vp, C.POINTER(srx_hmm_batch), vp, i64, i64, C.c_float, vp, vp, vp]),
    "srx_viterbi_batch_noalloc_dev": (C.c_int, [vp, C.POINTER(srx_hmm_batch), vp, i64, i64, C.c_float, vp, vp, vp]),
    "srx_align_plan_create": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(vp)]),
    "srx_align_plan_destroy": (None, [vp]),
    "srx_align_plan_emission_count": (C.c_int64, [vp]),
    "srx_align_plan_emission_offsets": (C.c_int, [vp, vp]),
    "srx_align_plan_emission_offsets_dev": (vp, [vp]),
    "srx_align_plan_cache_frames": (C.c_int, [vp, C.c_int]),
    "srx_gmm_score_align_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.POINTER(srx_hmm_batch), vp, vp]),
    "srx_viterbi_align_dev": (C.c_int, [vp, C.POINTER(srx_hmm_batch), vp, vp, C.c_float, vp, vp, vp]),
    "srx_hmm_build_lexicon": (C.c_int, [C.c_int, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, C.POINTER(i64), C.POINTER(i64)]),
    "srx_decode_plan_create": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, C.POINTER(vp)]),
    "srx_decode_plan_set_transitions": (C.c_int, [vp, vp]),
    "srx_decode_plan_destroy": (None, [vp]),
    "srx_decode_plan_set_sweep": (C.c_int, [vp, C.c_int, C.c_size_t]),
    "srx_viterbi_graph_dev": (C.c_int, [vp, vp, vp, i64, i64, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp, vp]),
    "srx_train_transitions_dev": (C.c_int, [vp, C.POINTER(srx_hmm_batch), vp, vp]),
    "srx_state_model_dev": (C.c_int, [vp, C.POINTER(srx_hmm_batch), vp, vp, vp, C.c_int, vp, vp]),
    "srx_em_gather_gid_dev": (C.c_int, [vp, C.POINTER(srx_hmm_batch), vp, vp, i64, vp]),
    "srx_em_shift_dev": (C.c_int, [vp, vp, vp, C.c_int, vp, i64, vp, vp]),
    "srx_em_pass1_dev": (C.c_int, [vp, vp, vp, C.c_int, vp, i64, vp, vp, vp]),
    "srx_em_means_dev": (C.c_int, [vp, vp, vp, vp]),
    "srx_em_pass2_dev": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, i64, vp, vp, vp]),
    "srx_em_finish_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_float]),
    "srx_em_fused_len": (i64, [vp]),
    "srx_em_fused_dev": (C.c_int, [vp, vp, vp, C.c_int, vp, i64, vp, vp, vp]),
    "srx_em_fused_finish_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_float, vp]),
    "srx_comm_unique_id": (C.c_int, [vp]),
    "srx_comm_create": (C.c_int, [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]),
    "srx_comm_create_host": (C.c_int, [vp, C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)]),
    "srx_comm_destroy": (None, [vp]),
    "srx_comm_rank": (C.c_int, [vp]),
    "srx_comm_world": (C.c_int, [vp]),
    "srx_comm_bytes_reduced": (i64, [vp]),
    "srx_allreduce_max_f32_dev": (C.c_int, [vp, vp, i64]),
    "srx_allreduce_sum_f64_dev": (C.c_int, [vp, vp, i64]),
    "srx_allreduce_sum_i32_dev": (C.c_int, [vp, vp, i64]),
}

_lib = None


def load():
    """Loads libsrx.so (needs libamdhip64; works without a GPU) and types every entry point."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C speechrecognition_amd/csrc).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, ctx=None):
    if rc != SRX_OK:
        msg = load().srx_last_error(ctx)
        raise SrxError(rc, msg.decode() if msg else "?")
