"""speechrecognition_amd: MI355X (gfx950) implementation of the acoustic hot path of
albinoBandicoot/speechRecognition -- MFCC extraction, diagonal-GMM emission scoring, Viterbi forced
alignment and embedded-training statistics -- behind a C ABI (include/srx.h, lib/libsrx.so).

The compute lives in hand-written HIP kernels (csrc/); this package is the Python host layer used by
the tests, the benchmark and the multi-GPU driver.  There is no CPU fallback."""
from . import _capi  # noqa: F401
from ._capi import (SRX_MIX_COMPONENT0, SRX_MIX_LOGSUMEXP, SRX_SCORE_EXACT, SRX_SCORE_MFMA, SrxError)  # noqa: F401

__all__ = ["_capi", "host", "synth", "SrxError", "SRX_MIX_COMPONENT0", "SRX_MIX_LOGSUMEXP", "SRX_SCORE_EXACT",
           "SRX_SCORE_MFMA"]
