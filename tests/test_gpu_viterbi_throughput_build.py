"""k_viterbi exists in two builds (srx_viterbi.hip, VitCfg): the latency build the small batches of the other tests get, and
the throughput build (four waves per SIMD, 128 registers, a shallower back-pointer staging buffer) that batches of 8,192
utterances and more get.  The switch is read once per process, so the bit-exactness tests of tests/test_gpu_viterbi.py and
the alignment-mode tests are run again here in a child process with the throughput build forced for every batch size."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bit_exactness_suite_on_the_throughput_build():
    env = dict(os.environ, SRX_VIT_OCC_MIN_UTT="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "--timeout", "120", "--timeout-method", "thread",
                        os.path.join(ROOT, "tests", "test_gpu_viterbi.py"), os.path.join(ROOT, "tests", "test_gpu_align_mode.py"),
                        os.path.join(ROOT, "tests", "test_gpu_edge_cases.py")],
                       env=env, cwd=ROOT, text=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]
