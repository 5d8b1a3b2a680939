"""The CPU-only paths of the C++ host layer (text front end, WAV I/O, topology builders, unigram table, sharding) under
AddressSanitizer + UndefinedBehaviorSanitizer: tests/cpp/host_cpu_test.cpp, built by `make -C speechrecognition_amd/host`
(__graft_entry__.build()).  Host code only -- no GPU is touched, so it runs here.  Leak checking is off: the reference's API
hands out raw owning pointers that its callers never free (read_wav's ClipArrayBuffer, src/wav.cpp:82; mel_filterbank's
filterbank, src/mfcc.cpp:190), and the mirror keeps those signatures."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "speechrecognition_amd", "lib", "srx_host_cpu_test_san")


def test_host_layer_cpu_paths_under_asan_ubsan(tmp_path):
    if not os.path.exists(BIN):
        subprocess.run(["make", "-C", os.path.join(ROOT, "speechrecognition_amd", "host"), "sanitize"], check=True,
                       stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([BIN, os.path.join(ROOT, "tests", "golden", "lexicon"), str(tmp_path)], env=env, text=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout, r.stdout[-4000:]
    assert r.stdout.strip().endswith("all ok"), r.stdout[-2000:]
