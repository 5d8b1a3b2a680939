"""Every entry point of include/srx.h called with null pointers and zero sizes: a status-returning one must come back
with an error code (or SRX_OK where nothing-to-do is a legitimate request), a destroyer must do nothing, a getter must
return null / a negative count -- and none may fault.  A null context is refused before any HIP call, so this runs without
a GPU.  The calls are made in a child process so that a fault shows as a failed test, not a dead test run."""
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
from speechrecognition_amd import _capi
lib = _capi.load()
OK_WITH_NOTHING = set()          # (none today: every all-null call is an error)
for name, (res, args) in sorted(_capi.SIGNATURES.items()):
    fn = getattr(lib, name)
    zeros = []
    for a in args:
        if a in (C.c_int, C.c_int64, C.c_size_t):
            zeros.append(0)
        elif a is C.c_float:
            zeros.append(0.0)
        else:
            zeros.append(None)
    r = fn(*zeros)
    if name in ("srx_version",):
        assert r, name
    elif res is C.c_int and name not in ("srx_comm_rank", "srx_comm_world", "srx_mfcc_plan_bin_max"):
        assert r < 0 or name in OK_WITH_NOTHING, (name, r)
    elif res is None:
        pass
    elif res in (C.c_void_p, C.c_char_p):
        assert name == "srx_last_error" or not r, (name, r)
    print("ok", name, r if not isinstance(r, bytes) else r[:40], flush=True)
print("all ok")
'''


def test_every_entry_point_survives_null_arguments():
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], text=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.strip().endswith("all ok"), r.stdout[-3000:]
