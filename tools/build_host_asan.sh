#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the whole stack (device code is NOT instrumented: the pool has no GPU
# sanitizer): libsrx.so, the headless driver and the C++ API test, into gpurun_tmp/san/.  Run the C++ GPU tests on it with
#   SRX_TEST_BIN_DIR=gpurun_tmp/san ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_gpu_host_cpp.py
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_tmp/san
mkdir -p $OUT/obj
C=speechrecognition_amd/csrc
H=speechrecognition_amd/host
for f in $C/srx_*.hip; do
    o=$OUT/obj/$(basename ${f%.hip}).o
    if [ ! -f $o ] || [ $f -nt $o ]; then
        /opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result \
            -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -c $f -o $o &
    fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
    -o $OUT/libsrx.so $OUT/obj/*.o -ldl -lpthread -lrt
CXX=/opt/rocm/lib/llvm/bin/clang++
FL="-O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unused-result"
$CXX $FL -o $OUT/srx $H/srx_main.cpp $H/srx_host.cpp $H/srx_text.cpp -L$OUT -lsrx -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib
$CXX $FL -o $OUT/srx_host_api_test tests/cpp/host_api_test.cpp $H/srx_host.cpp $H/srx_text.cpp -L$OUT -lsrx -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath-link,/opt/rocm/lib
ls -la $OUT
