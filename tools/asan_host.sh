#!/bin/bash
# Sanitizer build of the host-side integer association logic (SciPy-compatible rectangular LSAP, thresholded matching,
# matching cascade): the code track ids depend on.  CPU only (GPU AddressSanitizer is not available on the pool).
#   tools/asan_host.sh            build ai-camera_amd/csrc/build/libaicam_host_asan.so
#   tools/asan_host.sh --test     ... and run tests/asan_driver.py under it (also done by tests/test_host_asan.py)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=$R/ai-camera_amd/csrc/build
mkdir -p $B
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -fPIC -shared \
    $R/ai-camera_amd/csrc/lsap.cpp $R/ai-camera_amd/csrc/assoc_host.cpp $R/ai-camera_amd/csrc/global_id.cpp $R/ai-camera_amd/csrc/host_stub.cpp -o $B/libaicam_host_asan.so
echo $B/libaicam_host_asan.so
if [ "$1" = "--test" ]; then
    LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
        python3 $R/tests/asan_driver.py $B/libaicam_host_asan.so
fi
