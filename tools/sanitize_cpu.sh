#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU builds (GPU sanitizers are not available on this pool): the device
# headers compiled for the host (tests/hostsim), the oracle, and the product's own host side (capi_host.cpp: scene builder, BIH /
# Mesh builders, flattener, show-format and NFF loaders -- linked with the regular device object into build_old/libglome_san.so
# and loaded through GLOME_DEBUG_LIB), each rebuilt instrumented, their test files run, the regular builds put back.
# usage: tools/sanitize_cpu.sh   (after __graft_entry__.build())
set -e
root=$(cd "$(dirname "$0")/.." && pwd); cd "$root"
pre=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)
san="-O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined"
cp tests/hostsim/libhostsim.so /tmp/libhostsim_keep.so; cp oracle/liboracle.so /tmp/liboracle_keep.so
trap 'cp /tmp/libhostsim_keep.so tests/hostsim/libhostsim.so; cp /tmp/liboracle_keep.so oracle/liboracle.so' EXIT
(cd tests/hostsim && g++ $san -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-but-set-variable -I. -shared -o libhostsim.so hostsim.cpp)
(cd oracle && g++ $san -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -pthread -shared -o liboracle.so oracle_capi.cpp)
mkdir -p build_old
g++ $san -std=c++17 -fPIC -Wall -Iinclude -c glome_amd/csrc/capi_host.cpp -o /tmp/capi_host_san.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_old/libglome_san.so /tmp/capi_host_san.o glome_amd/csrc/obj/glome_device_p*.o
GLOME_DEBUG_LIB=build_old/libglome_san.so LD_PRELOAD=$pre ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_host_builder.py tests/test_show_format.py tests/test_nff.py -x -q
LD_PRELOAD=$pre ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_hostsim_parity.py tests/test_golden.py tests/test_oracle_kat.py tests/test_np_crosscheck.py tests/test_nff.py tests/test_show_format.py -x -q
