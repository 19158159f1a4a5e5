#!/bin/bash
# AddressSanitizer on HOST code only (the pool runs no GPU sanitizers): the library's host side and the
# host mirror are built with -fsanitize=address (hipcc: after -Xarch_host, so that device code is left
# alone) into search_engine_amd/csrc/build_asan, and stress_mirror runs against that build.
#   tools/asan_stress.sh build     here (no GPU needed)
#   tools/asan_stress.sh run       on the GPU box
set -e
cd "$(dirname "$0")/../search_engine_amd/csrc"
case $1 in
build)
  mkdir -p build_asan
  for f in vs_kernels vs_batch vs_runtime vs_group; do
    hipcc --offload-arch=gfx950 -O2 -g -std=c++17 -fPIC -Wno-unused-function -ffp-contract=off -Xarch_host -fsanitize=address -c $f.hip -o build_asan/$f.o
  done
  for f in vs_roaring vs_ingest; do
    hipcc -x hip --offload-arch=gfx950 -O2 -g -std=c++17 -fPIC -ffp-contract=off -Xarch_host -fsanitize=address -c $f.cpp -o build_asan/$f.o
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -Xarch_host -fsanitize=address -shared-libsan -o build_asan/libvs_amd.so build_asan/*.o -ldl
  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -fsanitize=address -shared-libsan -mpopcnt -std=c++17 -I../../include stress_mirror.cpp host_engine.cpp \
      -o build_asan/stress_mirror -Lbuild_asan -lvs_amd -pthread -Wl,-rpath,'$ORIGIN'
  ;;
run)
  rt=$(dirname "$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)")
  LD_LIBRARY_PATH=$rt ASAN_OPTIONS=verify_asan_link_order=0:protect_shadow_gap=0:detect_leaks=0 \
      timeout -k 10 400 ./build_asan/stress_mirror 6 20
  ;;
*) echo "build | run"; exit 2 ;;
esac
