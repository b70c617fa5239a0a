#!/bin/bash
# ASan + UBSan over the host-only sources of libtracs_hip.so (CPU build).  usage: bash scripts/sanitize_host.sh
set -e
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/tracs_san
mkdir -p $OUT
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude \
    scripts/san_driver.cpp tracs_amd/csrc/fasta.cpp tracs_amd/csrc/alignio.cpp -lz -lpthread -o $OUT/san_driver
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 $OUT/san_driver $OUT
# second pass: ThreadSanitizer over the same driver (parallel readers, writers)
g++ -std=c++17 -O1 -g -fsanitize=thread -Iinclude \
    scripts/san_driver.cpp tracs_amd/csrc/fasta.cpp tracs_amd/csrc/alignio.cpp -lz -lpthread -o $OUT/tsan_driver
TSAN_OPTIONS=halt_on_error=1 $OUT/tsan_driver $OUT
