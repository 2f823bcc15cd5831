# Top-level helper targets (the library itself is built by `python -m imageexperiments_amd.build`, hipcc for gfx950).
#
#   make asan        host code of the product + the oracle under AddressSanitizer and UBSan (g++, CPU only):
#                      1. tests/cpp/asan_host: the product's host sources (entropy stage, container parser, dictionary, statistics)
#                         in one translation unit, driven through round trips and a corpus of truncated / bit-flipped containers
#                      2. the oracle built with the sanitizers, its golden-fixture tests run with libasan preloaded
#   make asan-host   step 1 only (what tests/test_asan_host.py runs)
#
# GPU AddressSanitizer is not available on the test pool; the kernels are covered by the parity suite instead.
SAN = -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer
GOLDEN_MN = tests/golden/r0c1de5e1t_3_5.mn

tests/cpp/asan_host_bin: tests/cpp/asan_host.cpp $(wildcard imageexperiments_amd/csrc/host_*.cpp imageexperiments_amd/csrc/host_*.h)
	g++ -std=c++17 -O1 -g $(SAN) -ffp-contract=off -pthread -Wall -Wno-unused-function tests/cpp/asan_host.cpp -o $@

asan-host: tests/cpp/asan_host_bin
	ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 MPC_HOST_THREADS=4 ./tests/cpp/asan_host_bin $(GOLDEN_MN)

oracle/_build/liboracle_asan.so: $(wildcard oracle/*.c oracle/*.h)
	mkdir -p oracle/_build
	gcc -std=c11 -O1 -g $(SAN) -ffp-contract=off -fPIC -shared -o $@ oracle/mpo_*.c -lm

asan-oracle: oracle/_build/liboracle_asan.so
	LD_PRELOAD=$$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 ORACLE_LIB=$(CURDIR)/oracle/_build/liboracle_asan.so \
	    python -m pytest tests/test_oracle_golden.py -x -q -p no:cacheprovider

asan: asan-host asan-oracle

.PHONY: asan asan-host asan-oracle
