#!/bin/bash
# TEST INFRASTRUCTURE.  Builds, into build/hostlogic/, the BGZF front-end on the zlib test double
# (tests/native/fake_engine.c) and the reference's own test/test_bgzf.c + bgzip.c on top of it, so that the
# front-end's host logic can be checked without a GPU.  usage: build_hostlogic.sh [sanitizer]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); R=${REF:-/root/reference}; O=$ROOT/oracle; OUT=$ROOT/build/hostlogic
SAN=${1:-}; SUF=${SAN:+_$SAN}; SFLAG=${SAN:+-fsanitize=$SAN}
mkdir -p "$OUT"; cd "$OUT"
[ -f $O/_ref/config.h ] || make -C $O _ref/config.h >/dev/null
gcc -O1 -g $SFLAG -fPIC -c -I$ROOT/include $ROOT/tests/native/fake_engine.c -o fake_engine$SUF.o
g++ -O1 -g $SFLAG -std=c++17 -fPIC -shared -Wall -I$ROOT/include $ROOT/htslib_amd/csrc/bgzf_front.cpp $ROOT/htslib_amd/csrc/hfile_min.cpp \
    fake_engine$SUF.o -o libhts_bgzf_fake$SUF.so -lz -lpthread
for prog in test/test_bgzf.c bgzip.c; do
  gcc -O1 -g $SFLAG -o $(basename $prog .c)_fake$SUF -I$O/_ref -I$R $R/$prog $O/ref_stubs.c -rdynamic -L. -lhts_bgzf_fake$SUF \
      -Wl,-rpath,'$ORIGIN' -lpthread -lm
done
