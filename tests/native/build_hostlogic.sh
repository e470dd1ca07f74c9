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
# The same front-end inside the reference's WHOLE libhts (oracle/_ref/hts_obj/*.o minus bgzf.o; CRAM block layer and codecs stay the reference's +
# oracle/htscodecs_stub): test/test_view.c and test/test_index.c on it check bam_read1 / bam_write1 / the on-the-fly indexes over our bgzf_* host logic
# without a GPU (tests/test_front_host_logic.py); the GPU twin is oracle/_ref/libhts_gpu.so (tests/test_libhts_gpu.py).
if [ -d $O/_ref/hts_obj ] && [ -z "$SAN" ]; then
  OBJS=$(ls $O/_ref/hts_obj/*.o | grep -v '/bgzf.o$')
  g++ -O1 -g -std=c++17 -fPIC -Wall -Wno-sign-compare -I$ROOT/include -c $ROOT/htslib_amd/csrc/bgzf_front.cpp -o bgzf_front_libhts.o
  gcc -O1 -fPIC -c -I$O/htscodecs_stub $O/htscodecs_stub/htscodecs_stub.c -o stub_libhts.o
  g++ -shared -o libhts_fake.so $OBJS bgzf_front_libhts.o fake_engine.o stub_libhts.o $O/liboracle.so -Wl,-rpath,$O \
      $O/_ref/ld/libdeflate.so.0 /usr/lib/x86_64-linux-gnu/libbz2.so.1.0 /usr/lib/x86_64-linux-gnu/liblzma.so.5 -Wl,-rpath,$O/_ref/ld -lz -lpthread -lm
  for prog in test_view test_index; do
    gcc -O1 -g -o ${prog}_fake -I$O/_ref -I$R $R/test/$prog.c -L. -lhts_fake -Wl,-rpath,'$ORIGIN' -Wl,-rpath-link,$O/_ref/ld -lz -lpthread -lm
  done
fi
