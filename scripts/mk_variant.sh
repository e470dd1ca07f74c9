#!/bin/bash
# scripts/mk_variant.sh NAME SRC.hip [-Dflags...]: libhtsgpu.so with ONE object rebuilt with extra flags -> variants/NAME.so
set -e
name=$1; src=$2; shift 2
o=build/obj/variant_$name.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ihtslib_amd/csrc -Wno-unused-function "$@" -c htslib_amd/csrc/$src.hip -o $o
objs=$(ls build/obj/*.o | grep -v variant_ | grep -v "/$src.o")
hipcc -fPIC --offload-arch=gfx950 -shared $objs $o -o variants/$name.so
