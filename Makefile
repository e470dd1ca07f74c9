# Build the gfx950 block-codec engine (C-ABI shared library) and the test oracle.
HIPCC   ?= hipcc
ARCH    ?= gfx950
CSRC    := htslib_amd/csrc
HIPSRC  := $(wildcard $(CSRC)/*.hip)
HDRS    := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.inc) $(wildcard include/*.h)
LIB     := htslib_amd/libhtsgpu.so
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -I$(CSRC) -Wall -Wno-unused-function

FRONT   := htslib_amd/libhts_bgzf.so

all: $(LIB) $(FRONT) oracle

# htslib's BGZF front-end API (bgzf_open/read/write/...) over the engine: host C++ only
# (hfile_min.cpp = bundled local-file hFILE provider; a libhts build links hfile.c instead, see oracle/Makefile)
FRONTSRC := $(CSRC)/bgzf_front.cpp $(CSRC)/cram_block_front.cpp $(CSRC)/htscodecs_front.cpp $(CSRC)/hfile_min.cpp
$(FRONT): $(FRONTSRC) include/hts_bgzf_gpu.h include/hts_cram_gpu.h include/hts_hfile_abi.h include/htsgpu.h $(LIB)
	g++ -O2 -std=c++17 -fPIC -shared -Wall -Iinclude $(FRONTSRC) -o $@ -Lhtslib_amd -lhtsgpu -lpthread -ldl -Wl,-rpath,'$$ORIGIN'

# one object per source (make -j builds them in parallel; a kernel edit recompiles one file)
OBJDIR  := build/obj
HIPOBJ  := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(HIPSRC))
$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIB): $(HIPOBJ)
	$(HIPCC) $(HIPFLAGS) -shared $(HIPOBJ) -o $@

oracle:
	$(MAKE) -C oracle all

resource-usage:
	$(HIPCC) $(HIPFLAGS) -shared $(HIPSRC) -o /dev/null -Rpass-analysis=kernel-resource-usage

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean
.PHONY: all oracle clean resource-usage

# bring-up probe: separate library variant with in-kernel tracing
diag: tests/native/diag
tests/native/diag: tests/native/diag.cpp $(HIPSRC) $(HDRS) $(LIB)
	$(HIPCC) $(HIPFLAGS) -DHG_DEBUG_TRACE -shared $(HIPSRC) -o tests/native/libhtsgpu_trace.so
	$(HIPCC) -O2 --offload-arch=$(ARCH) -Iinclude tests/native/diag.cpp -o $@ -Ltests/native -lhtsgpu_trace -Wl,-rpath,'$$ORIGIN'
	$(HIPCC) -O2 --offload-arch=$(ARCH) -Iinclude tests/native/diag.cpp -o $@_plain -Lhtslib_amd -lhtsgpu -Wl,-rpath,'$$ORIGIN/../../htslib_amd'
.PHONY: diag

# kernel A/B timing tool (dlopens any number of library builds)
tests/native/kbench: tests/native/kbench.cpp include/htsgpu.h
	$(HIPCC) -O2 --offload-arch=$(ARCH) -Iinclude tests/native/kbench.cpp -o $@ -ldl
tests/native/latprobe: tests/native/latprobe.cpp include/htsgpu.h
	$(HIPCC) -O2 --offload-arch=$(ARCH) -Iinclude tests/native/latprobe.cpp -o $@ -ldl
kbench: tests/native/kbench tests/native/latprobe
.PHONY: kbench

# the fqzcomp decoder with per-phase tick counters (HG_FQZ_PROFILE): HTSGPU_LIB=tests/native/libhtsgpu_fqzprof.so python bench.py --op fqz ...
fqzprof: $(LIB)
	$(HIPCC) $(HIPFLAGS) -DHG_FQZ_PROFILE -c $(CSRC)/fqzcomp.hip -o $(OBJDIR)/fqzcomp_prof.o
	$(HIPCC) $(HIPFLAGS) -shared $(filter-out $(OBJDIR)/fqzcomp.o,$(HIPOBJ)) $(OBJDIR)/fqzcomp_prof.o -o tests/native/libhtsgpu_fqzprof.so
.PHONY: fqzprof
