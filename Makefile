# Top-level build: the gfx950 engine (libjfgpu.so), the host CLI, the oracle.
#   make            -> engine + CLI + oracle
#   make engine     -> jellyfish_amd/lib/libjfgpu.so   (hipcc cross-compiles without a GPU)
HIPCC    ?= hipcc
CXX      ?= g++
ARCH     ?= gfx950
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result
CSRC     := jellyfish_amd/csrc
LIBDIR   := jellyfish_amd/lib

all: engine cli oracle

engine: $(LIBDIR)/libjfgpu.so

$(LIBDIR)/libjfgpu.so: $(wildcard $(CSRC)/*.hip) $(wildcard $(CSRC)/*.hpp) $(wildcard $(CSRC)/*.inl) include/jfgpu.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/jfgpu.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib

cli: bin/jellyfish-amd

bin/jellyfish-amd: $(wildcard jellyfish_amd/cli/*.cc) $(wildcard jellyfish_amd/include/jellyfish_amd/*.hpp) include/jfgpu.h $(LIBDIR)/libjfgpu.so
	@mkdir -p bin
	@if [ -f jellyfish_amd/cli/jellyfish_amd.cc ]; then \
	  $(CXX) -O2 -std=c++17 -Wall -Iinclude -Ijellyfish_amd/include -o $@ jellyfish_amd/cli/jellyfish_amd.cc -L$(LIBDIR) -ljfgpu -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -pthread; \
	fi

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf $(LIBDIR) bin oracle/_build
.PHONY: all engine cli oracle clean
