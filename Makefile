# Build everything for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
HIPCC   ?= hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-result
PKG      = n-body_amd
CSRC     = $(PKG)/csrc
LIB      = $(PKG)/lib/libnbody_amd.so
KERNEL_HDRS = $(wildcard $(CSRC)/*.hip.h) include/nbody.h

all: lib host oracle

lib: $(LIB)

$(LIB): $(CSRC)/nbody_capi.hip $(KERNEL_HDRS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) $(HIPFLAGS) -shared $< -o $@ -ldl

# developer build: bh_walk_quad_kernel returns per-phase ticks (tools/quad_timing.py)
lib-timing: $(CSRC)/nbody_capi.hip $(KERNEL_HDRS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) $(HIPFLAGS) -DNBODY_WIDE_TIMING -shared $< -o $(PKG)/lib/libnbody_amd_timing.so -ldl

# C++ host drivers: counterparts of the reference's benchmark.cpp / main.cpp loops, calling the C-ABI.
host: $(PKG)/bin/nbody_benchmark $(PKG)/bin/nbody_sim $(PKG)/bin/exact_core_check

# GPU self-check of csrc/exact_core.hip.h against hipcc's IEEE sqrt / division (tests/test_gpu_exact_core.py)
$(PKG)/bin/exact_core_check: tools/exact_core_check.hip $(CSRC)/exact_core.hip.h
	@mkdir -p $(PKG)/bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -Wno-format $< -o $@

$(PKG)/bin/nbody_benchmark: $(PKG)/host/benchmark_main.cpp $(PKG)/host/config_reader.hpp $(PKG)/host/initial_conditions.hpp include/nbody.h $(LIB)
	@mkdir -p $(PKG)/bin
	g++ -O2 -std=c++17 -Iinclude $< -o $@ -L$(PKG)/lib -lnbody_amd -Wl,-rpath,'$$ORIGIN/../lib'

$(PKG)/bin/nbody_sim: $(PKG)/host/sim_main.cpp $(PKG)/host/config_reader.hpp $(PKG)/host/initial_conditions.hpp include/nbody.h $(LIB)
	@mkdir -p $(PKG)/bin
	g++ -O2 -std=c++17 -Iinclude $< -o $@ -L$(PKG)/lib -lnbody_amd -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle all

tools: tools/ubench tools/kbench2

tools/ubench: tools/ubench.hip
	$(HIPCC) --offload-arch=$(ARCH) -O2 $< -o $@

tools/kbench2: tools/kbench2.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 $< -o $@

clean:
	rm -rf $(PKG)/lib $(PKG)/bin tools/ubench tools/kbench2
	$(MAKE) -C oracle clean

.PHONY: all lib lib-timing host oracle tools clean
