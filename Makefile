# Builds libgulon_hip.so (gfx950 only) and the CPU oracle.
# -ffp-contract=off: every kernel reproduces the JVM's unfused binary32 arithmetic;
# the only fused arithmetic in the library is the explicit MFMA in the k-means filter.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC = gulon_amd/csrc
OBJDIR = build/obj
LIB = gulon_amd/lib/libgulon_hip.so
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
           -fvisibility=hidden -Wall -Wno-unused-function -Iinclude
SRCS = $(CSRC)/api_core.hip $(CSRC)/scan.hip $(CSRC)/knn.hip $(CSRC)/kmeans.hip $(CSRC)/kmeans_stream.hip $(CSRC)/kmeans_mfma.hip $(CSRC)/replay.hip $(CSRC)/filter.hip $(CSRC)/grouped.hip $(CSRC)/grouped_filter.hip $(CSRC)/wide.hip $(CSRC)/wide_filter.hip $(CSRC)/conflict_order.hip $(CSRC)/sharded.hip $(CSRC)/literal.hip
OBJS = $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
HDRS = $(CSRC)/common.hpp $(CSRC)/scan.hpp $(CSRC)/kmeans.hpp include/gulon_hip.h $(CSRC)/grouped_filter.hpp
# The product library carries no test code.  The self-tests of three kernels (gulon_selftest_*) and the measured-and-
# dropped fused k-means update (kmeans_fused.hip) live in a second library that only tests/ loads: the same objects,
# with the four files that have hooks compiled again under -DGULON_TEST_HOOKS.
HOOKLIB = gulon_amd/lib/libgulon_hip_testhooks.so
HOOKED = kmeans kmeans_stream kmeans_mfma conflict_order kmeans_fused
HOOKOBJS = $(patsubst %,$(OBJDIR)/%.hooks.o,$(HOOKED)) $(filter-out $(patsubst %,$(OBJDIR)/%.o,$(HOOKED)),$(OBJS))

all: $(LIB) $(HOOKLIB) oracle build/test_host_api

# MFMA results are consumed by VALU/permlane code: keep accumulators in VGPRs (no v_accvgpr moves)
$(OBJDIR)/kmeans_mfma.o $(OBJDIR)/kmeans_mfma.hooks.o: HIPFLAGS += -mllvm -amdgpu-mfma-vgpr-form

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJDIR)/%.hooks.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DGULON_TEST_HOOKS -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p gulon_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

$(HOOKLIB): $(HOOKOBJS)
	@mkdir -p gulon_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HOOKOBJS) -ldl

oracle:
	$(MAKE) -C oracle

# C++ host-API parity test (tests/cpp): gulon.hpp over the C ABI, checked against the oracle
build/test_host_api: tests/cpp/test_host_api.cpp include/gulon/gulon.hpp include/gulon_hip.h $(LIB) oracle
	@mkdir -p build
	g++ -O2 -std=c++17 -Iinclude tests/cpp/test_host_api.cpp -o $@ -Lgulon_amd/lib -lgulon_hip -Loracle/build -lgulon_oracle \
	    -Wl,-rpath,'$$ORIGIN/../gulon_amd/lib' -Wl,-rpath,'$$ORIGIN/../oracle/build'

clean:
	rm -rf build $(LIB) $(HOOKLIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
