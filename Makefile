# Build recipe: the HIP library (product) and the CPU checker (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950

LIB := ray_tracing_amd/libmrt.so
ORACLE := oracle/libmrt_oracle.so
CSRC := ray_tracing_amd/csrc

# -disable-machine-licm: the segment kernels are one long loop; hoisting constants and kernel arguments out of it
# (MachineLICM) keeps them in registers across the whole loop, which the allocator then spills to scratch and VGPR lanes
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -disable-machine-licm -Wall -Wno-unused-function
CXXFLAGS := -O3 -std=c++17 -fPIC -Wall -Wextra

all: lib oracle

lib: $(LIB)

$(LIB): $(CSRC)/mrt_device.hip $(CSRC)/mrt_post.hip $(CSRC)/mrt_host.cpp $(CSRC)/mrt_octree.cpp $(CSRC)/mrt_internal.h include/mrt.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/mrt_device.hip $(CSRC)/mrt_post.hip $(CSRC)/mrt_host.cpp $(CSRC)/mrt_octree.cpp

# Diagnostic build: in-kernel cycle stamps per region of the segment kernel (never benchmarked).
stamps: ray_tracing_amd/libmrt_stamps.so
ray_tracing_amd/libmrt_stamps.so: $(CSRC)/mrt_device.hip $(CSRC)/mrt_post.hip $(CSRC)/mrt_host.cpp $(CSRC)/mrt_octree.cpp $(CSRC)/mrt_internal.h include/mrt.h
	$(HIPCC) $(HIPFLAGS) -DMRT_STAMPS -shared -o $@ $(CSRC)/mrt_device.hip $(CSRC)/mrt_post.hip $(CSRC)/mrt_host.cpp $(CSRC)/mrt_octree.cpp

# Experiment builds: make variant TAG=name DEFS="-DMRT_..." -> ray_tracing_amd/libmrt_name.so (timed with scripts/perf_variants.sh)
variant:
	$(HIPCC) $(HIPFLAGS) $(DEFS) -shared -o ray_tracing_amd/libmrt_$(TAG).so $(CSRC)/mrt_device.hip $(CSRC)/mrt_post.hip $(CSRC)/mrt_host.cpp $(CSRC)/mrt_octree.cpp

oracle: $(ORACLE)

# -march=native is deliberately NOT used: the .so is built here and run on the GPU box's host.
# -ffp-contract=off: the f32 image arithmetic of post_ref.cpp must round after every operation.
$(ORACLE): oracle/cpu_ref.cpp oracle/octree_api.cpp oracle/post_ref.cpp oracle/octree_ref.h include/mrt.h
	$(CXX) $(CXXFLAGS) -ffp-contract=off -shared -pthread -o $@ oracle/cpu_ref.cpp oracle/octree_api.cpp oracle/post_ref.cpp

clean:
	rm -f $(LIB) $(ORACLE)

.PHONY: all lib oracle stamps variant clean
