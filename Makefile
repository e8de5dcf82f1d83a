# Builds the product library (HIP, gfx950) and the CPU oracle (plain C).
# Both are built with -ffp-contract=off: the only fused multiply-adds are the
# ones include/bdpt_clmath.h writes out.

HIPCC   ?= hipcc
CC      ?= gcc
PKG     := opencl-path-tracer_amd
CSRC    := $(PKG)/csrc
ARCH    ?= gfx950

HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function
ORCFLAGS := -O2 -std=c11 -ffp-contract=off -fPIC -shared -fopenmp -Wall -Wno-unused-function

LIB      := $(PKG)/libbdpt_amd.so
HOSTBIN  := $(PKG)/bdpt_host
ORACLE   := oracle/libbdpt_oracle.so
ORACLE_NOFMA := oracle/libbdpt_oracle_nofma.so

LIB_SRCS := $(CSRC)/bdpt_api.cpp $(CSRC)/bdpt_group.cpp $(CSRC)/bdpt_kernels.hip
LIB_DEPS := $(LIB_SRCS) $(wildcard $(CSRC)/*.h) include/bdpt.h include/bdpt_clmath.h $(CSRC)/bdpt_slice_probe.hip tools/check_slice_isa.py

HOSTLIB  := $(PKG)/libbdpt_host.so
HOST_SRCS := $(PKG)/host/camera.cpp $(PKG)/host/scenes.cpp $(PKG)/host/renderer.cpp $(PKG)/host/scene_io.cpp $(PKG)/host/image_io.cpp $(PKG)/host/capi.cpp
HOST_DEPS := $(HOST_SRCS) $(PKG)/host/bdpt_host.h include/bdpt.h
CXX      ?= g++
CXXFLAGS := -O2 -std=c++17 -fPIC -Wall

CLBASE   := tools/cl_baseline

all: $(LIB) $(HOSTLIB) $(HOSTBIN) $(ORACLE) $(ORACLE_NOFMA) $(CLBASE)

# harness for the reference's own kernel on a CPU OpenCL device (a reported baseline; needs only the OpenCL loader)
$(CLBASE): tools/cl_baseline.c
	$(CC) -O2 -std=c99 -Wall tools/cl_baseline.c -o $@ -lOpenCL

# host-side renderer surface (C++), linked against the C-ABI library beside it
$(HOSTLIB): $(HOST_DEPS) $(LIB)
	$(CXX) $(CXXFLAGS) -shared $(HOST_SRCS) -o $@ -L$(PKG) -lbdpt_amd -Wl,-rpath,'$$ORIGIN'

$(HOSTBIN): $(PKG)/host/main.cpp $(HOSTLIB)
	$(CXX) $(CXXFLAGS) $(PKG)/host/main.cpp -o $@ -L$(PKG) -lbdpt_host -lbdpt_amd -Wl,-rpath,'$$ORIGIN'

# the library is built only if the sliced-launch hand-off (csrc/bdpt_slice.h) compiles to the store, wait and
# invalidate instructions its correctness rests on
$(LIB): $(LIB_DEPS)
	python3 tools/check_slice_isa.py
	$(HIPCC) $(HIPFLAGS) -shared -x hip $(LIB_SRCS) -o $@ -ldl

# -mfma: fmaf() becomes one instruction; the no-FMA build calls libm's fmaf(),
# which rounds identically, for hosts without the instruction.
$(ORACLE): oracle/bdpt_oracle.c oracle/bdpt_oracle.h include/bdpt.h include/bdpt_clmath.h
	$(CC) $(ORCFLAGS) -mfma oracle/bdpt_oracle.c -o $@ -lm

$(ORACLE_NOFMA): oracle/bdpt_oracle.c oracle/bdpt_oracle.h include/bdpt.h include/bdpt_clmath.h
	$(CC) $(ORCFLAGS) oracle/bdpt_oracle.c -o $@ -lm

clean:
	rm -f $(LIB) $(HOSTLIB) $(ORACLE) $(ORACLE_NOFMA) $(HOSTBIN) $(CLBASE)

.PHONY: all clean
