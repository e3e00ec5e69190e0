# Builds the MI355X (gfx950) product: libqvzhip.so (HIP kernels + C-ABI, include/qvz_hip.h).
# hipcc cross-compiles without a GPU. -ffp-contract=off: the codebook path is decision-bearing FP64 and the
# reference is built without FMA (hard part 2).
HIPCC    ?= hipcc
ARCH     ?= gfx950
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Iqvz_amd/csrc
CSRC     := qvz_amd/csrc
HIP_SRCS := $(wildcard $(CSRC)/*.hip)
HDRS     := $(wildcard include/*.h) $(wildcard $(CSRC)/*.h)

all: qvz_amd/libqvzhip.so

qvz_amd/libqvzhip.so: $(HIP_SRCS) $(HDRS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRCS)

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f qvz_amd/libqvzhip.so

.PHONY: all oracle clean
