# Builds the MI355X (gfx950) product:
#   qvz_amd/libqvzhip.so  HIP kernels (hipcc) + host C layer (gcc): everything declared in include/*.h
#   qvz_amd/bin/qvz       the drop-in command line front end (C), linked against the library
# hipcc cross-compiles without a GPU. -ffp-contract=off: the codebook path is decision-bearing FP64 and the
# reference is built without FMA (hard part 2).
MAKEFLAGS += -r          # no built-in rules ("%: %.o" would make x.hip depend on x.hip.o)
.SUFFIXES:
HIPCC    ?= hipcc
CC       ?= gcc
ARCH     ?= gfx950
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Iqvz_amd/csrc
CFLAGS   ?= -O2 -ffp-contract=off -fPIC -std=gnu11 -Wall -Wextra -Iinclude
CSRC     := qvz_amd/csrc
HIP_SRCS := $(wildcard $(CSRC)/*.hip)
HOST_SRCS:= $(CSRC)/host/qvz_coder.c $(CSRC)/host/qvz_encode.c $(CSRC)/host/qvz_shard.c $(CSRC)/host/qvz_decode.c $(CSRC)/host/qvz_fastq.c $(CSRC)/host/qvz_shardset.c $(CSRC)/host/qvz_multi.c $(CSRC)/host/qvz_codebook_host.c $(CSRC)/host/qvz_egress.c
HOST_OBJS:= $(HOST_SRCS:.c=.o)
HIP_OBJS := $(HIP_SRCS:.hip=.hip.o)
HDRS     := $(wildcard include/*.h) $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/host/*.h)

all: qvz_amd/libqvzhip.so qvz_amd/libqvzcompat.so qvz_amd/bin/qvz qvz_amd/bin/qvz-shardset

# the reference's own entry-point names (include/qvz_compat.h) in a library of their own
qvz_amd/libqvzcompat.so: $(CSRC)/host/qvz_compat.c qvz_amd/libqvzhip.so $(HDRS)
	$(CC) $(CFLAGS) -I$(CSRC)/host -shared -o $@ $(CSRC)/host/qvz_compat.c -Lqvz_amd -lqvzhip -Wl,-rpath,'$$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib

$(CSRC)/host/%.o: $(CSRC)/host/%.c $(HDRS)
	$(CC) $(CFLAGS) -c $< -o $@

$(CSRC)/%.hip.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

qvz_amd/libqvzhip.so: $(HIP_OBJS) $(HOST_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIP_OBJS) $(HOST_OBJS) -lm -lpthread -ldl

qvz_amd/bin/qvz: $(CSRC)/host/qvz_main.c qvz_amd/libqvzhip.so $(HDRS)
	mkdir -p qvz_amd/bin
	$(CC) $(CFLAGS) -fPIE -o $@ $(CSRC)/host/qvz_main.c -Lqvz_amd -lqvzhip -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath-link,/opt/rocm/lib

qvz_amd/bin/qvz-shardset: $(CSRC)/host/qvz_shardset_tool.c qvz_amd/libqvzhip.so $(HDRS)
	mkdir -p qvz_amd/bin
	$(CC) $(CFLAGS) -fPIE -o $@ $(CSRC)/host/qvz_shardset_tool.c -Lqvz_amd -lqvzhip -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath-link,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f qvz_amd/libqvzhip.so qvz_amd/libqvzcompat.so qvz_amd/bin/qvz qvz_amd/bin/qvz-shardset $(HOST_OBJS) $(HIP_OBJS)

.PHONY: all oracle clean
