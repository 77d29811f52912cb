# Builds libsegan_hip.so (the C-ABI HIP library) in-tree for gfx950, and the oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := segan_pytorch_amd/csrc
SRCS := $(CSRC)/segan_api.hip $(CSRC)/segan_conv.hip $(CSRC)/segan_conv_edge.hip $(CSRC)/segan_wgrad.hip $(CSRC)/segan_pack.hip $(CSRC)/segan_conv_bf2.hip $(CSRC)/segan_wgrad_bf2.hip $(CSRC)/segan_pointwise.hip $(CSRC)/segan_stft.hip $(CSRC)/segan_snorm.hip $(CSRC)/segan_gemm.hip $(CSRC)/segan_audio.hip $(CSRC)/segan_comm.hip
OBJS := $(SRCS:.hip=.o)
LIB := segan_pytorch_amd/libsegan_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/segan_common.h $(CSRC)/segan_conv_shared.h include/segan_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean
