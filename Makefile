# Build the gfx950 HIP library in-tree (the .so travels to the GPU box with the snapshot).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := innfer_amd/csrc/conv3x3.hip innfer_amd/csrc/hr_chain.hip innfer_amd/csrc/conv_first.hip innfer_amd/csrc/tiles.hip innfer_amd/csrc/net.hip innfer_amd/csrc/unet.hip innfer_amd/csrc/pan.hip innfer_amd/csrc/pan_scpa.hip innfer_amd/csrc/pan_scpa_split.hip innfer_amd/csrc/f32ops.hip innfer_amd/csrc/colorfix.hip innfer_amd/csrc/ppon.hip innfer_amd/csrc/resnet.hip innfer_amd/csrc/wbcunet.hip innfer_amd/csrc/comm.hip
OBJ   := $(SRC:.hip=.o)
LIB   := innfer_amd/lib/libinnfer_amd.so
FLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function

all: $(LIB)

innfer_amd/csrc/tiles.o innfer_amd/csrc/colorfix.o: FLAGS += -ffp-contract=off

%.o: %.hip $(wildcard innfer_amd/csrc/*.h) $(wildcard innfer_amd/csrc/*.inc) include/innfer_amd.h
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p innfer_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ) -ldl

# diagnostic library with in-kernel s_memtime stamps (never shipped, never benchmarked)
stamps:
	@mkdir -p build/stamps
	$(HIPCC) $(FLAGS) -DINNFER_STAMPS -DINNFER_ABLATE -shared -o innfer_amd/lib/libinnfer_amd_stamps.so $(SRC)

# diagnostic library whose conv kernel can skip its stores / DMA / MFMA phase (INNFER_ABL bits; never shipped)
ablate:
	$(HIPCC) $(FLAGS) -DINNFER_ABLATE -shared -o innfer_amd/lib/libinnfer_amd_ablate.so $(SRC)

clean:
	rm -f $(OBJ) $(LIB)

.PHONY: all clean stamps ablate ab_rowp0

# A/B library: the 64-output layers of the SR networks in the lane-contiguous row order (INNFER_ROWP_DEFAULT 0; scripts/r4/rowp_ab.sh; never shipped)
ab_rowp0: $(OBJ)
	$(HIPCC) $(FLAGS) -DINNFER_ROWP_DEFAULT=0 -c innfer_amd/csrc/net.hip -o innfer_amd/csrc/net_rowp0.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o innfer_amd/lib/libinnfer_amd_rowp0.so $(filter-out innfer_amd/csrc/net.o,$(OBJ)) innfer_amd/csrc/net_rowp0.o -ldl
	rm -f innfer_amd/csrc/net_rowp0.o
