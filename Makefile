# Build the MI355X (gfx950) hot-path library in-tree.  hipcc cross-compiles
# without a GPU; the .so is git-ignored but travels to the GPU box.
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := nautilus_amd/csrc
OBJDIR := build/obj
LIB := nautilus_amd/lib/libnautilus_hip.so
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-unused-value

all: $(LIB)

$(OBJDIR)/%.o: $(CSRC)/%.hip $(CSRC)/nb_common.h $(CSRC)/nb_tile.h include/nautilus_hip.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p nautilus_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(LIB)
.PHONY: all clean
