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

$(OBJDIR)/%.o: $(CSRC)/%.hip $(CSRC)/nb_common.h $(CSRC)/nb_tile.h $(CSRC)/nb_sym.h $(CSRC)/nb_mlp.h $(CSRC)/nb_draw.h include/nautilus_hip.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p nautilus_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# Debug / instrumented build (in-kernel cycle stamps etc.), never shipped:
#   make debug DEFS="-DNB_MVEE_TIMING"  ->  nautilus_amd/lib/libnautilus_hip_dbg.so
# (select it with NAUTILUS_HIP_LIB=<path>)
DBGDIR := build/obj_dbg
DBGLIB := nautilus_amd/lib/libnautilus_hip_dbg.so
DBGOBJS := $(patsubst $(CSRC)/%.hip,$(DBGDIR)/%.o,$(SRCS))
$(DBGDIR)/%.o: $(CSRC)/%.hip $(CSRC)/nb_common.h $(CSRC)/nb_tile.h $(CSRC)/nb_sym.h $(CSRC)/nb_mlp.h $(CSRC)/nb_draw.h include/nautilus_hip.h FORCE
	@mkdir -p $(DBGDIR)
	$(HIPCC) $(FLAGS) $(DEFS) -c $< -o $@
debug: $(DBGOBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $(DBGLIB) $(DBGOBJS)
FORCE:

clean:
	rm -rf build $(LIB) $(DBGLIB)
.PHONY: all clean debug FORCE
