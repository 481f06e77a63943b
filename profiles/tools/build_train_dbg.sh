#!/bin/bash
# Stamped trainer libraries (never shipped): one per stamped workgroup.
#   profiles/tools/build_train_dbg.sh 0 13 30  ->  nautilus_amd/lib/libnautilus_hip_dbg<slot>.so
set -e
make >/dev/null 2>&1
for slot in "$@"; do
  mkdir -p build/obj_dbg$slot
  defs="-DNB_TRAIN_TIMING_SLOT=$slot"
  # "slots": the per-workgroup table of profiles/tools/train_slots.py
  [ $slot = slots ] && defs="-DNB_TRAIN_SLOT_TIMING"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DNB_TRAIN_TIMING $defs \
    -c nautilus_amd/csrc/nb_mlp_train.hip -o build/obj_dbg$slot/nb_mlp_train.o 2>/dev/null
  objs=$(ls build/obj/*.o | grep -v nb_mlp_train.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o nautilus_amd/lib/libnautilus_hip_dbg$slot.so $objs build/obj_dbg$slot/nb_mlp_train.o
done
