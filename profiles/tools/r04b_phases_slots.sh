mkdir -p gpurun_out/r04b
bash profiles/tools/r04b_train.sh dd 2>&1 | grep -v "^\s*fb \|^\s*g \|barrier\|loop top\|amdgpu.ids" 
for s in 0 13; do echo "== stamped workgroup $s"; NB_PHASE_SLOT=$s NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_dbg$s.so timeout 300 python profiles/tools/train_phases.py 50 4 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r04b/train_phases_slots.txt; cat gpurun_out/r04b/train_phases_slots.txt
