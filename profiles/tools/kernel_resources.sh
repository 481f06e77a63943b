#!/bin/bash
# Registers / spills / LDS of every kernel of one source file (compile only):
#   profiles/tools/kernel_resources.sh nautilus_amd/csrc/nb_eval_fast.hip [filter]
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Iinclude -Inautilus_amd/csrc -c "$1" -o /tmp/nb_res.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re, subprocess
flt = sys.argv[1] if len(sys.argv) > 1 else ''
name = None; rec = {}
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '')
        rec = {}
        continue
    for key in ('VGPRs', 'AGPRs', 'VGPRs Spill', 'SGPRs Spill', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]'):
        m = re.search(re.escape(key) + r': (\d+)', l)
        if m and key not in rec:
            rec[key] = m.group(1)
    if 'LDS Size' in l and name and flt in name:
        print(re.sub(r'\(.*', '', name), ' '.join('%s=%s' % (k.split(' [')[0].replace(' ', '_'), v) for k, v in rec.items()))
" "$2"
