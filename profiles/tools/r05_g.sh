#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
rm -f $O/slow_mode_probe.txt
for mode in plain alone trim half reverse; do
  for i in 1 2 3 4; do
    timeout 300 python profiles/tools/slow_mode_probe.py $mode 2>&1 | grep -E "D=|torch allocated" >> $O/slow_mode_probe.txt
  done
done
cut -c1-260 $O/slow_mode_probe.txt
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gmm or mvee or moments" 2>&1 | tail -3
timeout 300 python profiles/tools/gmm_bench.py > $O/gmm_bench_3.txt 2>&1
NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_gmmdbg.so timeout 300 python profiles/tools/gmm_bench.py 50 2000 50 10000 100 10000 > $O/gmm_phases_3.txt 2>&1
grep "^d=" $O/gmm_bench_3.txt; grep "\[gmm\]" $O/gmm_phases_3.txt | sort | uniq -c | sort -rn | awk 'NR%5==1' | head -6
