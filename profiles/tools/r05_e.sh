#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "list_eval or gmm or mvee or moments or narrow" 2>&1 | tail -30 > $O/parity_e.log
tail -4 $O/parity_e.log
rm -f $O/accept_cprofile.txt
for i in 1 2 3 4; do
  echo "== process $i" >> $O/accept_cprofile.txt
  NB_ACCEPT_CPROFILE=1 timeout 300 python profiles/tools/accept_bench.py 50 100 2>&1 | grep -v "^$" | grep -E "proposals:|tottime|\{|py:|function calls" | cut -c1-150 >> $O/accept_cprofile.txt
done
grep -E "^==|D=100" $O/accept_cprofile.txt
timeout 300 python profiles/tools/gmm_bench.py > $O/gmm_bench_2.txt 2>&1
NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip_gmmdbg.so timeout 300 python profiles/tools/gmm_bench.py 50 2000 50 10000 100 10000 > $O/gmm_phases_2.txt 2>&1
grep "^d=" $O/gmm_bench_2.txt; grep "\[gmm\]" $O/gmm_phases_2.txt | sort | uniq -c | sort -rn | awk 'NR%5==1' | head -6
timeout 900 python profiles/tools/funnel_shell_check.py 20 100 0 > $O/funnel_shell_check_D20.txt 2>&1
cat $O/funnel_shell_check_D20.txt | cut -c1-330
