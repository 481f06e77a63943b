#!/bin/bash
# prefetch depth of the geometric stage (tuning builds, -DNB_CAND_PD=...)
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O; rm -f $O/cand_pd.txt
for lib in "" _pd7 _pd12 _pd14 _pd16 _pd20 _T3 _T3pd14; do
  echo "== libnautilus_hip$lib.so" >> $O/cand_pd.txt
  NB_STAGE_TIMING=1 NB_ACCEPT_REPS=10 NAUTILUS_HIP_LIB=nautilus_amd/lib/libnautilus_hip$lib.so timeout 300 python profiles/tools/accept_bench.py 50 100 20 2>&1 | grep -E "proposals:|call (8|18|28):" | cut -c1-170 >> $O/cand_pd.txt
done
cat $O/cand_pd.txt
