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
