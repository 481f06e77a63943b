#!/bin/bash
# (Run WITHOUT a pipe behind it: the jobs inherit stdout.)
# (Re)start the long reference jobs of round 6; every one resumes from its
# checkpoint under gpurun_out/refjobs/ and skips itself when its result exists.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/refjobs
export OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1
running() { ps aux | grep -v grep | grep -q "$1"; }
for t in reduced_D50_seed0 reduced_D50_seed1; do
  [ -f tests/golden/funnel_parts/$t.json ] || running "make_golden_funnel.py $t" || \
    nohup setsid python tests/golden/make_golden_funnel.py $t >> gpurun_out/refjobs/$t.log 2>&1 &
done
for s in 0 1; do
  running "make_golden_c5_prefix.py $s " || \
    nohup setsid python tests/golden/make_golden_c5_prefix.py $s 36000 >> gpurun_out/refjobs/c5prefix_$s.log 2>&1 &
done
# (a second C3 run on one core was started in round 6 and stopped 1.7 h in:
# 36 of ~185 bounds -- it needs ~9 h; `make_golden_c3.py 1 single` resumes)
sleep 3
ps aux | grep make_golden | grep -v grep | wc -l
