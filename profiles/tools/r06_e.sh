# round 6, fifth call: where a reference-settings run at 50 dimensions spends
# its time (kernel statistics + host profile), MVEE after the one-level
# candidate selection
O=gpurun_out/r06e; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -k "mvee or whiten or ellipsoid" > $O/tests_mvee.txt 2>&1; tail -2 $O/tests_mvee.txt
python profiles/tools/mvee_phases.py > $O/mvee_fit_times.txt 2>&1; grep "per fit" $O/mvee_fit_times.txt
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rm -rf /tmp/r06_f50 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_f50 -o f50 -- python $R/profiles/tools/r06_anchor_runs.py funnel 50 0 > /tmp/f50.log 2>&1)
f=$(find /tmp/r06_f50 -name '*kernel_stats.csv' | head -1)
head -25 "$f" | cut -c1-230 > $O/funnel50_kernel_stats.csv; cat $O/funnel50_kernel_stats.csv | cut -c1-200
timeout 900 python profiles/tools/small_batch_profile.py 50 100 > $O/small_batch_profile_D50.txt 2>&1; grep -E "exploration|sampling phase|log Z" $O/small_batch_profile_D50.txt
