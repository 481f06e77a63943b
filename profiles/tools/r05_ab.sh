#!/bin/bash
# Counters of the streaming contains kernel at n_dim 50 and 100 (one counter
# per pass, --kernel-trace only): HBM traffic against the algorithmic bytes,
# matrix-pipe and LDS activity.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/s5; mkdir -p $O
cd /tmp
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU; do
  rm -rf /tmp/pm_$c
  NB_STREAM_N=4194304 timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$c -o p -- python $R/profiles/tools/stream_bench.py 50 100 > /dev/null 2>&1
  python - "$c" <<'PY'
import csv, glob, sys, collections
c = sys.argv[1]
fs = glob.glob('/tmp/pm_%s/**/*counter_collection.csv' % c, recursive=True)
if not fs: print(c, 'no data'); sys.exit()
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'nb_ell_stream' in r['Kernel_Name'] and r['Counter_Name'] == c:
        agg[r['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0]].append(float(r['Counter_Value']))
for k, v in agg.items():
    print('%-28s %-62s launches %3d mean %.5g' % (c, k, len(v), sum(v) / len(v)))
PY
done > $O/stream_pmc.txt 2>&1
cat $O/stream_pmc.txt
