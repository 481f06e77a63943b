# Counter passes over the dominant kernel of the timed step (the fused
# acceptance kernel nb_eval_fast_kernel<4,4,false>): one counter per rocprofv3
# pass, kernel trace only; the large launches (the main refill of every step)
# of `python bench.py --no-cpu-baseline --timed-region-only`.
#   bash profiles/tools/fast_pmc.sh > gpurun_out/fast_pmc.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for c in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY; do
  rm -rf /tmp/fp_$c
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/fp_$c -o p -- python $R/bench.py --no-cpu-baseline --timed-region-only --steps 6 --warmup 1 > /dev/null 2>&1
  python3 - "$c" <<'PY'
import csv, glob, sys
c = sys.argv[1]
fs = glob.glob('/tmp/fp_%s/**/*counter_collection.csv' % c, recursive=True)
if not fs: print(c, 'no data'); sys.exit()
rows = [r for r in csv.DictReader(open(fs[0])) if 'nb_eval_fast_kernel<4, 4, false>' in r['Kernel_Name'] and r['Counter_Name'] == c]
big = max(int(r['Grid_Size']) for r in rows) if rows else 0
vals = [float(r['Counter_Value']) for r in rows if int(r['Grid_Size']) == big]
print(c, 'launches', len(vals), 'grid', big, 'mean %.6g' % (sum(vals) / max(1, len(vals))))
PY
done
