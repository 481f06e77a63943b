export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u | head -20
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT; do
  rm -rf /tmp/pm_$c
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pm_$c -o p -- python profiles/tools/eval_pmc.py > /dev/null 2>&1
  python - "$c" <<'PY'
import csv, glob, sys
c = sys.argv[1]
fs = glob.glob('/tmp/pm_%s/*counter_collection.csv' % c)
if not fs: print(c, 'no data'); sys.exit()
vals = [float(r['Counter_Value']) for r in csv.DictReader(open(fs[0])) if 'nb_eval_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c]
print(c, 'launches', len(vals), 'mean %.4g' % (sum(vals)/max(1,len(vals))))
PY
done
