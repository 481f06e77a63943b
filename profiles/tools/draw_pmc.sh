# PMC passes over nb_draw_kernel (one counter per pass, kernel trace only):
#   bash profiles/tools/draw_pmc.sh [n_dim] > gpurun_out/draw_pmc.txt
export TMPDIR=/tmp
D=${1:-50}
for c in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pd_$c
  timeout 180 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pd_$c -o p -- python profiles/tools/draw_pmc.py $D > /dev/null 2>&1
  python - "$c" <<'PY'
import csv, glob, sys
c = sys.argv[1]
fs = glob.glob('/tmp/pd_%s/**/*counter_collection.csv' % c, recursive=True)
if not fs: print(c, 'no data'); sys.exit()
vals = [float(r['Counter_Value']) for r in csv.DictReader(open(fs[0])) if 'nb_draw_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c]
print(c, 'launches', len(vals), 'mean %.6g' % (sum(vals)/max(1,len(vals))))
PY
done
