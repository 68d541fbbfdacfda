set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r4e
mkdir -p $OUT
export TMPDIR=/tmp
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  n=$(echo $P | cut -d" " -f1)
  (cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $OLDPWD/$OUT/pmc/$n -o p -- python $OLDPWD/tools/pmc_probe2.py > $OLDPWD/$OUT/pmc_$n.log 2>&1)
done
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_traffic.json 2>&1 | tail -6
(cd /tmp && timeout -s KILL 60 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OLDPWD/$OUT/grbm_probe -o g -- $OLDPWD/tools/probes/mfma_rate > $OLDPWD/$OUT/grbm_probe.txt 2>&1)
head -5 $OUT/grbm_probe.txt
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r4e/grbm_probe/*/*counter_collection.csv')+glob.glob('gpurun_out/r4e/grbm_probe/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        ns=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
        if ns>5e6: print(r['Kernel_Name'][:40], r['Counter_Name'], 'clock GHz (counter/8/ns)', float(r['Counter_Value'])/8/ns, 'ms', ns/1e6)
PY
