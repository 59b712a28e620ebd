#!/bin/bash
# a 4 MB chunk through the HDF5 filter function: wall time per call (host clock) and the device's busy time per forward call
# (sum of all kernel and copy durations of the run's 12 forward calls, the one read-direction call's kernels left out by name)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r6
for args in "1d default" "3d default" "1d lorenzo_reg" "3d lorenzo_reg"; do
rm -rf /tmp/h5; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/h5 -o r -- python $R/tools/r6/h5z_chunk.py $args > /tmp/h5.log 2>&1
grep "h5z chunk" /tmp/h5.log
python - <<PY
import csv,glob
dec=("k_dec","k_scan","k_blkn_pre","k_blkn2","k_blkn_apply","k_blkn_scan","k_blk_patch","k_blk_side_sel","k_blk_coef_parse","k_blk_coef_gscan","k_blk_coef_apply","k_patch","k_interp_first_dec","k_decode","k_blk_wave","k_blk_local","k_blk_final")
kt=0; n=0
for f in glob.glob("/tmp/h5/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        nm=r["Kernel_Name"]
        if any(d in nm for d in dec): continue
        kt+=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); n+=1
ct=0
for f in glob.glob("/tmp/h5/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): ct+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
print("   device busy per forward call: kernels %.1f us (%d launches per call), copies %.1f us (both directions of the run, per call)" % (kt/12/1000, round(n/12), ct/13/1000))
PY
done 2>&1 | tee $R/gpurun_out/r6/h5z_tl.log
