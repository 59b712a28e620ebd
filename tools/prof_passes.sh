# per-dispatch durations of the interpolation pass kernels of one compress call (which levels cost what)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LAB_SHAPE=${LAB_SHAPE:-512,512,512} LAB_ALGO=interp LAB_EB=1e-3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o r -- python $R/tools/shape_lab.py > /tmp/pp.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('/tmp/pp/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last compress call: from the last k_interp_anchors to the following k_hist_codes
idx=[i for i,r in enumerate(rows) if 'k_interp_anchors' in r['Kernel_Name']]
i0=idx[-1]
for r in rows[i0:i0+40]:
    n=r['Kernel_Name']
    print(n[:60].ljust(60), r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''), '%.1f us'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
    if 'k_hist_codes' in n: break
PY
