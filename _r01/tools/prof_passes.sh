# per-dispatch durations of the interpolation pass kernels of one compress call (which levels cost what)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LAB_SHAPE=${LAB_SHAPE:-512,512,512} LAB_ALGO=${LAB_ALGO:-interp} LAB_EB=${LAB_EB:-1e-3} timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pp -o r -- python $R/tools/shape_lab.py > /tmp/pp.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('/tmp/pp/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last complete compress call: from the k_assemble before the second-last k_interp_anchors to the next k_assemble
idx=[i for i,r in enumerate(rows) if 'k_interp_anchors' in r['Kernel_Name']]
i0=idx[-2] if len(idx) > 1 else idx[-1]
j=i0
while j > 0 and 'k_assemble' not in rows[j-1]['Kernel_Name']: j-=1
t0=int(rows[j]['Start_Timestamp']); prev=t0
for r in rows[j:j+80]:
    n=r['Kernel_Name']; st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
    print(n[:56].ljust(56), str(r.get('Grid_Size_X', r.get('Grid_Size',''))).rjust(9), 'at %8.1f us'%((st-t0)/1e3), 'dur %7.1f'%((en-st)/1e3), 'gap %6.1f'%((st-prev)/1e3))
    prev=en
    if 'k_assemble' in n: break
PY
python - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pp/**/*kernel_trace.csv',recursive=True)[0]
c=collections.Counter(r['Kernel_Name'].split('(')[0][:50] for r in csv.DictReader(open(f)))
for k in ('void k_interp_anchors<double>','void k_interp_anchors<float>','k_assemble','void k_hist_codes<double, false, false>','void k_hist_codes<float, false, false>'):
    print(k, c.get(k))
PY
grep -a "compress" /tmp/pp.log | cut -c1-330
