#!/bin/bash
# rocprofv3 kernel trace of tools/dec_lab.py (decoder, C2 + C3) -> gpurun_out/prof_dec/<variant>_kernel_stats.csv ; VARIANTS="default dB"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_dec
for v in ${VARIANTS:-default}; do
  rm -rf $R/gpurun_out/prof_dec/raw
  if [ "$v" = default ]; then unset SZ3HIP_LIB; else export SZ3HIP_LIB=$R/sz3_amd/lab/libsz3hip_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec/raw -o r -- python $R/tools/dec_lab.py > $R/gpurun_out/prof_dec/$v.log 2>&1
  python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("$R/gpurun_out/prof_dec/raw/*kernel_stats.csv")[0])))
with open("$R/gpurun_out/prof_dec/${v}_kernels.txt","w") as f:
    for r in rows:
        if any(k in r["Name"] for k in ("k_decode","k_scan","k_dec_tables","k_patch","k_dequant","k_interp_level","k_interp_pass","k_scatter")):
            line="%-110s calls %4s  avg %9.1f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1000)
            print(line); f.write(line+"\n")
PY
  rm -rf $R/gpurun_out/prof_dec/raw
done
