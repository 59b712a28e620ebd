#!/bin/bash
# SQ counters of the C3 bench's kernels (interpolation level kernel in particular) -> stdout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B3="python $R/bench.py --algo interp --eb 1e-4 --steps 4 --warmup 1 --no-cpu-baseline --no-host-e2e --no-extra"
rm -rf $R/gpurun_out/pmc3_sq $R/gpurun_out/pmc3_lds
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc3_sq -o p -- $B3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc3_lds -o p -- $B3 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
for d in ("pmc3_sq","pmc3_lds"):
    f=glob.glob("$R/gpurun_out/%s/*counter_collection.csv"%d)[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "k_interp_level<float, false>" not in k: continue
        acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g in sorted(acc, key=lambda x:int(x)):
        print("grid", g, {c: round(sum(v)/len(v)) for c,v in acc[g].items()})
PY
