#!/bin/bash
# PMC counters of the decoder kernels (tools/dec_lab.py: C2 + C3) -> gpurun_out/pmc_dec.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/tools/dec_lab.py"
: > $R/gpurun_out/pmc_dec.txt
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM"; do
  rm -rf /tmp/pd; rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pd -o p -- $B > /dev/null 2>&1
  python - >> $R/gpurun_out/pmc_dec.txt <<PY
import csv,glob,collections
f=glob.glob("/tmp/pd/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "k_decode" in k or "k_scan_strided_half" in k:
        acc[k.split("(")[0][-36:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(k, {c: "%.4g" % (sum(v)/len(v)) for c,v in acc[k].items()})
PY
done
cat $R/gpurun_out/pmc_dec.txt
