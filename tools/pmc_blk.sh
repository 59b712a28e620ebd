#!/bin/bash
# PMC counters of the block-composed path's passes (selection, fit, Lorenzo) at C4's slab; $FIELD = default | c4a
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --algo composed --field ${FIELD:-default} --dtype f64 --shape 128,1024,1024 --eb 1e-6 --steps 3 --warmup 2 --no-cpu-baseline --no-host-e2e --no-extra"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_GDS SQ_INSTS_FLAT"; do
  rm -rf /tmp/pb; rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pb -o p -- $B > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pb/*counter_collection.csv")[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "k_blk_fit" in k or "k_blk_lorenzo" in k or "k_blk_select" in k or "k_blk_rows" in k or "k_lorenzo_quant_march" in k:
        import re
        if "k_lorenzo_quant_march" in k and ", 1, false>" in k: continue  # (the one-byte twin of the two-launch form: returns at once here)
        acc[re.search(r"k_blk_\w+|k_lorenzo_quant_march\w*", k).group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc: print(k, {c: "%.3g" % (sum(v)/len(v)) for c,v in acc[k].items()})
PY
done
