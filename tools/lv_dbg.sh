for d in ${LVD:-0 12 44}; do echo "dbg $d"; SZ3HIP_LVDBG=$d bash tools/prof_c3.sh 2>&1 | grep "^level launches"; done
