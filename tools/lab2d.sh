for sh in 8192,8192 1800,3600 134217728; do
for nv in 0 128; do
echo "== $sh novec=$nv"
LAB_DBG=$nv LAB_SHAPE=$sh LAB_ALGO=interp LAB_EB=1e-3 timeout 120 python tools/shape_lab.py 2>&1 | tail -1
done; done
LAB_SHAPE=8192,8192 LAB_ALGO=default LAB_EB=1e-3 timeout 120 python tools/shape_lab.py 2>&1 | tail -1
