for t in "oracle_sweep.py 24" "early_reject_sweep.py 40" "sharded_sweep.py 60" "devpack_sweep.py 60" "dense_sweep.py 16" "lk_graph_sweep.py" "shi_sweep.py"; do
  echo "== $t"; timeout 1200 python tools/$t 2>&1 | grep -v "^HIP\|^ROCm\|^Host\|^Librccl" | tail -3
done
