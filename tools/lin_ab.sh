# lineariser / operator timings (tools/lin_probe.py: HIP events, profiling context) of the built library against a second one: bash tools/lin_ab.sh <other .so> [configs]
O=$1; shift
for rep in 1 2; do
  echo "== built library (pass $rep)"; python tools/lin_probe.py "${@:-C2 C4}" 2>&1 | grep workload
  echo "== $O (pass $rep)"; NRS_LIB=$PWD/$O python tools/lin_probe.py "${@:-C2 C4}" 2>&1 | grep workload
done
