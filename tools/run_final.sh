timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/run_full_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/run_full_tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_r03.sh > gpurun_out/profile_r03.log 2>&1
tail -3 gpurun_out/r03/bench.json | cut -c1-300
