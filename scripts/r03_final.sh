#!/bin/bash
# round 3, last call: the whole GPU suite, smoke, the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
out=gpurun_out/r03_final.txt
: > $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -v -p no:cacheprovider --timeout 500 > gpurun_out/r03_final_pytest.log 2>&1
echo "pytest rc=$?" >> $out
tail -2 gpurun_out/r03_final_pytest.log >> $out
python -c "import __graft_entry__ as g; g.smoke()" >> $out 2>&1
timeout -k 10 400 python bench.py >> $out 2> gpurun_out/r03_final_bench.err
echo "bench rc=$?" >> $out
cat $out
