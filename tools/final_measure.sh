set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/final_tests.log 2>&1
tail -1 gpurun_out/final_tests.log
timeout -k 10 400 python bench.py > gpurun_out/final_bench.log 2>&1
tail -1 gpurun_out/final_bench.log | cut -c1-400
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/final_prof -o final --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/final_prof.log 2>&1
tail -1 gpurun_out/final_prof.log | cut -c1-200
timeout -k 10 700 bash tools/pmc_traffic.sh final6 > gpurun_out/final_pmc.log 2>&1
tail -3 gpurun_out/final_pmc.log | cut -c1-300
