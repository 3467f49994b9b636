# Measurement set on the GPU box: GPU tests, the default bench line, its rocprofv3 kernel summary, the HBM counter passes.
# usage (from the repository root on the box): bash tools/final_measure.sh <tag>     (results under gpurun_out/<tag>/)
set -e
TAG=${1:-final}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/$TAG/tests.log 2>&1
tail -1 gpurun_out/$TAG/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1
tail -1 gpurun_out/$TAG/smoke.log
timeout -k 10 400 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
cut -c1-300 gpurun_out/$TAG/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/prof -o bench --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/$TAG/prof.log 2>&1
tail -1 gpurun_out/$TAG/prof.log | cut -c1-200
timeout -k 10 900 bash tools/pmc_traffic.sh $TAG > gpurun_out/$TAG/pmc.log 2>&1
tail -4 gpurun_out/$TAG/pmc.log | cut -c1-300
