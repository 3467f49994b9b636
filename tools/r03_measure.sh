# Measurement set on the GPU box: the default bench line, its rocprofv3 kernel summary, the HBM counter passes, and the side
# configurations.  usage (from the repository root on the box): bash tools/r03_measure.sh     (results under gpurun_out/r03/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -3 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -3 $O/bench.err; exit 1; }
cut -c1-200 $O/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- python3 bench.py --no-cpu-baseline > $O/prof.log 2>&1 || { tail -3 $O/prof.log; exit 1; }
timeout -k 10 900 bash tools/pmc_traffic.sh r03 > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
tail -3 $O/pmc.log | cut -c1-200
timeout -k 10 300 python bench.py --config trmoe306-synth --batch 16 --no-cpu-baseline > $O/bench_trmoe306_batch16.json 2> $O/x.err || { tail -3 $O/x.err; exit 1; }
timeout -k 10 300 python bench.py --config dense306-synth --batch 1 --no-cpu-baseline > $O/bench_dense306_batch1.json 2> $O/x.err || { tail -3 $O/x.err; exit 1; }
timeout -k 10 300 python bench.py --weights fp8 --no-cpu-baseline > $O/bench_fp8_weights.json 2> $O/x.err || { tail -3 $O/x.err; exit 1; }
I64_BENCH_FORCE_DIST=1 I64H_EP_FORCE_COLLECTIVES=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout -k 10 300 python bench.py --parallelism ep --steps 32 --no-cpu-baseline > $O/bench_ep_one_rank_forced.json 2> $O/x.err || { tail -3 $O/x.err; exit 1; }
for f in bench_trmoe306_batch16 bench_dense306_batch1 bench_fp8_weights bench_ep_one_rank_forced; do python - $f $O/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"].get("avg_launch_ms"), d.get("roofline_step",{}).get("frac"), d.get("prefill_tokens_per_s"))
PY
done
