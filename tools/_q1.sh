set -u
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_quad.py tests/test_gpu_fuzz.py tests/test_gpu_traceback.py -x -q > gpurun_out/q1_tests.log 2>&1
echo "tests rc=$?" ; tail -5 gpurun_out/q1_tests.log
for wl in lnw_100k_short lsw_100k_short anw_100k_short; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline >> gpurun_out/q1_bench.jsonl 2>> gpurun_out/q1_bench.err
done
cat gpurun_out/q1_bench.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['config']['algorithm'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
