# full check on the GPU box: tests, smoke, bench (+ rocprofv3 kernel stats), HBM traffic passes
set -eo pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
TAG=${1:-r01d}
mkdir -p $R/gpurun_out/$TAG
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > $R/gpurun_out/$TAG/pytest_gpu.txt
cat $R/gpurun_out/$TAG/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $R/gpurun_out/$TAG/bench_line.json 2> $R/gpurun_out/$TAG/bench_err.txt
tail -c 3000 $R/gpurun_out/$TAG/bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -- python3 $R/bench.py --no-cpu --no-e2e --no-overlap > $R/gpurun_out/$TAG/bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap > /dev/null 2>&1
for c in rox rolz; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats_$c -- python3 $R/bench.py --no-cpu --no-e2e --no-overlap --codec $c > $R/gpurun_out/$TAG/bench_line_$c.json 2>/dev/null
done
cd $R
python3 tools/collect_traffic.py gpurun_out/$TAG/pmc_fetch gpurun_out/$TAG/pmc_write gpurun_out/$TAG/traffic.json "bench.py --steps 1 --warmup 1, 1e8 B shard" > /dev/null
cat gpurun_out/$TAG/stats/*/*kernel_stats.csv | cut -c1-120 | head -14
for c in rox rolz; do cat gpurun_out/$TAG/stats_$c/*/*kernel_stats.csv | cut -c1-100 | head -10; done
find gpurun_out/$TAG -name "*.csv" -size +2M -delete
find gpurun_out/$TAG -name "*counter_collection.csv" -delete
