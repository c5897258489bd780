# round-2 measurement run on the GPU box: bench lines (all codecs, both stages), rocprofv3 kernel stats, HBM traffic passes.
# usage (through gpurun): bash tools/gpu_measure.sh <tag> [codecs="rop rox rolz"]
set -eo pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
TAG=${1:-r02a}
CODECS=${2:-rop rox rolz}
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_line_rop.json 2> $O/bench_err.txt
tail -c 2500 $O/bench_line_rop.json; echo
timeout -k 10 300 python bench.py --stage codec --no-cpu > $O/bench_line_rop_codec.json 2>> $O/bench_err.txt
cd /tmp && export TMPDIR=/tmp
for c in $CODECS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -- python3 $R/bench.py --no-cpu --no-e2e --no-overlap --codec $c > $O/bench_under_rocprof_$c.json 2>/dev/null
  cp $O/stats_$c/*/*kernel_stats.csv $O/kernel_stats_$c.csv
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap --codec $c > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap --codec $c > /dev/null 2>&1
  (cd $R && python3 tools/collect_traffic.py gpurun_out/$TAG/pmc_fetch_$c gpurun_out/$TAG/pmc_write_$c gpurun_out/$TAG/traffic_$c.json "--steps 1 --warmup 1 --no-cpu --codec $c --stage full" > /dev/null)
  if [ $c = rop ]; then    # the codec-stage line's dominant kernel too (64 KiB of text per block)
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_${c}_codec -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap --codec $c --stage codec > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_${c}_codec -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap --codec $c --stage codec > /dev/null 2>&1
    (cd $R && python3 tools/collect_traffic.py gpurun_out/$TAG/pmc_fetch_${c}_codec gpurun_out/$TAG/pmc_write_${c}_codec gpurun_out/$TAG/traffic_${c}_codec.json "--steps 1 --warmup 1 --no-cpu --codec $c --stage codec" > /dev/null)
  fi
  if [ $c != rop ]; then timeout -k 10 300 python3 $R/bench.py --no-cpu --codec $c > $O/bench_line_$c.json 2>> $O/bench_err.txt; fi
  head -14 $O/kernel_stats_$c.csv | cut -c1-110
done
cd $R
find gpurun_out/$TAG -name "*.csv" -size +1M -delete
find gpurun_out/$TAG -name "*counter_collection.csv" -delete
find gpurun_out/$TAG -name "*.db" -delete
rm -rf gpurun_out/$TAG/stats_* gpurun_out/$TAG/pmc_*
