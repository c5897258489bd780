# BASELINE configs 3 and 5 profiled where THEY run (VERDICT r4 task 4): SQ counters of every kernel + FETCH_SIZE / WRITE_SIZE passes of
#   bench.py --bytes 1000000000                         (config 3's whole load on one GPU: 15 259 blocks, four decoder waves per SIMD)
#   bench.py --workload markov --bytes 4294967296       (config 5's stream, a 4 GiB slice: 65 536 blocks in 4 batches)
# -> gpurun_out/<tag>/{sq_counters_1e9.txt, traffic_1e9.json, bench_line_1e9.json, ..._markov...}   (copied into profiles/)
# usage (through gpurun): bash tools/gpu_cfg35.sh <tag> [1e9|markov|both]
set -eo pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
TAG=${1:-r06g}
WHAT=${2:-both}
O=$R/gpurun_out/$TAG
mkdir -p $O
run_cfg() {   # name, bench arguments
  local name=$1; shift
  echo "== $name: $*"
  (cd $R && timeout -k 10 900 python3 bench.py --no-cpu --no-e2e --no-overlap --steps 3 --warmup 1 "$@" > $O/bench_line_$name.json 2>> $O/err_$name.txt) || { tail -5 $O/err_$name.txt; return 1; }
  python3 -c "import json,sys; d=json.load(open('$O/bench_line_$name.json')); print(d['value'], d['ms_per_step'], d['roundtrip_ok'], d['bytes_equal_golden'])"
  bash $R/tools/sq_counters.sh $TAG/sq_$name "$@" > /dev/null 2>> $O/err_$name.txt
  cp $O/sq_$name/sq_counters.txt $O/sq_counters_$name.txt
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-include-regex "^k_" --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$name -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap "$@" > /dev/null 2>> $O/err_$name.txt
  rocprofv3 --kernel-include-regex "^k_" --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$name -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap "$@" > /dev/null 2>> $O/err_$name.txt
  (cd $R && python3 tools/collect_traffic.py gpurun_out/$TAG/pmc_fetch_$name gpurun_out/$TAG/pmc_write_$name gpurun_out/$TAG/traffic_$name.json "--steps 1 --warmup 1 --no-cpu $* --stage full" > /dev/null)
  cd $R
  find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/pmc_fetch_$name $O/pmc_write_$name $O/sq_$name
  head -8 $O/sq_counters_$name.txt | cut -c1-200
}
if [ $WHAT = both ] || [ $WHAT = markov ]; then run_cfg markov --workload markov --bytes ${MARKOV_BYTES:-4294967296}; fi
if [ $WHAT = both ] || [ $WHAT = 1e9 ]; then run_cfg 1e9 --bytes 1000000000; fi
