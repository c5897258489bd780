# HBM traffic passes (FETCH_SIZE / WRITE_SIZE, separately) of the comprox or comprolz bench on the GPU box -> gpurun_out/<tag>/<codec>_traffic.json
set -eo pipefail
R=$GRAFT_REPO_ROOT
TAG=${1:-r01l}
C=${2:-rox}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch_$C -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --codec $C > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write_$C -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --codec $C > /dev/null 2>&1
cd $R
python3 tools/collect_traffic.py gpurun_out/$TAG/pmc_fetch_$C gpurun_out/$TAG/pmc_write_$C gpurun_out/$TAG/${C}_traffic.json "bench.py --codec $C --steps 1 --warmup 1, 1e8 B shard" > /dev/null
find gpurun_out/$TAG -name "*counter_collection.csv" -delete
python3 -c "
import json; t=json.load(open('gpurun_out/$TAG/${C}_traffic.json'))
print({k: round(v['hbm_raw']/1e9,2) for k,v in t['kernels'].items()})"
