# the bench lines that are not the headline, with the code as it stands: comprox / comprolz, the harder corpus, config 3 on one GPU,
# config 5 at its size. usage (through gpurun): bash tools/gpu_lines.sh <tag>
set -eo pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-lines}
mkdir -p $O
for c in rox rolz; do timeout -k 10 300 python3 bench.py --no-cpu --codec $c --steps 10 --warmup 2 > $O/bench_line_$c.json 2>> $O/err.txt; done
for c in rop rox rolz; do timeout -k 10 300 python3 bench.py --no-cpu --no-e2e --workload enwik-hard --codec $c --steps 5 --warmup 1 > $O/bench_line_hard_$c.json 2>> $O/err.txt; done
timeout -k 10 600 python3 bench.py --no-cpu --bytes 1000000000 --steps 3 --warmup 1 > $O/bench_line_1e9.json 2>> $O/err.txt
timeout -k 10 900 python3 bench.py --no-cpu --workload markov --bytes 17179869184 --steps 1 --warmup 0 > $O/bench_line_markov_16g.json 2>> $O/err.txt
python3 - <<PY
import json, glob
for f in sorted(glob.glob('$O/bench_line_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], 'MB/s', d['ms_per_step'], 'ms', 'roundtrip', d['roundtrip_ok'], 'golden', d['bytes_equal_golden'], (d.get('two_steps_in_flight') or {}).get('value'), (d.get('end_to_end') or {}).get('roundtrip_MBps'))
PY
