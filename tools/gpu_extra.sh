# extra measurements of round 2: config 5 (Markov) workload, strong-scaling code path on one rank, decode time against resident blocks
set -eo pipefail
cd $GRAFT_REPO_ROOT
TAG=${1:-r02c}
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 300 python bench.py --workload markov --no-cpu > $O/bench_line_markov.json 2> $O/err.txt
timeout -k 10 300 python bench.py --workload markov --stage codec --no-cpu > $O/bench_line_markov_codec.json 2>> $O/err.txt
timeout -k 10 300 python bench.py --scaling strong --no-cpu > $O/bench_line_strong_n1.json 2>> $O/err.txt
timeout -k 10 300 python tools/dec_bench.py v5 1526,1024,512,256,64,1 > $O/dec_bench.txt 2>> $O/err.txt
cat $O/dec_bench.txt
python3 - <<PY
import json
for f in ("bench_line_markov", "bench_line_markov_codec", "bench_line_strong_n1"):
    d = json.load(open("$O/%s.json" % f))
    print(f, d["value"], d["ms_per_step"], d["ratio"], d["roundtrip_ok"], d["bytes_equal_golden"], {k: v for k, v in d["kernel_ms"].items() if v > 1})
PY
# config 3's per-node load on one GPU: 1e9 bytes = 15 259 blocks (no golden for this stream: round trip + oracle sample)
if [ "${2:-}" = "big" ]; then
  timeout -k 10 900 python bench.py --bytes 1000000000 --steps 2 --warmup 1 --no-cpu > $O/bench_line_1e9.json 2>> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/bench_line_1e9.json')); print('1e9', d['value'], d['ms_per_step'], d['roundtrip_ok'], d['kernel_ms'])"
fi
