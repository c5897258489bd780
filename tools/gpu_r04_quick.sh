# round 4 inner loop: parity of the three codecs' pre-pass kernels + the bench lines' kernel times
# usage: tools/gpu_r04_quick.sh <tag> [extra pytest files]
cd $GRAFT_REPO_ROOT
T=gpurun_out/$1; mkdir -p $T
timeout -k 10 600 python -m pytest tests/test_gpu_rop.py tests/test_gpu_rox.py tests/test_gpu_rolz.py $2 -x -q -m gpu > $T/pytest.txt 2>&1 || { tail -30 $T/pytest.txt; exit 1; }
tail -2 $T/pytest.txt
for c in rop rox rolz; do
  timeout -k 10 300 python bench.py --codec $c --steps 5 --warmup 2 --no-cpu --no-e2e --no-overlap > $T/bench_$c.json 2> $T/bench_$c.err || { tail -20 $T/bench_$c.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$T/bench_$c.json"))
print("$c", d["value"], d["ms_per_step"], d["roundtrip_ok"], d["bytes_equal_golden"], {k: round(v,3) for k,v in d["kernel_ms"].items() if v > 0.3})
PY
done
