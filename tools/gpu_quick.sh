cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_rop.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_chains3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/bench_chains3.log 2>&1
grep '^{' $R/gpurun_out/bench_chains3.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['encode_ms'], d['decode_ms'], d['roundtrip_ok'])"
head -8 $R/gpurun_out/prof_chains3/*/*kernel_stats.csv | cut -c1-100
