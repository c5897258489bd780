cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rop.py tests/test_gpu_robust.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_quick -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/bench_quick.log 2>&1
grep '^{' $R/gpurun_out/bench_quick.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['encode_ms'], d['decode_ms'], d['roundtrip_ok'])"
head -10 $R/gpurun_out/prof_quick/*/*kernel_stats.csv | cut -c1-100
