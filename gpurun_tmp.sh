cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -type f | head; 
