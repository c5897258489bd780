# decoder work loop on the GPU box: parity tests of the comprop codec, then decode timings per variant
set -eo pipefail
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_rop.py tests/test_gpu_robust.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/dec_bench.py ${1:-v5} ${2:-1526,64,1}
