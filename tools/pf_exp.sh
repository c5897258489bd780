# decoder experiments on the GPU box: one dec_bench run (raw 64 KiB text, then the bench's dictionary-stage stream) per
# prebuilt variant library comprox_amd/libcrgpu_<tag>.so (built here with CRGPU_CFLAGS=... python -m comprox_amd.build)
# usage: bash tools/pf_exp.sh <out-tag> <lib-tag> [<lib-tag> ...]
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift
mkdir -p $O
for t in "$@"; do
  echo "== $t" | tee -a $O/pf_exp.txt
  CRGPU_LIB=$GRAFT_REPO_ROOT/comprox_amd/libcrgpu_$t.so timeout -k 10 300 python3 tools/dec_bench.py v5 1526,1 2>&1 | grep blocks= | tee -a $O/pf_exp.txt || exit 1
  CRGPU_LIB=$GRAFT_REPO_ROOT/comprox_amd/libcrgpu_$t.so timeout -k 10 300 python3 tools/dec_bench.py v5 1526 full 2>&1 | grep blocks= | tee -a $O/pf_exp.txt || exit 1
done
