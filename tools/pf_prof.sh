# in-kernel stamps of the decoder step (tools/dec_profile.py) per prebuilt -DCR_V5_PROF=1 variant library
# usage: bash tools/pf_prof.sh <out-tag> <lib-tag> ...
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift
mkdir -p $O
for t in "$@"; do
  echo "== $t" | tee -a $O/pf_prof.txt
  CRGPU_LIB=$GRAFT_REPO_ROOT/comprox_amd/libcrgpu_$t.so timeout -k 10 300 python3 tools/dec_profile.py 1526 1 2>&1 | grep -v Warn | tee -a $O/pf_prof.txt || exit 1
done
