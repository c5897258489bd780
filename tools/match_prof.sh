# match-token profile of the assembly decoder on the GPU box: one diagnostic build per probe (crgpu_rop5.h, CR_V5_PROF=k)
set -eo pipefail
cd $GRAFT_REPO_ROOT
for k in ${1:-2 3 4 5}; do
  CRGPU_CFLAGS=-DCR_V5_PROF=$k python -m comprox_amd.build --force > /dev/null 2>&1      # -> comprox_amd/libcrgpu_diag.so
  echo "== CR_V5_PROF=$k"
  CRGPU_LIB=$GRAFT_REPO_ROOT/comprox_amd/libcrgpu_diag.so timeout -k 10 200 python tools/dec_profile.py ${2:-1526}
done
