set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lz2exp
for e in 0 1 2 3; do
  if [ $e = 0 ]; then export CRGPU_CFLAGS="-DCR_DUMMY_EXP=1"; else export CRGPU_CFLAGS="-DCR_LZ2_EXP=$e"; fi
  python3 -m comprox_amd.build > /dev/null 2>&1
  CRGPU_LIB=$PWD/comprox_amd/libcrgpu_diag.so timeout -k 10 200 python3 bench.py --no-cpu --no-e2e --no-overlap --steps 3 --warmup 1 > gpurun_out/lz2exp/l$e.json 2> gpurun_out/lz2exp/e$e.txt
  python3 -c "
import json
try:
    d=json.loads(open('gpurun_out/lz2exp/l$e.json').read().strip().splitlines()[-1]); print('EXP', $e, {k: round(v,2) for k,v in d['kernel_ms'].items() if 'lzp' in k or 'links' in k}, d['roundtrip_ok'])
except Exception as ex: print('EXP', $e, 'no line', ex)"
done
