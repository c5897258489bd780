# k_dict_match: positions per workgroup (diagnostic builds -DCR_DM_CHUNK=...)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dmexp
for e in 1024u 2048u 4096u 8192u; do
  export CRGPU_CFLAGS="-DCR_DM_CHUNK=$e"
  python3 -m comprox_amd.build > /dev/null 2>&1
  CRGPU_LIB=$PWD/comprox_amd/libcrgpu_diag.so timeout -k 10 200 python3 bench.py --no-cpu --no-e2e --no-overlap --steps 5 --warmup 1 > gpurun_out/dmexp/l$e.json 2> gpurun_out/dmexp/e$e.txt
  python3 -c "
import json
d=json.loads(open('gpurun_out/dmexp/l$e.json').read().strip().splitlines()[-1]); print('CHUNK', '$e', {k: round(v,3) for k,v in d['kernel_ms'].items() if 'dict' in k}, d['ms_per_step'], d['roundtrip_ok'], d['bytes_equal_golden'])"
done
