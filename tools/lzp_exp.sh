# LZP pre-pass experiment on the GPU box: kernel time per sweep variant (2, 3, 4 = no phase B / no phase A: timing only)
set -eo pipefail
cd $GRAFT_REPO_ROOT
for k in ${1:-0 1 2 3}; do
  CRGPU_CFLAGS=-DCR_LZP_SWEEP=$k python -m comprox_amd.build --force > /dev/null 2>&1
  echo "== CR_LZP_SWEEP=$k"
  timeout -k 10 200 python bench.py --no-cpu --steps 2 --warmup 1 | python3 -c "import sys,json; d=json.load(sys.stdin); print(d['kernel_ms']['k_rop_lzp'], d['roundtrip_ok'])"
done
