cd $GRAFT_REPO_ROOT
for g in 1526 1024 768 512 384 256 192 128 96 64; do
  CRGPU_LZP_GRID=$g python bench.py --steps 2 --warmup 1 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($g, d['kernel_ms']['k_rop_lzp'], d['roundtrip_ok'])"
done
