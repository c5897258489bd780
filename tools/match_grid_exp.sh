cd $GRAFT_REPO_ROOT
for c in rox rolz; do for g in 1526 1024 768 512 384 256; do
  CRGPU_MATCH_GRID=$g python bench.py --steps 2 --warmup 1 --no-cpu --codec $c 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[x for x in d['kernel_ms'] if x.endswith('_match')][0]; print('$c', $g, d['kernel_ms'][k], d['roundtrip_ok'])"
done; done
