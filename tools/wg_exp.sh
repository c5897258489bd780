cd $GRAFT_REPO_ROOT
for w in 8 4 2 1; do
  echo "WG_PER_CU=$w"
  CRGPU_WG_PER_CU=$w python bench.py --steps 2 --warmup 1 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_ms'], d['value'])"
done
