cd $GRAFT_REPO_ROOT
for pad in 0 30 94; do
  echo "arena pad ${pad} MB per workgroup"
  CRGPU_ARENA_PAD_MB=$pad timeout -k 10 200 python tools/dec_bench.py v5 1526
done
