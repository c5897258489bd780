# rate of the STOCK drop-in path on the GPU box (no -k: the reference's container, default 16 MiB DEPENDENT blocks, models
# carried from block to block): one wavefront per lzencode / lzdecode call. Parity-only; this prints what that costs.
set -eo pipefail
cd $GRAFT_REPO_ROOT
N=${1:-33554432}
python3 -c "
import sys; sys.path.insert(0,'.')
from comprox_amd import corpus
corpus.enwik_like($N, 8).tofile('/tmp/enwik_like_stock')"
for cli in comprop-gpu comprox-gpu comprolz-gpu; do
  s=$(date +%s.%N); timeout -k 10 500 comprox_amd/bin/$cli -q e /tmp/enwik_like_stock /tmp/sout.$cli; m=$(date +%s.%N)
  timeout -k 10 500 comprox_amd/bin/$cli -q d /tmp/sout.$cli /tmp/sback.$cli; e=$(date +%s.%N)
  cmp /tmp/enwik_like_stock /tmp/sback.$cli
  python3 -c "print('$cli (stock container, 16 MiB dependent blocks): %d -> %d bytes, encode %.2f s (%.2f MB/s), decode %.2f s (%.2f MB/s) wall clock incl. file I/O, dictionary stage and process start' % ($N, __import__('os').path.getsize('/tmp/sout.$cli'), $m-$s, $N/1e6/($m-$s), $e-$m, $N/1e6/($e-$m)))"
done
