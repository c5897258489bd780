# end-to-end rate of the drop-in command lines on the GPU box: 1e8 bytes of enwik-shaped text, 64 KiB independent blocks
set -eo pipefail
cd $GRAFT_REPO_ROOT
python3 -c "
import sys; sys.path.insert(0,'.')
from comprox_amd import corpus
corpus.enwik_like(100_000_000, 8).tofile('/tmp/enwik_like')"
for cli in comprop-gpu comprox-gpu comprolz-gpu; do
  s=$(date +%s.%N); comprox_amd/bin/$cli -q -k64 e /tmp/enwik_like /tmp/out.$cli; m=$(date +%s.%N)
  comprox_amd/bin/$cli -q d /tmp/out.$cli /tmp/back.$cli; e=$(date +%s.%N)
  cmp /tmp/enwik_like /tmp/back.$cli
  python3 -c "print('$cli -k64: %d -> %d bytes, encode %.2f s (%.0f MB/s), decode %.2f s (%.0f MB/s) wall clock incl. file I/O, dictionary stage and process start' % (100000000, __import__('os').path.getsize('/tmp/out.$cli'), $m-$s, 100/($m-$s), $e-$m, 100/($e-$m)))"
done
