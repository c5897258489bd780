# wall-clock breakdown of the -k64 command lines on the GPU box (1e8 bytes of enwik-shaped text): comp*-gpu -t
set -eo pipefail
cd $GRAFT_REPO_ROOT
python3 -c "
import sys; sys.path.insert(0,'.')
from comprox_amd import corpus
corpus.enwik_like(100_000_000, 8).tofile('/tmp/enwik_like')"
for rep in 1 2; do
for cli in ${1:-comprop-gpu}; do
  s=$(date +%s.%N); comprox_amd/bin/$cli -q -t -k64 e /tmp/enwik_like /tmp/out.$cli; m=$(date +%s.%N)
  comprox_amd/bin/$cli -q -t d /tmp/out.$cli /tmp/back.$cli; e=$(date +%s.%N)
  cmp /tmp/enwik_like /tmp/back.$cli
  python3 -c "print('$cli -k64 run $rep: encode %.3f s, decode %.3f s wall clock of the whole process' % ($m-$s, $e-$m))"
done
done
