# rate of the STOCK drop-in path on the GPU box (no -k: the reference's container, default 16 MiB blocks, models carried from
# block to block), for the reference's own front-ends relinked against libcrgpu.so (oracle/_ref/bin/comp*-dropin) and the
# unmodified reference (oracle/_ref/lib*_ref.so, cr_main on this host's CPU, one pinned thread) on the same files:
#   8 MiB  = one block: it starts from fresh models, so the shims run the batched kernel pipeline on it;
#   33 MiB = three blocks: blocks two and three continue the previous block's models — one dependent chain per call, which is
#            the one shape a GPU wave cannot win (the fresh first block is coded fast and once more by the model-carrying coder).
set -eo pipefail
cd $GRAFT_REPO_ROOT
for N in ${1:-8388608 34603008}; do
python3 -c "
import sys; sys.path.insert(0,'.')
from comprox_amd import corpus
corpus.enwik_like($N, 8).tofile('/tmp/enwik_like_stock')"
for c in rop rox rolz; do
  exe=oracle/_ref/bin/comp$c-dropin
  s=$(date +%s.%N); timeout -k 10 500 $exe -q e /tmp/enwik_like_stock /tmp/sout.$c; m=$(date +%s.%N)
  timeout -k 10 500 $exe -q d /tmp/sout.$c /tmp/sback.$c; e=$(date +%s.%N)
  cmp /tmp/enwik_like_stock /tmp/sback.$c
  python3 - <<PY
import ctypes, os, sys, time
sys.path.insert(0, 'tests')
import crlib
n, codec = $N, '$c'
L = ctypes.CDLL(crlib.REF_LIBS[codec])
def run(args):
    pid = os.fork()
    if pid == 0:
        try: os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
        except Exception: pass
        argv = (ctypes.c_char_p * (len(args) + 1))(*[a.encode() for a in args], None)
        os._exit(L.cr_main(len(args), argv) & 255)
    assert os.waitpid(pid, 0)[1] == 0
t0 = time.time(); run(['comp' + codec, '-q', 'e', '/tmp/enwik_like_stock', '/tmp/rout.' + codec]); t1 = time.time()
run(['comp' + codec, '-q', 'd', '/tmp/rout.' + codec, '/tmp/rback.' + codec]); t2 = time.time()
same = open('/tmp/rout.' + codec, 'rb').read() == open('/tmp/sout.' + codec, 'rb').read()
ge, gd = $m - $s, $e - $m
print('comp%s-dropin (reference main.c + front-end on libcrgpu.so), %d bytes, stock container: encode %.2f s (%.1f MB/s), decode %.2f s (%.1f MB/s); '
      'unmodified reference on this host, 1 thread: encode %.2f s (%.1f MB/s), decode %.2f s (%.1f MB/s); same file: %s'
      % (codec, n, ge, n / 1e6 / ge, gd, n / 1e6 / gd, t1 - t0, n / 1e6 / (t1 - t0), t2 - t1, n / 1e6 / (t2 - t1), same))
PY
done
done
