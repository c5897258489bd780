# where the stock (no -k) command lines spend their time on the two slow goldens of tests/test_gpu_cli.py: comp*-gpu -t marks
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import crlib
open('/tmp/text33','wb').write(crlib.gen_text(33 * 1048576 + 54321, 8))
open('/tmp/rand17','wb').write(crlib.gen_rand(17_000_000, seed=5))
PY
for f in text33 rand17; do
  for cli in ${1:-comprop-gpu}; do
    s=$(date +%s.%N); comprox_amd/bin/$cli -q -t e /tmp/$f /tmp/$f.out 2> /tmp/$f.e.marks; m=$(date +%s.%N)
    comprox_amd/bin/$cli -q -t d /tmp/$f.out /tmp/$f.back 2> /tmp/$f.d.marks; e=$(date +%s.%N)
    cmp /tmp/$f /tmp/$f.back && echo same
    python3 -c "print('$cli $f: encode %.2f s, decode %.2f s' % ($m-$s, $e-$m))"
    cat /tmp/$f.e.marks /tmp/$f.d.marks
  done
done
