# kernel timeline of bench.py's two-steps-in-flight measurement (rocprofv3 --kernel-trace): start / end of every kernel of the
# last steps relative to the first of them, to see what really runs beside the decoder.
# usage (through gpurun): bash tools/overlap_trace.sh <tag>
set -eo pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-ovt}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu --no-e2e ${2:+--codec $2} > $O/line.json 2> $O/err.txt
cd $R
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob('$O/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0], r.get('Queue_Id', ''), r.get('Stream_Id', '')))
rows.sort()
last = rows[-int('${3:-40}'):]
t0 = last[0][0]
for s, e, k, q, st in last:
    print('%9.3f %9.3f  %7.3f ms  q%-3s s%-3s %s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, st, k))
PY
find $O -name "*.db" -delete
