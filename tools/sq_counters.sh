# where every kernel of the bench step spends its wave cycles: rocprofv3 --kernel-include-regex "^k_" --pmc SQ counters (two passes), one line per kernel.
# usage (through gpurun): bash tools/sq_counters.sh <tag> [bench.py arguments]
set -eo pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-sq}
shift || true
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-include-regex "^k_" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap "$@" > /dev/null 2>&1
rocprofv3 --kernel-include-regex "^k_" --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap "$@" > /dev/null 2>&1
rocprofv3 --kernel-include-regex "^k_" --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc_sq3 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap "$@" > /dev/null 2>&1 || true
cd $R
python3 - <<PY > $O/sq_counters.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ('pmc_sq', 'pmc_sq2', 'pmc_sq3'):
    for f in glob.glob('$O/%s/**/*counter_collection.csv' % d, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']] += float(r['Counter_Value'])
print('# SQ counters of two launches of every kernel (bench.py --steps 1 --warmup 1 $*); WAVE_CYCLES / WAIT / ACTIVE in quad-cycles')
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    if not k.startswith('k_'): continue
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    print('%-18s waves %6d  wave_cycles %8.3fe9  parked %4.1f%%  issuing %4.1f%%  issue-stalled %4.1f%% (LDS %4.1f%%)  VALU %7.1fM SALU %7.1fM VMEM rd/wr %6.1fM/%6.1fM LDS %6.1fM  bank-conflict cycles %7.1fM of %7.1fM active' % (
        k, v.get('SQ_WAVES', 0), wc / 1e9, 100 * v.get('SQ_WAIT_ANY', 0) / wc, 100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc, 100 * v.get('SQ_WAIT_INST_ANY', 0) / wc,
        100 * v.get('SQ_WAIT_INST_LDS', 0) / wc, v.get('SQ_INSTS_VALU', 0) / 1e6, v.get('SQ_INSTS_SALU', 0) / 1e6, v.get('SQ_INSTS_VMEM_RD', 0) / 1e6, v.get('SQ_INSTS_VMEM_WR', 0) / 1e6,
        v.get('SQ_INSTS_LDS', 0) / 1e6, v.get('SQ_LDS_BANK_CONFLICT', 0) / 1e6, v.get('SQ_LDS_IDX_ACTIVE', 0) / 1e6))
PY
cat $O/sq_counters.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/pmc_sq $O/pmc_sq2 $O/pmc_sq3
