# how the decoder's rate follows its residency (diagnostic build -DCR_DEC_OCC_EXP: $CRGPU_DEC_LDS_PAD bytes of dynamic LDS per
# one-wave decoder workgroup cap the workgroups a CU holds at 160 KiB / pad), and where its wave cycles go (SQ counters).
# usage (through gpurun): bash tools/dec_occupancy.sh <tag>
set -eo pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-occ}
mkdir -p $O
for pad in 0 16384 20480 40960; do
  CRGPU_LIB=$R/comprox_amd/libcrgpu_diag.so CRGPU_DEC_LDS_PAD=$pad timeout -k 10 300 python3 bench.py --no-cpu --no-e2e --no-overlap --steps 5 --warmup 2 > $O/pad_$pad.json 2> $O/pad_$pad.err
  python3 - <<PY
import json
d = json.loads([l for l in open('$O/pad_$pad.json') if l.startswith('{')][0])
print('pad', $pad, 'ms_per_step', d['ms_per_step'], {k: round(v, 2) for k, v in d['kernel_ms'].items() if 'decode' in k})
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-overlap > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
for d in ('pmc_sq', 'pmc_sq2'):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob('$O/%s/**/*counter_collection.csv' % d, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in acc.items():
        if 'decode' in k or 'k_rop_o2' in k or 'k_rop_lzp' in k:
            print(k, dict(v))
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
