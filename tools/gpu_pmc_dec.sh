# SQ instruction mix of the decode kernels for ONE block (tools/dec_bench.py <variants> 1), counters only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-v5,v4}
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_dec/$tag -- python3 $R/tools/dec_bench.py $V 1 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
R = os.environ["GRAFT_REPO_ROOT"]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(R + "/gpurun_out/pmc_dec/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_rop_decode"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(acc[k].items())})
PY
find $R/gpurun_out/pmc_dec -name "*.csv" -size +1M -delete
