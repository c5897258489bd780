# SQ instruction mix / wait counters for the codec kernels (separate passes, counters only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_sq/$tag -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
R = os.environ["GRAFT_REPO_ROOT"]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(R + "/gpurun_out/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(acc[k].items())})
PY
