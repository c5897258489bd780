# The north-star table in one command on an 8-GPU MI355X node: encode+decode MB/s at 1 / 2 / 4 / 8 GPUs, weak (every GPU its own
# 1e8-byte shard) and strong (ONE corpus cut into contiguous block ranges: 1e8 bytes = BASELINE config 2's enwik8 stand-in,
# 1e9 bytes = config 3's enwik9 stand-in), each line with roundtrip_ok / bytes_equal_golden / ranks_equal_golden / gather_checked.
#   usage: bash tools/scale_table.sh [outdir=gpurun_out/scale] [gpus="1 2 4 8"]
# bench.py --gpus N starts its N ranks itself (one process per GPU, RCCL); nothing here needs more than the repository. A run that
# hangs (a rank missing from the rendezvous, a collective that never completes) ends at bench.py's --deadline with an error line,
# which becomes a row of the table.
set -eo pipefail
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/scale}
G=${2:-1 2 4 8}
mkdir -p "$O"
avail=$(python3 -c "import torch; print(torch.cuda.device_count())")
for n in $G; do
  if [ "$n" -gt "$avail" ]; then echo "skipping N=$n: $avail GPU(s) on this box"; continue; fi
  timeout -k 10 900 python3 bench.py --gpus $n --no-cpu --no-e2e > $O/weak_n$n.json 2> $O/weak_n$n.err || echo "weak N=$n failed (see $O/weak_n$n.err)"
  timeout -k 10 900 python3 bench.py --gpus $n --no-cpu --no-e2e --scaling strong > $O/strong_1e8_n$n.json 2> $O/strong_1e8_n$n.err || echo "strong 1e8 N=$n failed"
  timeout -k 10 1500 python3 bench.py --gpus $n --no-cpu --no-e2e --scaling strong --bytes 1000000000 --steps 2 --warmup 1 > $O/strong_1e9_n$n.json 2> $O/strong_1e9_n$n.err || echo "strong 1e9 N=$n failed"
done
python3 - "$O" <<'PY'
import glob, json, os, sys
rows = []
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception:
        continue
    rows.append((os.path.basename(f)[:-5], d))
base = {}
print(f"{'run':16s} {'GPUs':>4s} {'MB/s':>10s} {'ms/step':>9s} {'vs N=1':>7s}  roundtrip  golden  ranks  gather")
for name, d in rows:
    kind = name.rsplit("_n", 1)[0]
    if d.get("error"):                                   # a run that was given up at its deadline (bench.py --deadline): a row, not a hang
        print(f"{kind:16s} {d['n_gpus']:4d} {'error':>10s}  {d['error']}")
        continue
    if d["n_gpus"] == 1:
        base[kind] = d["value"]
    sp = f"{d['value'] / base[kind]:.2f}x" if d.get("value") and base.get(kind) else "-"
    print(f"{kind:16s} {d['n_gpus']:4d} {d['value'] or 0:10.1f} {d['ms_per_step']:9.2f} {sp:>7s}  {str(d['roundtrip_ok']):9s}  {str(d['bytes_equal_golden']):6s}  {str(d.get('ranks_equal_golden')):5s}  {d.get('gather_checked')}")
PY
