# round 4, comprolz's 64 KiB searches from the sorted records: the parity tests that touch comprolz, the phases of the two kernels, and
# the bench lines they move.   usage (through gpurun): bash tools/gpu_r04_rolz.sh <tag>
set -eo pipefail
O=gpurun_out/${1:-r04w}
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_rolz.py tests/test_gpu_prepass64.py tests/test_gpu_fuzz.py tests/test_gpu_cli.py -x -q -m gpu > $O/pytest.txt 2>&1
CRGPU_LIB=comprox_amd/libcrgpu_diag.so timeout -k 10 120 python3 tools/rolz_match_profile.py 1526 hard-rings > $O/rolz_rings_profile.txt 2>&1
timeout -k 10 300 python3 bench.py --codec rolz --workload enwik-hard --no-cpu --no-e2e > $O/bench_hard_rolz.json 2> $O/bench_hard_rolz.err
timeout -k 10 300 python3 bench.py --codec rolz --stage codec --no-cpu --no-e2e > $O/bench_rolz_codec.json 2> $O/bench_rolz_codec.err
timeout -k 10 300 python3 bench.py --codec rolz --no-cpu --no-e2e > $O/bench_rolz.json 2> $O/bench_rolz.err
tail -3 $O/pytest.txt
