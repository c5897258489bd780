"""GPU rehearsal of bench.py's multi-rank paths on a one-GPU box: `python bench.py --gpus 2` starts two ranks itself;
both sit on GPU 0 and exchange through gloo (RCCL refuses two ranks on one device), everything else — block ranges, the
size all_gather inside the step, the strong-scaling gather of the packed runs to rank 0, the hash of the assembled
stream against the reference's — is the code the 8-GPU run executes."""
import json
import os
import subprocess
import sys

import pytest

import crlib

pytestmark = pytest.mark.gpu


def run_bench(args):
    env = dict(os.environ, CRBENCH_BACKEND="gloo", CRBENCH_ONE_GPU="1")
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(crlib.ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_strong_scaling_assembles_the_reference_stream():
    # (codec stage: the golden's 16 MiB cut; on the full path a 16 MiB file would pick another dictionary than the 1e8-byte one)
    d = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--scaling", "strong", "--bytes", "16777216", "--stage", "codec", "--no-cpu"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["roundtrip_ok"] is True
    assert d["gather_checked"] is True and d["bytes_equal_golden"] is True          # 16 MiB cut of seed 8: golden_scale.json
    assert d["config"]["total_bytes"] == 16777216 and d["value"] > 0
    d = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--scaling", "strong", "--bytes", "4194304", "--no-cpu"])
    assert d["n_gpus"] == 2 and d["roundtrip_ok"] is True and d["gather_checked"] is True and d["bytes_equal_golden"] is None


def test_two_ranks_weak_scaling():
    d = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--bytes", "1048576", "--stage", "codec", "--no-cpu"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["roundtrip_ok"] is True
    assert d["config"]["total_bytes"] == 2 * 1048576
    assert d["bytes_equal_golden"] is True                          # each rank's shard (seeds 8 and 9) against its own 1 MiB cut
