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


def run_bench(args, **more_env):
    env = dict(os.environ, CRBENCH_BACKEND="gloo", CRBENCH_ONE_GPU="1", **more_env)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(crlib.ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, "\n".join(l for l in p.stderr.splitlines() if "rank" in l or "Error" in l)[-3000:]
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


def test_config3_enwik9_shaped_stream_on_one_gpu_equals_the_reference():
    """BASELINE config 3's whole load on ONE GPU: enwik_like(1e9, seed 9) = 15 259 blocks through the timed step; the
    packed payloads must hash to what the unmodified reference produced (golden_scale.json, enwik_like_1e9_seed9)."""
    d = run_bench(["--bytes", "1000000000", "--steps", "1", "--warmup", "0", "--no-cpu"])
    assert d["n_gpus"] == 1 and d["roundtrip_ok"] is True and d["bytes_equal_golden"] is True
    assert d["config"]["blocks_per_gpu"] == 15259 and d["compressed_bytes"] == 234143229


def test_config3_two_ranks_check_their_own_runs():
    """The sharded form of config 3 (strong scaling, contiguous block ranges): every rank compares the run it packed with
    the reference's bytes for ITS range (golden `ranks`), and rank 0 the gathered stream."""
    # (two ranks share ONE card here: each takes 6 instead of 16 model arenas per CU so that both fit its 288 GB)
    d = run_bench(["--gpus", "2", "--bytes", "1000000000", "--scaling", "strong", "--steps", "1", "--warmup", "0", "--no-cpu"], CRGPU_WG_PER_CU="6")
    assert d["n_gpus"] == 2 and d["roundtrip_ok"] is True
    assert d["ranks_equal_golden"] is True and d["gather_checked"] is True and d["bytes_equal_golden"] is True


def test_config5_markov_stream_in_batches():
    """BASELINE config 5's stream worked off in batches (1 GiB here, four batches of 4 096 blocks; the 16 GiB run is
    `bench.py --workload markov --bytes 17179869184`): every batch's round trip is checked inside the step and blocks
    0 .. 255 against the reference's bytes (golden markov2_first256)."""
    for stage in ("codec", "full"):
        d = run_bench(["--workload", "markov", "--bytes", str(1 << 30), "--batch-blocks", "4096", "--stage", stage,
                       "--steps", "1", "--warmup", "0", "--no-cpu"])
        assert d["batches_per_step"] == 4 and d["roundtrip_ok"] is True and d["bytes_equal_golden"] is True, stage
        assert d["config"]["blocks_per_gpu"] == 16384


def test_harder_corpus_line_equals_the_reference():
    """--workload enwik-hard (corpus sensitivity: the dictionary stage leaves ~56 % of the bytes, most blocks exceed the small LDS
    pre-pass's 28 672 bytes and take the 64 KiB one, k_rop_lzp_lds64, since round 4; none goes to the table sweep): the timed
    step's bytes == the reference's (golden enwik_hard_1e8_seed8)."""
    d = run_bench(["--workload", "enwik-hard", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-e2e"])
    assert d["roundtrip_ok"] is True and d["bytes_equal_golden"] is True
    assert d["paths"]["prepass_lds_64k_blocks"] > 1000 and d["paths"]["prepass_table_sweep_blocks"] == 0 and d["dictionary_stage_bytes"] > 50_000_000


def test_two_steps_in_flight_decode_their_own_encodes():
    """The extra measurement of the bench line: a second context on a second stream encodes step i + 1 while the first one
    decodes step i (two buffer sets). Every decode must have read what its own encode packed: the round trip holds and both
    buffer sets end up with the bytes the serial steps produced (= the reference's: bytes_equal_golden)."""
    for codec in ("rop", "rox", "rolz"):
        d = run_bench(["--bytes", "16777216", "--stage", "codec", "--steps", "4", "--warmup", "1", "--no-cpu", "--no-e2e", "--codec", codec])
        assert d["roundtrip_ok"] is True and (codec != "rop" or d["bytes_equal_golden"] is True)
        o = d["two_steps_in_flight"]
        assert o["roundtrip_ok"] is True and o["packed_bytes_equal_the_serial_steps"] is True and o["steps"] == 4 and o["value"] > 0
