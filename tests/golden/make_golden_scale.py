#!/usr/bin/env python3
"""Generate tests/golden/golden_scale.json: bench-scale and container-level outputs of the UNMODIFIED reference.

Run in the build container (where oracle/_ref was built from /root/reference):  python tests/golden/make_golden_scale.py
Only sizes and SHA-256 digests are committed; the inputs are regenerated from the seeded generators.

Part "o2" (oracle O2): the bench corpus comprox_amd.corpus.enwik_like(1e8, seed) for seed 8 (the
enwik8 stand-in, rank 0 of the bench) .. 15 (ranks 1..7 of the weak-scaling bench), cut into 64 KiB independent
datablocks, each block through the reference's own functions:
    stage "codec":  reset_models(); lzencode(block)                                 (O2 variant A)
    stage "full":   dicpick(whole shard) + dictionary_load once, then per block
                    dictionary_encode(block) -> reset_models(); lzencode(...)       (O2 variant B = src/main.c:189-194)
and for each the SHA-256 + total size of the per-block outputs laid back to back, at 1 MiB (16 blocks), 16 MiB (256
blocks) and full length (1 526 blocks). bench.py hashes what its timed step produced and prints `bytes_equal_golden`.

Part "o1" (oracle O1): the reference's own cr_main() (src/main.c:89-331) — container, dictionary blob, DEPENDENT blocks
(models carried from block to block) — on seeded streams: `-q -b1 e` on 3 MiB + 12 345 B of text, the default
(`-q e`, 16 MiB blocks) on 33 MiB + 54 321 B of text, and the default on 17 MB of random bytes (the dictionary stage
hands lzencode 16 MiB + 1 bytes, src/cr-diccode.c:208-217). tests/test_gpu_cli.py compares the GPU command lines with them.

The reference leaks its 68 MB (comprop) / 85 MB (comprolz) tables at every reset_models(), so blocks are worked off in
short-lived child processes.
"""
import ctypes
import hashlib
import json
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import crlib  # noqa: E402
from comprox_amd import corpus  # noqa: E402

BLOCK = 65536
N = 100_000_000
CUTS = {"1MiB": 16, "16MiB": 256, "full": None}
CHUNK = 24                                       # blocks per child process


def _quiet():
    """The reference prints a progress line per dictionary call: send stderr of this (child) process to /dev/null."""
    fd = os.open(os.devnull, os.O_WRONLY)
    os.dup2(fd, 2)


def _job(args):
    path, nbytes, codec, stage, dic, first, count = args
    _quiet()
    data = np.memmap(path, dtype=np.uint8, mode="r")
    ref = crlib.Reference(codec)
    if stage == "full":
        ref.dictionary_load(dic, True)
    outs = []
    for b in range(first, first + count):
        blk = data[b * BLOCK:min(nbytes, (b + 1) * BLOCK)].tobytes()
        if stage == "full":
            blk = ref.dictionary_encode(blk)
        outs.append(ref.encode(blk))             # reset_models(); lzencode()
    return first, outs


def _dicpick(path):
    _quiet()
    ref = crlib.Reference("rop")
    dic = ref.dicpick(open(path, "rb").read())
    return dic, ref.dictionary_load(dic, True)


def shard_range(nb, g, r):
    """crgpu_shard_range (csrc/crgpu_multi.hip): rank r of g owns [r * ceil(nb / g), (r + 1) * ceil(nb / g))."""
    per = (nb + g - 1) // g
    lo = min(nb, per * r)
    return lo, min(nb, lo + per) - lo


def o2_record(pool, path, nbytes, codec, stage, dic, rank_counts=()):
    nb = (nbytes + BLOCK - 1) // BLOCK
    jobs = [(path, nbytes, codec, stage, dic, f, min(CHUNK, nb - f)) for f in range(0, nb, CHUNK)]
    parts = dict(pool.imap_unordered(_job, jobs))
    h = hashlib.sha256()
    hs = hashlib.sha256()
    total = 0
    rec = {"n": nbytes, "block": BLOCK, "blocks": nb, "cuts": {}}
    done = 0
    # the run of every rank of a G-GPU strong-scaling job: its blocks' outputs back to back (what k_pack leaves on that GPU)
    runs = {g: [[shard_range(nb, g, r), hashlib.sha256(), 0] for r in range(g)] for g in rank_counts}
    for f in range(0, nb, CHUNK):
        for out in parts[f]:
            h.update(out)
            hs.update(len(out).to_bytes(4, "little"))
            total += len(out)
            for g, rr in runs.items():
                r = done // ((nb + g - 1) // g)
                rr[r][1].update(out)
                rr[r][2] += len(out)
            done += 1
            for name, k in CUTS.items():
                if done == (k or nb):
                    rec["cuts"][name] = {"blocks": done, "size": total, "sha256": h.copy().hexdigest()}
    rec["sizes_sha256"] = hs.hexdigest()          # the uint32 LE per-block sizes, the table the gather exchanges
    if runs:
        rec["ranks"] = {str(g): [{"first": lo, "count": cnt, "size": sz, "sha256": hh.hexdigest()} for (lo, cnt), hh, sz in rr]
                        for g, rr in runs.items()}
    return rec


def _cr_main(codec, args, cwd):
    """cr_main() of the reference library in a child process (file-scope state, fclose(stderr) under -q)."""
    code = ("import ctypes,sys\n"
            f"L=ctypes.CDLL({crlib.REF_LIBS[codec]!r})\n"
            "argv=(ctypes.c_char_p*(len(sys.argv)))(*[a.encode() for a in sys.argv[1:]],None)\n"
            "sys.exit(L.cr_main(len(sys.argv)-1, argv) & 255)\n")
    r = subprocess.run([sys.executable, "-c", code, "comp" + codec] + args, cwd=cwd, capture_output=True)
    assert r.returncode == 0, (codec, args, r.stderr[-400:])


def o1_records():
    cases = {
        "text_b1": (lambda: crlib.gen_text(3 * 1048576 + 12345, 8), ["-q", "-b1", "e"]),
        "text_default": (lambda: crlib.gen_text(33 * 1048576 + 54321, 8), ["-q", "e"]),
        "rand_default": (lambda: crlib.gen_rand(17_000_000, seed=5), ["-q", "e"]),
    }
    out = {}
    with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
        for name, (gen, sw) in cases.items():
            data = gen()
            open(os.path.join(d, "in"), "wb").write(data)
            rec = {"n": len(data), "in_sha256": crlib.sha(data), "switches": sw}
            for codec in ("rop", "rox", "rolz"):
                _cr_main(codec, sw + ["in", "out." + codec], d)
                enc = open(os.path.join(d, "out." + codec), "rb").read()
                _cr_main(codec, ["-q", "d", "out." + codec, "back"], d)
                assert open(os.path.join(d, "back"), "rb").read() == data, (name, codec)
                rec[codec] = {"size": len(enc), "sha256": crlib.sha(enc)}
                print("o1", name, codec, len(enc), flush=True)
            out[name] = rec
    return out


def add_1e9():
    """BASELINE config 3: enwik_like(1e9, seed 9) = 15 259 blocks, the whole per-block path of
    src/main.c:174-206 with reset_models() per block, plus the run of every rank for G = 2, 4, 8 contiguous ranges, so
    that each rank of a strong-scaling job can check its own bytes without a gather. Merged into golden_scale.json."""
    n, seed = 1_000_000_000, 9
    out_path = os.path.join(HERE, "golden_scale.json")
    gold = json.load(open(out_path))
    key = f"enwik_like_1e9_seed{seed}"
    rec = gold["o2"].get(key, {})
    path = f"/dev/shm/crgold_{seed}_1e9.bin"
    corpus.enwik_like(n, seed).tofile(path)
    todo = sys.argv[2].split(",") if len(sys.argv) > 2 else ["rop/full", "rop/codec", "rox/full", "rolz/full"]
    try:
        with mp.get_context("fork").Pool(int(os.environ.get("CRGOLD_WORKERS", "8")), maxtasksperchild=1) as pool:
            dic, nword = pool.apply(_dicpick, (path,))
            rec["dictionary"] = {"size": len(dic), "sha256": crlib.sha(dic), "words": nword}
            for cs in todo:
                codec, stage = cs.split("/")
                rec[cs] = o2_record(pool, path, n, codec, stage, dic, rank_counts=(2, 4, 8))
                print("o2 1e9", cs, rec[cs]["cuts"]["full"], flush=True)
                gold["o2"][key] = rec
                with open(out_path, "w") as f:
                    json.dump(gold, f, indent=1, sort_keys=True)
    finally:
        os.unlink(path)


def add_markov():
    """BASELINE config 5: the order-2 Markov stream, blocks 0 .. 255 (16 MiB) through the reference's
    per-block path. The stream has no file to pick a dictionary from, so the per-file dictionary is the reference's
    dicpick over the stream's first 2^28 bytes (blocks 0 .. 4 095) — what bench.py --workload markov does too. Merged
    into golden_scale.json under o2/markov2_first256."""
    nb_dic, nb = 4096, 256
    out_path = os.path.join(HERE, "golden_scale.json")
    gold = json.load(open(out_path))
    data = corpus.markov2_blocks(nb_dic, 0, BLOCK, device="cpu").numpy().reshape(-1)
    for b in (0, 1, 255, 4095):                  # the vectorised generator against the scalar definition
        assert data[b * BLOCK:(b + 1) * BLOCK].tobytes() == corpus.markov2(BLOCK, b).tobytes(), b
    path = "/dev/shm/crgold_markov.bin"
    data.tofile(path)
    try:
        with mp.get_context("fork").Pool(int(os.environ.get("CRGOLD_WORKERS", "8")), maxtasksperchild=1) as pool:
            dic, nword = pool.apply(_dicpick, (path,))
            rec = {"dictionary": {"size": len(dic), "sha256": crlib.sha(dic), "words": nword, "picked_from_bytes": nb_dic * BLOCK},
                   "in_sha256": crlib.sha(data[:nb * BLOCK].tobytes())}
            for codec in ("rop", "rox", "rolz"):
                for stage in ("codec", "full"):
                    rec[f"{codec}/{stage}"] = o2_record(pool, path, nb * BLOCK, codec, stage, dic)
                    print("o2 markov", codec, stage, rec[f"{codec}/{stage}"]["cuts"]["full"], flush=True)
    finally:
        os.unlink(path)
    gold["o2"]["markov2_first256"] = rec
    with open(out_path, "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)


def add_hard():
    """The harder corpus (corpus.enwik_hard(1e8, seed 8): flatter vocabulary, numbers, URLs, markup — the dictionary stage
    leaves ~56 % instead of 36 %), cut like the headline corpus: the reference's per-block outputs at 1 MiB / 16 MiB / full
    length for both stages. Merged into golden_scale.json under o2/enwik_hard_1e8_seed8."""
    seed = 8
    out_path = os.path.join(HERE, "golden_scale.json")
    gold = json.load(open(out_path))
    path = f"/dev/shm/crgold_hard_{seed}.bin"
    corpus.enwik_hard(N, seed).tofile(path)
    try:
        with mp.get_context("fork").Pool(int(os.environ.get("CRGOLD_WORKERS", "8")), maxtasksperchild=1) as pool:
            dic, nword = pool.apply(_dicpick, (path,))
            rec = {"dictionary": {"size": len(dic), "sha256": crlib.sha(dic), "words": nword},
                   "in_sha256": crlib.sha(open(path, "rb").read())}
            for codec in ("rop", "rox", "rolz"):
                for stage in ("codec", "full"):
                    if codec != "rop" and stage == "codec":
                        continue
                    rec[f"{codec}/{stage}"] = o2_record(pool, path, N, codec, stage, dic)
                    print("o2 hard", codec, stage, rec[f"{codec}/{stage}"]["cuts"]["full"], flush=True)
    finally:
        os.unlink(path)
    gold["o2"][f"enwik_hard_1e8_seed{seed}"] = rec
    with open(out_path, "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "hard":
        return add_hard()
    if len(sys.argv) > 1 and sys.argv[1] == "1e9":
        return add_1e9()
    if len(sys.argv) > 1 and sys.argv[1] == "markov":
        return add_markov()
    seeds = [int(s) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(8, 16))
    gold = {"_about": "outputs of the unmodified reference (oracle/_ref) on the bench corpus and of its cr_main() — see make_golden_scale.py",
            "o2": {}, "o1": o1_records()}
    with mp.get_context("fork").Pool(8, maxtasksperchild=1) as pool:
        for seed in seeds:
            path = f"/dev/shm/crgold_{seed}.bin"
            corpus.enwik_like(N, seed).tofile(path)
            try:
                dic, nword = pool.apply(_dicpick, (path,))
                rec = {"dictionary": {"size": len(dic), "sha256": crlib.sha(dic), "words": nword}}
                for codec in ("rop", "rox", "rolz"):
                    if codec != "rop" and seed != 8:
                        continue                     # ranks > 0 only run the default codec in the bench
                    for stage in ("codec", "full"):
                        rec[f"{codec}/{stage}"] = o2_record(pool, path, N, codec, stage, dic)
                        print("o2", seed, codec, stage, rec[f"{codec}/{stage}"]["cuts"]["full"], flush=True)
                gold["o2"][f"enwik_like_1e8_seed{seed}"] = rec
            finally:
                os.unlink(path)
    with open(os.path.join(HERE, "golden_scale.json"), "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
