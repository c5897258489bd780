#!/usr/bin/env python3
"""bench.py — encode+decode throughput of the MI355X block codec on an enwik8-shaped stream.

Workload (BASELINE.json configs[1]): 100 000 000 bytes of enwik-shaped text per GPU (enwik8 is not on disk;
comprox_amd.corpus.enwik_like(1e8, seed 8+rank), or $ENWIK8 when present), cut into independent 64 KiB datablocks
(1 526 blocks), comprop codec (LZP + PPM + range coder). A "step" is the reference's per-block path for every block,
there and back (src/main.c:189-205 and :277-281), with the input already resident in HBM:

    dictionary_encode -> lzencode -> k_pack (payloads back to back) -> [size all-gather] -> lzdecode -> dictionary_decode

(`--stage codec` leaves the dictionary stage out: lzencode / lzdecode only, round 1's step.) The per-file dictionary
(dicpick, a host pass that runs once per file, src/main.c:156-171) is built before the timed region.
value = uncompressed bytes of all ranks / max-over-ranks step time.

N > 1: `python bench.py --gpus N` starts N ranks itself (a child `torch.distributed.run`, started before this process
touches the GPU); under `torch.distributed.run` it is one of the ranks. One process per GPU, backend nccl = RCCL.
Blocks are independent, so ranks code disjoint block ranges with no data-path collective; the one exchange is the
all_gather of the per-block output sizes (the table a container writer needs), inside the timed step.
  --scaling weak   (default) every rank its own 1e8-byte shard (seed 8 + rank)
  --scaling strong ONE corpus (seed 8, --bytes in total) cut into contiguous block ranges (comprox_amd.shard.partition);
                   rank 0 receives every rank's packed payloads after the timed region, assembles the stream in block
                   order and hashes it.
What was timed is checked: the round trip must reproduce the input (else the run FAILS), and the SHA-256 of the
concatenated payloads is compared with the unmodified reference's (tests/golden/golden_scale.json) -> bytes_equal_golden.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLOCK = 65536
SHARD_BYTES = 100_000_000
HBM_PEAK_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bytes", type=int, default=0, help="uncompressed bytes per GPU (weak) or in total (strong); default 1e8 (enwik), 2^28 (markov)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (host memory in, host memory out) measurement")
    ap.add_argument("--no-overlap", action="store_true", help="skip the extra measurement with two steps in flight (encode of step i + 1 beside the decode of step i)")
    ap.add_argument("--codec", choices=["rop", "rox", "rolz"], default="rop", help="comprop (default, the bench workload), comprox or comprolz block codec")
    ap.add_argument("--stage", choices=["full", "codec"], default="full", help="full: dictionary stage + codec (the reference's per-block path); codec: lzencode / lzdecode only")
    ap.add_argument("--workload", choices=["enwik", "enwik-hard", "markov"], default="enwik",
                    help="enwik: configs[1]; enwik-hard: the same shape with a flatter vocabulary, numbers, URLs and markup (corpus.enwik_hard: "
                         "the dictionary stage leaves ~56 %% instead of 36 %%); markov: config 5's order-2 Markov stream (a slice of the 16 GiB)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--batch-blocks", type=int, default=16384, help="a rank works its blocks off in batches of at most this many (0 = one batch)")
    ap.add_argument("--deadline", type=float, default=float(os.environ.get("CRBENCH_DEADLINE_S", "900")),
                    help="seconds after which a run that has not finished is given up: a rank exits non-zero by itself (watchdog), and the "
                         "parent of --gpus N kills its child job and prints an error line (0 = none)")
    return ap.parse_args(argv)


def error_line(args, what: str) -> str:
    """the one JSON line of a run that failed: same keys a reader of the scaling table looks at, value null"""
    return json.dumps({"metric": "encode+decode MB/s", "value": None, "unit": "MB/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                       "higher_is_better": True, "scaling": args.scaling, "error": what})


def launch_ranks(args) -> int:
    """--gpus N outside torch.distributed.run: start the N ranks as a child job. This process has not initialised the GPU."""
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # The child job runs in its own process group; this process (which never touched a GPU) only waits. A job that hangs — a
    # rank missing from the rendezvous, a collective that never completes on the first 8-GPU lease — is killed as a group
    # when the deadline passes and reported as an error line; nothing is re-executed.
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        rc = child.wait(timeout=args.deadline + 30.0 if args.deadline > 0 else None)      # (the ranks' own watchdogs fire first)
        if rc != 0:
            # rank 0 prints the error line of a run it gives up — unless another rank went first and torch.distributed.run had
            # already ended rank 0: the line a reader of the table looks for must not depend on which rank's timer fired first
            print(error_line(args, f"the {args.gpus}-rank job ended with exit code {rc}: a rank failed, or gave up at its deadline of {args.deadline:.0f} s"), flush=True)
        return rc
    except subprocess.TimeoutExpired:
        import signal
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 10.0)):
            try:
                os.killpg(child.pid, sig)
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        print(error_line(args, f"the {args.gpus}-rank job did not finish within {args.deadline:.0f} s and was killed"), flush=True)
        return 124


def arm_watchdog(args, rank: int):
    """A rank gives up by itself when the deadline passes (a hung collective never returns): message, exit code 124 —
    torch.distributed.run then ends the other ranks. A plain exit, nothing is exec'ed."""
    if args.deadline <= 0:
        return None
    import threading

    def fire():
        sys.stderr.write(f"bench.py rank {rank}: not finished after {args.deadline:.0f} s (deadline) - giving up\n")
        sys.stderr.flush()
        if rank == 0:
            print(error_line(args, f"rank 0 gave up after {args.deadline:.0f} s (deadline)"), flush=True)
        os._exit(124)
    t = threading.Timer(args.deadline, fire)
    t.daemon = True
    t.start()
    return t


# ---------------------------------------------------------------------------------------------- data

def shm_cached(name, make, make_to_file=None):
    """One generation per box: the first local rank writes /dev/shm/<name>, the others wait for it and map it.
    make_to_file(path) writes the stream itself (the parallel generator for the large corpora)."""
    import numpy as np
    path = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", name)
    if not os.path.exists(path):
        lock = path + ".lock"
        try:
            fd = os.open(lock, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
        except FileExistsError:
            fd = None
        if fd is not None:
            try:
                tmp = path + f".{os.getpid()}.tmp"
                if make_to_file is not None:
                    make_to_file(tmp)
                else:
                    make().tofile(tmp)
                os.replace(tmp, path)
            finally:
                os.close(fd)
                os.unlink(lock)
        else:
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > 1800 or not os.path.exists(lock) and not os.path.exists(path):
                    if os.path.exists(path):
                        break
                    return make()            # the writer vanished: generate privately
                time.sleep(0.5)
    return np.fromfile(path, dtype=np.uint8)


def load_corpus(args, seed: int, nbytes: int):
    """The uncompressed stream of one 'file' as a host uint8 array."""
    import numpy as np
    from comprox_amd import corpus
    path = os.environ.get("ENWIK8")
    if args.workload == "enwik" and path and os.path.exists(path) and seed == 8:
        return np.fromfile(path, dtype=np.uint8)[:nbytes], "enwik8"
    if args.workload == "enwik":
        big = (lambda p: corpus.enwik_like_to_file(p, nbytes, seed)) if nbytes >= 200_000_000 else None     # a pool of processes: ~10x
        return shm_cached(f"crbench_enwik_{nbytes}_{seed}.bin", lambda: corpus.enwik_like(nbytes, seed=seed), big), f"synthetic (enwik-shaped generator, seed {seed})"
    if args.workload == "enwik-hard":
        return (shm_cached(f"crbench_hard_{nbytes}_{seed}.bin", lambda: corpus.enwik_hard(nbytes, seed=seed), lambda p: corpus.enwik_hard_to_file(p, nbytes, seed)),
                f"synthetic (harder enwik-shaped generator corpus.enwik_hard, seed {seed})")
    return None, "synthetic (order-2 Markov stream of BASELINE config 5, generated on the device)"


# ---------------------------------------------------------------------------------------------- CPU side-by-side

def _cpu_info():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def _port_job(job):
    """One worker of the all-cores figure: the oracle (CPU restatement) on its own blocks, full path there and back."""
    blocks, dic, codec, full = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import crlib
    o = crlib.Oracle()
    enc_f = {"rop": o.rop_encode, "rox": o.rox_encode, "rolz": o.rolz_encode}[codec]
    dec_f = {"rop": o.rop_decode, "rox": o.rox_decode, "rolz": o.rolz_decode}[codec]
    d = None
    if full:
        d = crlib.DictOracle(o)
        d.load(dic, True)
    t0 = time.perf_counter()
    st = [d.encode(b) for b in blocks] if full else blocks
    enc = [enc_f(s) for s in st]
    t1 = time.perf_counter()
    back = [dec_f(e, len(s)) for e, s in zip(enc, st)]
    if full:
        back = [d.decode(s, len(b)) for s, b in zip(back, blocks)]
    t2 = time.perf_counter()
    assert back == blocks
    return t1 - t0, t2 - t1, enc


def _port_worker(job, barrier, q):
    """One process of the all-cores figure: loads what it needs, waits for the others, codes its blocks there and back."""
    blocks, dic, codec, full = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import crlib
    o = crlib.Oracle()
    enc_f = {"rop": o.rop_encode, "rox": o.rox_encode, "rolz": o.rolz_encode}[codec]
    dec_f = {"rop": o.rop_decode, "rox": o.rox_decode, "rolz": o.rolz_decode}[codec]
    d = None
    if full:
        d = crlib.DictOracle(o)
        d.load(dic, True)
    try:
        barrier.wait(120)                               # (a worker that died on the way leaves the others here: give up together)
    except Exception:
        q.put((0.0, 0.0, -1))
        return
    t0 = time.time()
    st = [d.encode(b) for b in blocks] if full else blocks
    enc = [enc_f(x) for x in st]
    back = [dec_f(e, len(x)) for e, x in zip(enc, st)]
    if full:
        back = [d.decode(x, len(b)) for x, b in zip(back, blocks)]
    t1 = time.time()
    q.put((t0, t1, sum(map(len, blocks)) if back == blocks else -1))


def _ref_job(job):
    """The unmodified reference (oracle/_ref) on a few blocks in a process of its own: it leaks its tables at every reset."""
    blocks, dic, codec, full = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import crlib
    fd = os.open(os.devnull, os.O_WRONLY)
    os.dup2(fd, 2)
    r = crlib.Reference(codec)
    if full:
        r.dictionary_load(dic, True)
    t0 = time.perf_counter()
    st = [r.dictionary_encode(b) for b in blocks] if full else blocks
    enc = [r.encode(s) for s in st]
    t1 = time.perf_counter()
    back = [r.decode(e) for e in enc]
    if full:
        back = [r.dictionary_decode(s) for s in back]
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, enc, back == blocks


def _stock_job(job):
    """BASELINE config 1: the reference's own cr_main(), default 16 MiB dependent blocks, pinned to one core."""
    path_in, codec, workdir = job
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import crlib
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
    except (AttributeError, OSError):
        pass
    L = ctypes.CDLL(crlib.REF_LIBS[codec])

    def run(args):
        pid = os.fork()                       # cr_main closes stderr under -q and keeps file-scope state
        if pid == 0:
            argv = (ctypes.c_char_p * (len(args) + 1))(*[a.encode() for a in args], None)
            os._exit(L.cr_main(len(args), argv) & 255)
        return os.waitpid(pid, 0)[1]

    enc, back = os.path.join(workdir, "stock.enc"), os.path.join(workdir, "stock.back")
    t0 = time.perf_counter()
    rc1 = run(["comp" + codec, "-q", "e", path_in, enc])
    t1 = time.perf_counter()
    rc2 = run(["comp" + codec, "-q", "d", enc, back])
    t2 = time.perf_counter()
    ok = rc1 == 0 and rc2 == 0 and open(back, "rb").read() == open(path_in, "rb").read()
    size = os.path.getsize(enc) if os.path.exists(enc) else 0
    return t1 - t0, t2 - t1, size, ok


def cpu_baseline(data, dic, codec, full, budget_blocks=48):
    """The reference's CPU path beside the GPU's, on a bounded sample of the same block list, on this host's cores."""
    import multiprocessing as mp
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import crlib
    model, ncpu, usable = _cpu_info()
    blocks = [data[i:i + BLOCK].tobytes() for i in range(0, min(data.size, 16 * budget_blocks * BLOCK), BLOCK)]
    one = blocks[:budget_blocks]
    ctx = mp.get_context("fork")
    with ctx.Pool(1) as pool:
        te, td, enc = pool.apply(_port_job, ((one, dic, codec, full),))
    nbytes = sum(map(len, one))
    res = {"value": round(nbytes / 1e6 / (te + td), 3), "unit": "MB/s", "cores": 1, "kind": "port",
           "sample": f"first {len(one)} of the 64 KiB datablocks ({nbytes} B), {'dictionary stage + ' if full else ''}codec, encode+decode round trip, 1 thread of the CPU restatement (oracle/)",
           "encode_MBps": round(nbytes / 1e6 / te, 3), "decode_MBps": round(nbytes / 1e6 / td, 3),
           "cpu_model": model, "host_cores": ncpu, "usable_cores": usable}
    # one worker per usable core over independent blocks: every worker codes its own run of 8 blocks (the block
    # list is walked cyclically when the cores outnumber it), the rate is what all of them moved between the first worker's
    # start and the last one's end, clocks read inside the workers (process start is not the codec's time)
    per = 16
    all_blocks = [data[i:i + BLOCK].tobytes() for i in range(0, data.size, BLOCK)] if usable * per > len(blocks) else blocks
    w = max(1, usable)
    if w > 1:
        barrier, q = ctx.Barrier(w), ctx.SimpleQueue()
        procs = [ctx.Process(target=_port_worker, args=(([all_blocks[(k * per + j) % len(all_blocks)] for j in range(per)], dic, codec, full), barrier, q))
                 for k in range(w)]
        for pr in procs:
            pr.start()
        outs, t_give_up = [], time.time() + 300
        while len(outs) < len(procs) and time.time() < t_give_up:
            if not q.empty():
                outs.append(q.get())
            elif not any(pr.is_alive() for pr in procs) and q.empty():
                break
            else:
                time.sleep(0.01)
        for pr in procs:
            pr.join(5)
            if pr.is_alive():
                pr.kill()
        if len(outs) == len(procs) and all(o[2] >= 0 for o in outs):
          t_first, t_last = min(o[0] for o in outs), max(o[1] for o in outs)
          nb_all = sum(o[2] for o in outs)
          res["all_cores"] = {"value": round(nb_all / 1e6 / (t_last - t_first), 3), "unit": "MB/s", "cores": w, "bytes": nb_all, "blocks_per_worker": per,
                            "note": "one process per usable core, each on its own run of blocks, all released together by a barrier once their "
                                    "dictionaries are loaded; total bytes / (last worker's end - the release)"}
        else:
          res["all_cores"] = {"value": None, "cores": w, "note": f"{len(outs)} of {len(procs)} workers reported: figure left out"}
    if crlib.Reference.available(codec):
        k = 6
        with ctx.Pool(1, maxtasksperchild=1) as pool:
            te, td, renc, ok = pool.apply(_ref_job, ((one[:k], dic, codec, full),))
        rb = sum(map(len, one[:k]))
        res["reference"] = {"kind": "reference", "encode_MBps": round(rb / 1e6 / te, 3), "decode_MBps": round(rb / 1e6 / td, 3),
                            "value": round(rb / 1e6 / (te + td), 3), "bytes": rb, "blocks": k, "roundtrip_ok": ok,
                            "bytes_equal_port": renc == enc[:k],
                            "note": "unmodified reference (oracle/_ref): reset_models(); lzencode / lzdecode per 64 KiB block — it re-initialises "
                                    "and leaks its 68 MB LZP tables at every reset, a mode its CLI cannot even select"}
        # config 1: the stock tool, default 16 MiB dependent blocks, one pinned thread
        sample = data[:32 * 1048576]
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
            p = os.path.join(d, "stock.in")
            sample.tofile(p)
            with ctx.Pool(1, maxtasksperchild=1) as pool:
                te, td, size, ok = pool.apply(_stock_job, ((p, codec, d),))
        res["stock"] = {"kind": "reference", "config": "BASELINE configs[0]: stock CLI (cr_main), default 16 MiB dependent blocks, 1 pinned thread",
                        "bytes": int(sample.size), "encode_MBps": round(sample.size / 1e6 / te, 3), "decode_MBps": round(sample.size / 1e6 / td, 3),
                        "value": round(sample.size / 1e6 / (te + td), 3), "ratio": round(size / max(1, sample.size), 5), "roundtrip_ok": ok}
    return res


# ---------------------------------------------------------------------------------------------- the bench

def golden_record(seed, codec, stage, file_n, family="enwik_like"):
    """The unmodified reference's record for the corpus enwik_like(file_n, seed) (tests/golden/golden_scale.json), or None."""
    tag = {100_000_000: "1e8", 1_000_000_000: "1e9"}.get(file_n)
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_scale.json")))["o2"]
        return g[f"{family}_{tag}_seed{seed}"][f"{codec}/{stage}"] if tag else g[f"{family}_1e8_seed{seed}"][f"{codec}/{stage}"]
    except (OSError, KeyError):
        return None


def golden_cut(seed, codec, stage, n, family="enwik_like"):
    g = golden_record(seed, codec, stage, n, family)
    if g is None:
        return None
    for cut in g["cuts"].values():
        if cut["blocks"] == g["blocks"] and n == g["n"]:
            return cut                       # the whole 1e8-byte corpus
        # a prefix of the corpus is the same stream, but its dictionary would be picked from the prefix alone: the 1 MiB
        # and 16 MiB cuts only pin the codec stage
        if stage == "codec" and cut["blocks"] * BLOCK == n:
            return cut
    return None


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    from comprox_amd import CrGpu, CODEC_ROP, CODEC_ROX, CODEC_ROLZ, bound, shard, api

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    arm_watchdog(args, rank)
    if os.environ.get("CRBENCH_TEST_STALL_RANK") == str(rank) and world > 1:      # tests: the rank that never joins
        while True:
            time.sleep(1.0)
    # rehearsal on a one-GPU box (tests/test_gpu_bench.py): every rank on GPU 0 and gloo instead of RCCL, which refuses
    # two ranks on one device; the collectives below then travel through host tensors
    backend = os.environ.get("CRBENCH_BACKEND", "nccl")
    if os.environ.get("CRBENCH_ONE_GPU"):
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        import datetime
        patience = datetime.timedelta(seconds=args.deadline if args.deadline > 0 else 1800)
        try:
            if backend == "nccl":
                torch.cuda.set_device(local)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=patience)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=patience)    # (gloo: no device needed to rendezvous)
        except Exception as e:                       # a rank missing from the rendezvous: same report as the watchdog's
            sys.stderr.write(f"bench.py rank {rank}: rendezvous failed: {e}\n")
            if rank == 0:
                print(error_line(args, f"rank 0: rendezvous of {world} ranks failed within the deadline of {args.deadline:.0f} s: {type(e).__name__}"), flush=True)
            os._exit(124)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    on_host = backend != "nccl"

    def all_gather_into(out_t, in_t):
        if on_host:
            o = torch.empty(out_t.shape, dtype=out_t.dtype)
            dist.all_gather_into_tensor(o, in_t.cpu())
            out_t.copy_(o)
        else:
            dist.all_gather_into_tensor(out_t, in_t)

    def send_to(t, dst):
        dist.send(t.cpu() if on_host else t, dst=dst)

    def recv_from(t, src):
        if on_host:
            o = torch.empty(t.shape, dtype=t.dtype)
            dist.recv(o, src=src)
            t.copy_(o)
        else:
            dist.recv(t, src=src)
    full = args.stage == "full"
    strong = args.scaling == "strong"
    total_bytes = args.bytes or (SHARD_BYTES if args.workload != "markov" else 1 << 28)

    # ---- this rank's blocks
    # seed 8 / 1e8 bytes stands in for enwik8, seed 9 / 1e9 bytes for enwik9; weak scaling: rank r its own shard
    base_seed = 9 if total_bytes == 1_000_000_000 and args.workload == "enwik" else 8
    family = "enwik_hard" if args.workload == "enwik-hard" else "enwik_like"
    texty = args.workload in ("enwik", "enwik-hard")
    seed = base_seed if strong else base_seed + rank
    file_host, data_note = load_corpus(args, seed, total_bytes)          # the 'file' this rank's blocks come from
    if args.workload == "markov":
        from comprox_amd import corpus
        nb_file = total_bytes // BLOCK
        lo, hi = shard.partition(nb_file, world, rank) if strong else (0, nb_file)
        first_index = lo if strong else rank * nb_file
        # generated on the device, a slab of blocks at a time (the generator's temporaries are 64 KiB x 8 B per block)
        d_in = torch.empty((hi - lo) * BLOCK, dtype=torch.uint8, device=dev)
        for b0 in range(0, hi - lo, 16384):
            k = min(16384, hi - lo - b0)
            d_in[b0 * BLOCK:(b0 + k) * BLOCK] = corpus.markov2_blocks(k, first_index + b0, BLOCK, device=dev).reshape(-1)
        n = int(d_in.numel())
        file_n = nb_file * BLOCK
        host = None
    else:
        file_n = int(file_host.size)
        nb_file = (file_n + BLOCK - 1) // BLOCK
        lo, hi = shard.partition(nb_file, world, rank) if strong else (0, nb_file)
        host = file_host[lo * BLOCK:min(file_n, hi * BLOCK)]
        n = int(host.size)
        d_in = torch.from_numpy(np.ascontiguousarray(host)).to(dev)
    nb = hi - lo
    in_off_h = np.arange(nb, dtype=np.int64) * BLOCK
    in_size_h = np.minimum(BLOCK, n - in_off_h).astype(np.int32) if nb else np.zeros(0, dtype=np.int32)
    CODEC = {"rop": CODEC_ROP, "rox": CODEC_ROX, "rolz": CODEC_ROLZ}[args.codec]

    g = CrGpu(local)
    # ONE stream for the library's kernels and torch's own operations (the size exchange, the per-batch checks): torch's
    # default stream is the NULL stream, which crgpu_set_stream takes as "the context's own stream" — unordered against it
    stream = torch.cuda.Stream(dev)
    side_stream = torch.cuda.Stream(dev)                # for the measurements with two contexts in flight: created next to `stream`, because the runtime
                                                        # deals its few hardware queues out in creation order and two streams on ONE queue run one after the other
    torch.cuda.set_stream(stream)
    g.set_stream(stream.cuda_stream)
    assert stream.cuda_stream != 0

    # ---- per-file dictionary (host pass, once per file, outside the timed region: src/main.c:156-171)
    gdict, dic_text, t_dicpick = None, b"", 0.0
    MARKOV_DIC_BYTES = 1 << 28
    if full:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        t0 = time.perf_counter()
        if file_host is not None:
            file_bytes = file_host
        elif first_index == 0 and n >= min(file_n, MARKOV_DIC_BYTES):
            file_bytes = d_in[:min(file_n, MARKOV_DIC_BYTES)].cpu().numpy()
        else:   # the Markov stream is no file: its dictionary is picked from its first 2^28 bytes (blocks 0 .. 4 095) on every rank
            file_bytes = corpus.markov2_blocks(min(nb_file, MARKOV_DIC_BYTES // BLOCK), 0, BLOCK, device=dev).reshape(-1).cpu().numpy()
        dic_text = host_dicpick(g.lib, file_bytes)
        t_dicpick = time.perf_counter() - t0
        gdict = g.dict_create(dic_text)

    # ---- device buffers: every stage has its own strided slots, the pack is contiguous. A rank's blocks are worked off
    # in batches of at most --batch-blocks (the encode pipeline keeps ~4.9 MB of event scratch per 64 KiB block): one
    # batch for the enwik configs, 32 for the 16 GiB of config 5. The input stays resident in HBM as a whole.
    NBB = max(1, min(nb, args.batch_blocks)) if args.batch_blocks > 0 else max(1, nb)
    nbatch = (nb + NBB - 1) // NBB
    s1 = (BLOCK + 1 + 63) // 64 * 64
    s2 = (bound(CODEC, BLOCK + (1 if full else 0)) + 63) // 64 * 64
    i64, i32, u8 = torch.int64, torch.int32, torch.uint8
    d_in_off = torch.from_numpy(in_off_h).to(dev)
    d_in_size = torch.from_numpy(in_size_h).to(dev)
    d_rel_off = torch.arange(NBB, dtype=i64, device=dev) * BLOCK                       # a batch's blocks in d_dec
    d_st1 = torch.zeros(NBB * s1, dtype=u8, device=dev) if full else None
    d_st1_off = torch.arange(NBB, dtype=i64, device=dev) * s1
    d_len1 = torch.zeros(max(1, nb), dtype=i32, device=dev)
    d_enc = torch.zeros(NBB * s2, dtype=u8, device=dev)
    d_enc_off = torch.arange(NBB, dtype=i64, device=dev) * s2
    d_enc_size = torch.zeros(max(1, nb), dtype=i32, device=dev)
    d_pack = torch.zeros(NBB * s2, dtype=u8, device=dev)
    d_pack_off = torch.zeros(NBB, dtype=i64, device=dev)
    d_total = torch.zeros(2, dtype=i64, device=dev)
    d_st1b = torch.zeros(NBB * s1, dtype=u8, device=dev) if full else None
    d_len1b = torch.zeros(NBB, dtype=i32, device=dev)
    d_dec = torch.zeros(min(n, NBB * BLOCK) + 64, dtype=u8, device=dev)
    d_dec_size = torch.zeros(NBB, dtype=i32, device=dev)
    d_acc = torch.zeros(3, dtype=i64, device=dev)       # several batches: [blocks/bytes that came back wrong, packed bytes, failed blocks] of the step
    per = (nb_file + world - 1) // world if strong else nb
    d_mine = torch.zeros(max(1, per), dtype=i32, device=dev)
    d_all_sizes = torch.zeros(max(1, per * world), dtype=i32, device=dev) if world > 1 else None

    # kernel times: the library keeps the HIP-event boundaries of every launch of the timed steps (on the kernels' own
    # stream) and folds them up AFTER the timed region — no event wait between the calls of a step

    class Bufs:                                         # what the encode half of a step hands to its decode half
        def __init__(self, fresh):
            z = (lambda t: None if t is None else torch.zeros_like(t)) if fresh else (lambda t: t)
            self.st1, self.len1, self.enc, self.enc_size = z(d_st1), z(d_len1), z(d_enc), z(d_enc_size)
            self.pack, self.pack_off, self.total = z(d_pack), z(d_pack_off), z(d_total)
    bufs0 = Bufs(False)

    def run_encode(gx, dx, U, b0, k):
        """dictionary_encode -> lzencode -> k_pack for blocks [b0, b0 + k) of this rank, on context gx, into the buffer set U"""
        src, src_off, src_size = d_in, d_in_off[b0:], d_in_size[b0:]
        len1, enc_size = U.len1[b0:], U.enc_size[b0:]
        if full:                                        # dictionary_encode, src/main.c:189
            gx.lib.crgpu_dict_encode_blocks_dev(gx.h, dx.h, src.data_ptr(), src_off.data_ptr(), src_size.data_ptr(), k, BLOCK,
                                                U.st1.data_ptr(), d_st1_off.data_ptr(), len1.data_ptr(), 0)
            src, src_off, src_size = U.st1, d_st1_off, len1
        gx.encode_blocks_dev(CODEC, src.data_ptr(), src_off.data_ptr(), src_size.data_ptr(), k, BLOCK + (1 if full else 0),
                             U.enc.data_ptr(), d_enc_off.data_ptr(), enc_size.data_ptr())          # lzencode, src/main.c:194
        gx.pack_blocks_dev(U.enc.data_ptr(), d_enc_off.data_ptr(), enc_size.data_ptr(), k, U.pack.data_ptr(),
                           U.pack_off.data_ptr(), U.total.data_ptr())                                  # the write loop, src/main.c:198-205

    class Outs:                                         # where a decode puts its stages' results (a second set for a second half in flight)
        def __init__(self, fresh):
            z = (lambda t: None if t is None else torch.zeros_like(t)) if fresh else (lambda t: t)
            self.st1b, self.len1b, self.dec, self.dec_size = z(d_st1b), z(d_len1b), z(d_dec), z(d_dec_size)
    outs0 = Outs(False)

    def run_decode(gx, dx, U, b0, k, O=None):
        """lzdecode -> dictionary_decode of what run_encode left in U"""
        O = O or outs0
        len1, enc_size = U.len1[b0:], U.enc_size[b0:]
        cap = len1 if full else d_in_size[b0:]
        dst, dst_off = (O.st1b, d_st1_off) if full else (O.dec, d_rel_off)
        gx.decode_blocks_dev(CODEC, U.pack.data_ptr(), U.pack_off.data_ptr(), enc_size.data_ptr(), k, BLOCK + (1 if full else 0),
                             dst.data_ptr(), dst_off.data_ptr(), cap.data_ptr(), (O.len1b if full else O.dec_size).data_ptr())   # lzdecode, src/main.c:277
        if full:                                        # dictionary_decode, src/main.c:281
            gx.lib.crgpu_dict_decode_blocks_dev(gx.h, dx.h, O.st1b.data_ptr(), d_st1_off.data_ptr(), O.len1b.data_ptr(), k, BLOCK,
                                                O.dec.data_ptr(), d_rel_off.data_ptr(), d_in_size[b0:].data_ptr(), O.dec_size.data_ptr(), 0)

    def run_batch(b0, k):
        """dictionary_encode -> lzencode -> k_pack -> lzdecode -> dictionary_decode for blocks [b0, b0 + k) of this rank"""
        run_encode(g, gdict, bufs0, b0, k)
        if world > 1 and nbatch == 1:                   # the one exchange: every rank learns every block's size
            d_mine[:nb] = d_enc_size[:nb]
            all_gather_into(d_all_sizes, d_mine)
        run_decode(g, gdict, bufs0, b0, k)

    def step(record: bool):
        if nbatch == 1:
            run_batch(0, nb)
            return
        # several batches: every batch's round trip is compared on the device inside the step (two reads of the batch at
        # HBM speed, ~0.1 % of its time), because the next batch reuses the buffers
        d_acc.zero_()
        for b0 in range(0, nb, NBB):
            k = min(NBB, nb - b0)
            run_batch(b0, k)
            lo_b, hi_b = b0 * BLOCK, min(n, (b0 + k) * BLOCK)
            d_acc[0] += (d_dec[:hi_b - lo_b] != d_in[lo_b:hi_b]).sum() + (d_dec_size[:k] != d_in_size[b0:b0 + k]).sum()
            d_acc[1] += d_total[0]
            d_acc[2] += d_total[1]
        if world > 1:
            d_mine[:nb] = d_enc_size[:nb]
            all_gather_into(d_all_sizes, d_mine)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    fence()
    g.stage_log(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    stage_log = g.stage_log_read()                      # kernel -> (summed ms, launches) over the timed steps
    g.stage_log(False)
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_host else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- what was timed is checked: the round trip, the sizes, the bytes
    comp = int(d_enc_size[:nb].to(i64).sum().item())
    st1_bytes = int(d_len1[:nb].to(i64).sum().item()) if full else n
    if nbatch == 1:
        ok = bool(torch.equal(d_dec[:n], d_in)) and bool((d_dec_size[:nb] == d_in_size).all().item()) and int(d_total[1].item()) == 0
        ok = ok and comp == int(d_total[0].item())
        packed = d_pack[:comp].cpu().numpy()
    else:
        acc = d_acc.tolist()                            # of the last timed step
        ok = acc[0] == 0 and acc[2] == 0 and acc[1] == comp
        packed = None
        if not ok:
            print(f"rank {rank}: batches of the last step: {acc[0]} bytes / sizes came back wrong, {acc[2]} blocks failed, {acc[1]} bytes packed of {comp}", file=sys.stderr, flush=True)
    golden_equal = None
    gather_checked = None
    if strong and world > 1 and packed is not None:
        # the gather, checked: rank 0 receives the runs in rank order, derives their offsets from the size table the
        # timed step exchanged and compares the assembled stream with the reference's
        sizes_all = d_all_sizes.cpu().numpy().astype(np.int64)
        rank_bytes = [int(sizes_all[r * per:(r + 1) * per].sum()) for r in range(world)]
        if rank == 0:
            h = hashlib.sha256(packed.tobytes())
            for r in range(1, world):
                buf = torch.empty(max(1, rank_bytes[r]), dtype=u8, device=dev)
                recv_from(buf, r)
                h.update(buf[:rank_bytes[r]].cpu().numpy().tobytes())
            cut = golden_cut(base_seed, args.codec, args.stage, file_n, family) if texty and data_note.startswith("synthetic") else None
            gather_checked = True
            golden_equal = None if cut is None else (cut["size"] == sum(rank_bytes) and cut["sha256"] == h.hexdigest())
        else:
            send_to(d_pack[:max(1, comp)].contiguous(), 0)
    elif args.workload == "markov" and first_index == 0 and nb >= 256 and file_n >= MARKOV_DIC_BYTES:
        # config 5: blocks 0 .. 255 of the stream once more (untimed) against the unmodified reference's bytes for them
        try:
            mk = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_scale.json")))["o2"]["markov2_first256"][f"{args.codec}/{args.stage}"]["cuts"]["full"]
        except (OSError, KeyError):
            mk = None
        if mk is not None:
            run_batch(0, 256)
            torch.cuda.synchronize(dev)
            c256 = int(d_enc_size[:256].to(i64).sum().item())
            golden_equal = mk["size"] == c256 and mk["sha256"] == hashlib.sha256(d_pack[:c256].cpu().numpy().tobytes()).hexdigest()
    elif texty and data_note.startswith("synthetic") and packed is not None:
        cut = golden_cut(seed, args.codec, args.stage, n, family)
        if cut is not None:
            golden_equal = cut["size"] == comp and cut["sha256"] == hashlib.sha256(packed.tobytes()).hexdigest()
    # strong scaling: every rank checks ITS run against the reference's bytes for its contiguous range (no gather needed)
    rank_equal = None
    if strong and world > 1 and texty and data_note.startswith("synthetic") and packed is not None:
        rec = golden_record(base_seed, args.codec, args.stage, file_n, family)
        mine = ((rec or {}).get("ranks") or {}).get(str(world))
        if mine:
            mine = mine[rank]
            rank_equal = (mine["first"] == lo and mine["count"] == nb and mine["size"] == comp and
                          mine["sha256"] == hashlib.sha256(packed.tobytes()).hexdigest())
    flags = torch.tensor([n, comp, int(ok), st1_bytes, -1 if golden_equal is None else int(golden_equal),
                          -1 if rank_equal is None else int(rank_equal)], dtype=i64, device="cpu" if on_host else dev)
    if world > 1:
        gathered = [torch.zeros_like(flags) for _ in range(world)]
        dist.all_gather(gathered, flags)
    else:
        gathered = [flags]
    rows = [x.tolist() for x in gathered]
    total_n, total_comp, total_st1 = sum(r[0] for r in rows), sum(r[1] for r in rows), sum(r[3] for r in rows)
    all_ok = all(r[2] == 1 for r in rows)
    gold_rows = [r[4] for r in rows if r[4] >= 0]
    bytes_equal_golden = (all(v == 1 for v in gold_rows) if gold_rows else None)
    if strong and world > 1:
        bytes_equal_golden = None if rows[0][4] < 0 else bool(rows[0][4])
    rank_rows = [r[5] for r in rows if r[5] >= 0]
    ranks_equal_golden = (all(v == 1 for v in rank_rows) if rank_rows else None)
    if ranks_equal_golden is False:
        bytes_equal_golden = False

    # ---- an EXTRA measurement (never `value`): two steps in flight. The decoder is one dependent chain per block - 1 526
    # waves on 1 024 SIMDs, each waiting for memory half of the time - so a second context on a second stream can run the
    # encode of step i + 1 beside the decode of step i. Same work per step (every decode reads the packed bytes its own
    # encode wrote, two buffer sets alternate), all of it inside the bracketed region; what changes is the schedule.
    def two_in_flight():
        g2 = CrGpu(local)
        s_enc = side_stream
        g2.set_stream(s_enc.cuda_stream)
        gdict2 = g2.dict_create(dic_text) if full else None
        sets = [bufs0, Bufs(True)]
        dec_done, dec_go = [None, None], [None]

        def overlapped(steps):
            for i in range(steps):
                U = sets[i & 1]
                if dec_done[i & 1] is not None:
                    s_enc.wait_event(dec_done[i & 1])                   # step i - 2 has read this buffer set
                if dec_go[0] is not None:
                    s_enc.wait_event(dec_go[0])                         # the decoder's waves go onto the empty chip first
                run_encode(g2, gdict2, U, 0, nb)                        # (context g2 -> stream s_enc)
                ev = torch.cuda.Event()
                ev.record(s_enc)
                stream.wait_event(ev)
                dec_go[0] = torch.cuda.Event()
                dec_go[0].record(stream)                                # = step i's encode is done: its decode starts now
                run_decode(g, gdict, U, 0, nb)                          # (context g -> `stream`)
                dec_done[i & 1] = torch.cuda.Event()
                dec_done[i & 1].record(stream)
        # (the decoder with 272 bytes of LDS per block, k_rop_decode_v5s: its six workgroups per CU then leave room for a sorting
        # kernel's 152 KB; with the first dense nodes in LDS the decoder alone is 3 % faster and this schedule 12 % slower)
        g.set_option(api.OPT_DECODER_LDS_NODES, 0)
        try:
            overlapped(max(2, args.warmup))
            fence()
            t0 = time.perf_counter()
            overlapped(args.steps)                                      # (the first of them finds the chip empty, the last decode runs alone)
            fence()
            el2 = time.perf_counter() - t0
        finally:
            g.set_option(api.OPT_DECODER_LDS_NODES, 1)
        ok2 = bool(torch.equal(d_dec[:n], d_in)) and bool((d_dec_size[:nb] == d_in_size).all().item())
        d_pack_ref = torch.from_numpy(packed).to(dev)                   # what the serial steps packed
        same2 = all(int(U.total[0].item()) == comp and int(U.total[1].item()) == 0 and bool(torch.equal(U.pack[:comp], d_pack_ref)) for U in sets)
        del gdict2, g2
        return {"value": round(n / 1e6 / (el2 / args.steps), 2), "unit": "MB/s", "ms_per_step": round(el2 / args.steps * 1e3, 3), "steps": args.steps,
                "schedule": "two steps in flight: the encode of step i + 1 (second context, second stream) runs beside the decode of step i; "
                            "each decode reads what its own step's encode packed (two buffer sets)",
                "roundtrip_ok": ok2, "packed_bytes_equal_the_serial_steps": same2}

    overlap = None
    if world == 1 and nbatch == 1 and not args.no_overlap and 0 < nb <= 4096:     # (a second set of encoder arenas: not beside the 15 259 blocks of config 3)
        try:
            overlap = two_in_flight()
        except Exception as e:  # noqa: BLE001 — a side measurement must not take the line down
            overlap = {"error": repr(e)}
            torch.cuda.synchronize(dev)

    # which match pre-pass took the blocks of the last timed encode (crgpu_last_prepass_paths: the kernels mark every block they
    # finish): the LDS sorts hold blocks of up to 28 672 bytes, in groups by key up to 65 537; what is left goes to the table sweep
    codec_in = (d_len1[:nb] if full else d_in_size[:nb]).to(i64)
    pp = g.last_prepass_paths()
    paths = {"prepass_in_lds_blocks": pp["lds_28k"] + pp["lds_64k"], "prepass_lds_64k_blocks": pp["lds_64k"], "prepass_table_sweep_blocks": pp["table_sweep"],
             "blocks_of_the_last_batch": pp["lds_28k"] + pp["lds_64k"] + pp["table_sweep"],
             "codec_input_bytes_per_block_mean": round(float(codec_in.double().mean().item()), 1) if nb else 0.0,
             "codec_input_bytes_per_block_max": int(codec_in.max().item()) if nb else 0, "rank": 0,
             "note": "k_rop_lzp_lds / k_rox_links_lds / k_rolz_match_lds take the blocks the codec sees with up to 28 672 bytes, k_rop_lzp_lds64 comprop's "
                     "blocks of up to 65 537 bytes (sorted in groups by key), the table sweeps the rest"}
    rc = 0
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        parts = {k: ms / args.steps for k, (ms, cnt) in stage_log.items()}            # ms per step (all launches of a step)
        per_launch = {k: ms / max(1, cnt) for k, (ms, cnt) in stage_log.items()}      # average duration of one launch
        enc_names = [k for k in parts if k not in ("k_dict_encode", "k_dict_decode", "k_pack_scan", "k_pack_copy") and "decode" not in k]
        dec_names = [k for k in parts if "decode" in k and k != "k_dict_decode"]
        e_ms = sum(parts[k] for k in enc_names) + parts.get("k_dict_encode", 0.0)
        d_ms = sum(parts[k] for k in dec_names) + parts.get("k_dict_decode", 0.0)
        dom = max(parts, key=parts.get)
        dom_ms = per_launch[dom]
        # algorithmic bytes of one launch of a kernel (rank 0's blocks): what its stage must read + write
        stage_bytes = {"k_dict_encode": n + st1_bytes, "k_dict_decode": st1_bytes + n, "k_pack_scan": 12 * nb, "k_pack_copy": 2 * comp}
        algo = stage_bytes.get(dom, st1_bytes + comp) // nbatch       # one launch codes one batch
        ach = algo / (dom_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the value comes
        # from the committed rocprofv3 passes of this same command (tools/collect_traffic.py); the file is named
        traffic, traffic_src, traffic_when = None, None, None
        import glob
        want_cmd = f"--stage {args.stage}"
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
            try:
                tj = json.load(open(tpath))
            except (OSError, ValueError):
                continue
            # (a file names the workload and the bytes per GPU its passes ran with; older files: the default 1e8-byte enwik shard)
            if dom in tj.get("kernels", {}) and tj.get("codec", "rop") == args.codec and want_cmd in tj.get("command", "") \
                    and world == 1 and n == int(tj.get("bytes") or SHARD_BYTES) and args.workload == tj.get("workload", "enwik") \
                    and not os.environ.get("ENWIK8"):
                traffic, traffic_src = tj["kernels"][dom]["hbm_raw"], os.path.relpath(tpath, ROOT)
                # a committed constant, NOT a measurement of this run: say which profile and when
                tag = os.path.basename(tpath).split("_")[0]
                when = tj.get("measured_on") or "date not recorded in the file"
                traffic_when = f"{tag} ({when}): two rocprofv3 --pmc passes of this command on another box, not this run"
                break
        codec_note = {"rop": "comprop codec (LZP+PPM+range coder)", "rox": "comprox codec (LZ77+PPM+4 range-coder streams)",
                      "rolz": "comprolz codec (ROLZ+PPM+2 range-coder streams)"}[args.codec]
        wl = ("enwik8-shaped" if args.workload == "enwik" and total_bytes != 1_000_000_000 else "enwik9-shaped (BASELINE config 3's load)" if args.workload == "enwik"
              else "enwik8-shaped, HARDER variant (flatter vocabulary, numbers, URLs, markup)" if args.workload == "enwik-hard"
              else "order-2 Markov, the 16 GiB of BASELINE config 5" if total_bytes == 1 << 34 else "order-2 Markov (config 5 slice)") + \
             (f" {total_bytes} B per GPU" if not strong else f" {total_bytes} B in total, contiguous block ranges per GPU") + \
             ", 64 KiB independent datablocks, " + ("dictionary stage + " if full else "") + codec_note
        line = {
            "metric": "encode+decode MB/s on " + ("enwik8-shaped stream" if args.workload == "enwik" else "harder enwik8-shaped stream (corpus sensitivity)" if args.workload == "enwik-hard"
                                                  else "order-2 Markov stream (BASELINE config 5)") +
                      ", 64 KiB independent datablocks, " + ("dictionary stage + codec" if full else "codec stage only"),
            "value": round(total_n / 1e6 / (elapsed / args.steps), 2),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8/u32",
            "data": data_note if not strong else data_note + ", one corpus shared by all ranks",
            "config": {"workload": wl, "bytes_per_gpu": n, "blocks_per_gpu": nb, "block_bytes": BLOCK, "total_bytes": total_n,
                       "step": ("dictionary_encode -> lzencode -> k_pack -> " + ("size all_gather -> " if world > 1 else "") + "lzdecode -> dictionary_decode") if full
                               else ("lzencode -> k_pack -> " + ("size all_gather -> " if world > 1 else "") + "lzdecode"),
                       "parallelism": f"blocks sharded over {world} rank(s) ({'RCCL' if backend == 'nccl' else backend} world size {dist.get_world_size() if world > 1 else 1}), no data-path collective",
                       "dictionary": {"bytes": len(dic_text), "host_dicpick_s": round(t_dicpick, 3), "note": "per-file host pass (src/main.c:156-171), outside the timed step"} if full else None},
            "encode_MBps": round(n / 1e6 / (e_ms * 1e-3), 2),
            "decode_MBps": round(n / 1e6 / (d_ms * 1e-3), 2),
            "kernel_ms": {k: round(v, 3) for k, v in parts.items()},
            "kernel_ms_per_launch": {k: round(v, 3) for k, v in per_launch.items()} if nbatch > 1 else None,
            "encode_ms": round(e_ms, 3), "decode_ms": round(d_ms, 3),
            "compressed_bytes": total_comp,
            "dictionary_stage_bytes": total_st1 if full else None,
            "paths": paths,
            "ratio": round(total_comp / max(1, total_n), 5),
            "roundtrip_ok": all_ok,
            "bytes_equal_golden": bytes_equal_golden,
            "golden": ("tests/golden/golden_scale.json (SHA-256 of the unmodified reference's per-block outputs, back to back" +
                       ("; blocks 0 .. 255 of the stream)" if args.workload == "markov" else ")")) if bytes_equal_golden is not None else None,
            "batches_per_step": nbatch,
            "gather_checked": gather_checked,
            "ranks_equal_golden": ranks_equal_golden,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src, "traffic_measured_on": traffic_when,
                         "algorithmic_bytes": algo},
        }
        if overlap is not None:
            line["two_steps_in_flight"] = overlap
        if world == 1 and host is not None and not args.no_e2e and nbatch == 1:
            try:
                line["end_to_end"] = end_to_end(local, host, CODEC, dic_text, full)
            except Exception as e:  # noqa: BLE001 — a side measurement
                line["end_to_end"] = {"error": repr(e)}
        if not args.no_cpu and host is not None:
            try:
                line["cpu_baseline"] = cpu_baseline(host, dic_text, args.codec, full)
                # vs_baseline stays null (BASELINE.md holds no published number for this metric); the ratios to the CPU
                # paths timed beside it in this run are reported under their own name
                cb = line["cpu_baseline"]
                line["vs_cpu_baseline"] = {k: round(line["value"] / v, 1) for k, v in (
                    ("port_1_thread", cb.get("value")), ("port_all_cores", (cb.get("all_cores") or {}).get("value")),
                    ("reference_per_block", (cb.get("reference") or {}).get("value")), ("reference_stock_cli", (cb.get("stock") or {}).get("value"))) if v}
            except Exception as e:  # noqa: BLE001 — the baseline is a side measurement; the GPU line stands without it
                line["cpu_baseline"] = {"error": repr(e)}
        if not all_ok or bytes_equal_golden is False:
            line["value"] = None
            line["error"] = "round trip failed" if not all_ok else "compressed bytes differ from the reference's"
            rc = 1
        print(json.dumps(line), flush=True)
    if gdict is not None:
        gdict.close()
    g.close()
    if world > 1:
        code = torch.tensor([rc], dtype=i64, device="cpu" if on_host else dev)
        dist.broadcast(code, src=0)
        rc = int(code.item())
        dist.destroy_process_group()
    sys.exit(rc)


def end_to_end(local, host, codec, dic_text, full, reps=3):
    """The same blocks from HOST memory and back (kernel-only AND end-to-end): crgpu_multi_encode_blocks /
    crgpu_multi_decode_blocks (include/crgpu.h; what comp*-gpu -k runs) on one device — upload, dictionary stage, codec,
    k_pack, download of the packed run; then upload of the packed blocks, decode, download. The input is a pageable numpy buffer,
    the results come back in the contexts' page-locked pools (CRGPU_MULTI_PINNED_OUT, round 4: what comp*-gpu uses) — one context
    encodes, a second one decodes, because a pooled result only lives until its context's next job; best of `reps` after one
    warm-up. Not `value`: reported next to it."""
    import ctypes
    import numpy as np
    from comprox_amd import api
    n = int(host.size)
    nb = (n + BLOCK - 1) // BLOCK
    in_off = (np.arange(nb, dtype=np.uint64) * np.uint64(BLOCK))
    sizes = np.minimum(BLOCK, n - in_off.astype(np.int64)).astype(np.uint32)
    src = np.ascontiguousarray(host)
    m = api.CrMulti([local], host_gather=True, pinned_out=True)
    m2 = api.CrMulti([local], host_gather=True, pinned_out=True)
    flags = api.MULTI_DICT if full else 0
    if full:
        m.set_dictionary(dic_text)
        m2.set_dictionary(dic_text)
    L = m.lib
    L.crgpu_multi_timing.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int]

    def marks(mm):
        """ms from the call's start to: input on the device, stages done, sizes exchanged, output allocated, run copied out"""
        t = (ctypes.c_double * 6)()
        k = L.crgpu_multi_timing(mm.h, 0, t, 6)
        return [round((t[i] - t[0]) * 1e3, 2) for i in range(1, k)]
    best = None
    for rep in range(reps + 1):
        out, total = ctypes.c_void_p(), ctypes.c_uint64()
        out_off = np.zeros(nb, dtype=np.uint64)
        out_size = np.zeros(nb, dtype=np.uint32)
        t0 = time.perf_counter()
        m._check(L.crgpu_multi_encode_blocks(m.h, codec, flags, api._ptr(src), api._ptr(in_off), api._ptr(sizes), nb, None,
                                             ctypes.byref(out), ctypes.byref(total), api._ptr(out_off), api._ptr(out_size)), "crgpu_multi_encode_blocks")
        t1 = time.perf_counter()
        marks_e = marks(m)
        back, btotal = ctypes.c_void_p(), ctypes.c_uint64()
        m2._check(L.crgpu_multi_decode_blocks(m2.h, codec, flags, out, api._ptr(out_off), api._ptr(out_size), nb, None,
                                             ctypes.byref(back), ctypes.byref(btotal), None, None), "crgpu_multi_decode_blocks")
        t2 = time.perf_counter()
        marks_d = marks(m2)
        ok = btotal.value == n and ctypes.string_at(back.value, n) == src.tobytes()
        comp = int(total.value)
        L.crgpu_multi_free(out)
        L.crgpu_multi_free(back)
        if not ok:
            m.close(); m2.close()
            return {"error": "the end-to-end round trip does not reproduce the input"}
        if rep and (best is None or (t2 - t0) < sum(best[:2])):
            best = (t1 - t0, t2 - t1, marks_e, marks_d)
    m.close(); m2.close()
    e, d, marks_e, marks_d = best
    return {"encode_MBps": round(n / 1e6 / e, 1), "decode_MBps": round(n / 1e6 / d, 1), "roundtrip_MBps": round(n / 1e6 / (e + d), 1),
            "encode_ms": round(e * 1e3, 2), "decode_ms": round(d * 1e3, 2), "compressed_bytes": comp, "roundtrip_ok": True,
            "encode_marks_ms": marks_e, "decode_marks_ms": marks_d,
            "marks": "ms from the call's start to: input on the device, stages done, sizes exchanged, output allocated, run copied to the host",
            "path": "host numpy buffer (pageable) -> crgpu_multi_encode_blocks -> page-locked pool -> crgpu_multi_decode_blocks -> page-locked pool, one device, "
                    + ("dictionary stage + codec" if full else "codec stage") + "; wall clock of the two calls, best of %d" % reps}


def host_dicpick(lib, data) -> bytes:
    """dicpick() of libcrgpu.so (host C, csrc/crhost_dict.c) on an in-memory stream, through a temporary file."""
    import ctypes
    import tempfile
    from comprox_amd import api
    libc = ctypes.CDLL(None)
    libc.fopen.restype = ctypes.c_void_p
    libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    libc.fclose.argtypes = [ctypes.c_void_p]
    with tempfile.NamedTemporaryFile(delete=False, dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as t:
        data.tofile(t)
    try:
        fp = libc.fopen(t.name.encode(), b"rb")
        db = api.DataBlock()
        lib.dicpick.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.dicpick.restype = None
        lib.dicpick(fp, ctypes.byref(db))
        libc.fclose(fp)
    finally:
        os.unlink(t.name)
    out = ctypes.string_at(db.m_data, db.m_size)
    lib.data_block_destroy.argtypes = [ctypes.c_void_p]
    lib.data_block_destroy(ctypes.byref(db))
    return out


if __name__ == "__main__":
    main()
