"""Simulation asked for by VERDICT r3 (next #1a): what would an LDS cache of order-2 nodes catch PER DECODE STEP, with round 3's
pair format as the entry (a node of <= 6 / 14 / 30 {symbol, count} pairs + its flag word = 16 / 32 / 64 bytes; larger nodes
bypass the cache and stay in HBM)?

Not a test (pytest does not collect it) and not product code: it drives the CPU oracle (oracle/, test infrastructure) for the
dictionary stage and the LZP parse, restates the token loop's (context, symbol) sequence (ropmain/cr-coder.c:169-207 ->
cr-ppm.c:103-167) and the model bookkeeping that decides a node's size (cr-o2model.c:43-84, cr-ppm.c:66-88), and replays the
node sequence of the comprop decoder through cache organisations that fit the LDS a resident block can have at the bench's
residency (160 KB / 6 one-wave workgroups per CU = 26 KB).

    python tools/sim_node_cache.py [blocks per stream] > profiles/r04a_node_cache_sim.txt

Streams: the bench shard (enwik_like(1e8, seed 8)), the harder corpus (enwik_hard(1e8, seed 8)) and config 3's stream
(enwik_like(.., seed 9); its dictionary is picked from the first 1e8 bytes here), each after the dictionary stage — what the
decoder of the full path sees — and the bench shard's raw 64 KiB blocks (what --stage codec sees).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crlib                                    # noqa: E402
import comprox_amd                              # noqa: E402
from comprox_amd import corpus                  # noqa: E402

BLOCK = 65536


def events_of_block(o, data: bytes):
    """(ctx, sym) of every ppm_encode / ppm_decode call of a comprop block, in coding order (cro_rop_encode's loop)."""
    n = len(data)
    if n < 16:
        return []
    hist = np.bincount(np.frombuffer(data, np.uint8), minlength=256)
    esc = int(np.argmin(hist))                  # lowest value on ties, as the reference's scan
    lens = o.rop_parse(data)
    ev, ctx, pos = [], 0, 9
    for ln in lens:
        ln = int(ln)
        if ln > 1:
            ev.append((ctx, esc)); ctx = ((ctx << 8) | esc) & 0xFFFFFFFF
            ev.append((ctx, ln))
        else:
            b = data[pos]
            ev.append((ctx, b))
            if b == esc:
                ctx = ((ctx << 8) | esc) & 0xFFFFFFFF
                ev.append((ctx, 0))
        for _ in range(ln):
            ctx = ((ctx << 8) | data[pos]) & 0xFFFFFFFF
            pos += 1
    return ev


def trace_of_block(ev):
    """per step: node (16-bit context), pairs the node's line holds before the step, first visit?, order-3 key"""
    nodes = {}                                  # ctx16 -> [counts dict sym -> count, f_hit, f_esc, physical pairs]
    o3b, o3c = {}, {}
    tr = []
    for ctx, sym in ev:
        k16 = ctx & 0xFFFF
        key = (ctx ^ (ctx >> 2)) & 0x3FFFFF
        nd = nodes.get(k16)
        first = nd is None
        if first:
            nd = nodes[k16] = [{}, 1, 1, set()]
        tr.append((k16, len(nd[3]), first, key))
        cnt, pairs = nd[0], nd[3]
        pred = o3b.get(key, 0)

        def halve():
            singles = 1
            for s in list(cnt):
                cnt[s] >>= 1
                singles += cnt[s] == 1
            nd[1] = (nd[1] + 1) >> 1
            nd[2] = singles & 0xFF

        def bump(s, inc=1):
            if s == 256:
                nd[1] = (nd[1] + inc) & 0xFF
                v = nd[1]
            elif s == 257:
                nd[2] = (nd[2] + inc) & 0xFF
                v = nd[2]
            else:
                cnt[s] = (cnt.get(s, 0) + inc) & 0xFF
                pairs.add(s)
                v = cnt[s]
            if v <= 250:
                return False
            halve()
            return True

        if sym == pred:
            bump(256)
            c = o3c.get(key, 0)
            o3c[key] = c + (c < 15)
            o3b.setdefault(key, 0)
            continue
        if cnt.get(sym, 0):
            halved = bump(sym)
            if not halved and cnt[sym] == 2:
                bump(257, -1)
        else:
            halved = bump(257)
            if not halved:
                bump(sym)
        c = o3c.get(key, 0)
        c = (c > 1) + (c > 2) + (c > 4) + (c > 8)
        if c == 0:
            o3b[key] = sym
            c = 1
        o3c[key] = c
    return tr


def simulate(tr, entry_bytes, entries, ways, bitmap):
    """LRU-in-set cache of `entries` entries of `entry_bytes` (capacity entry_bytes / 2 - 2 pairs). Returns (hits, steps).
    bitmap: a 65 536-bit "node allocated in this block" map in LDS answers a first visit without a fetch (counts as a hit
    and installs the node). (The order-3 entry of such a step is still needed: its 22-bit key folds 24 context bits, so a
    new 16-bit context can meet a key that was used before.)"""
    cap = entry_bytes // 2 - 2
    sets = entries // ways
    tags = [[] for _ in range(sets)]            # most recent last
    hits = 0
    for k16, npairs, first, _ in tr:
        s = tags[(k16 * 0x9E37 >> 4) % sets]    # multiplicative index hash (best of the three round 2 tried)
        if npairs > cap or (npairs == cap and False):
            if k16 in s:
                s.remove(k16)                   # outgrew its entry: written back, bypasses from now on
            continue
        if k16 in s:
            hits += 1
            s.remove(k16); s.append(k16)
            continue
        if first and bitmap:
            hits += 1
        s.append(k16)
        if len(s) > ways:
            s.pop(0)
    return hits, len(tr)


def stream_blocks(name, nblk):
    o = crlib.Oracle()
    lib = comprox_amd.load_library()
    from test_host_dict import product_dicpick
    if name == "bench_raw":
        text = corpus.enwik_like(100_000_000, 8).tobytes()
        stride = (len(text) // BLOCK) // nblk
        return o, [text[b * stride * BLOCK:(b * stride + 1) * BLOCK] for b in range(nblk)]
    gen = {"bench": lambda: corpus.enwik_like(100_000_000, 8), "hard": lambda: corpus.enwik_hard(100_000_000, 8),
           "config3": lambda: corpus.enwik_like(100_000_000, 9)}[name]
    text = gen().tobytes()
    d = crlib.DictOracle(o)
    dic = product_dicpick(lib, text)
    d.load(dic, True)
    stride = (len(text) // BLOCK) // nblk
    return o, [d.encode(text[b * stride * BLOCK:(b * stride + 1) * BLOCK]) for b in range(nblk)]


def main():
    nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    print("LDS node cache for the comprop decoder, simulated on the oracle's symbol trace (tools/sim_node_cache.py)")
    print("hit = the step's order-2 node is in LDS when the step starts (no HBM fetch for the node); per STEP, not per node")
    for name in ("bench", "hard", "config3", "bench_raw"):
        o, blocks = stream_blocks(name, nblk)
        traces = [trace_of_block(events_of_block(o, b)) for b in blocks]
        steps = sum(map(len, traces))
        firsts = sum(t[2] for tr in traces for t in tr)
        sizes = np.array([t[1] for tr in traces for t in tr])
        nodes = sum(len({t[0] for t in tr}) for tr in traces)
        o3 = sum(len({t[3] for t in tr}) for tr in traces)
        print(f"\nstream {name}: {len(blocks)} blocks of {sum(map(len, blocks)) // len(blocks)} bytes, {steps // len(blocks)} steps per block, "
              f"{nodes // len(blocks)} distinct nodes, {o3 // len(blocks)} distinct order-3 keys per block")
        print(f"  steps whose node is new in this block: {100 * firsts / steps:.1f} %;  steps by pairs in the node before the step: "
              f"<=6: {100 * np.mean(sizes <= 6):.1f} %  <=14: {100 * np.mean(sizes <= 14):.1f} %  <=30: {100 * np.mean(sizes <= 30):.1f} %  <=62: {100 * np.mean(sizes <= 62):.1f} %")
        print(f"  {'entry':>6} {'entries':>8} {'ways':>5} {'LDS KB':>7} {'bitmap':>7} {'hit % of steps':>15}")
        for entry in (16, 32, 64):
            for entries in (400, 800, 1600):
                kb = entry * entries / 1024
                for bitmap in (False, True):
                    if kb + (8 if bitmap else 0) > 26.5:
                        continue
                    for ways in (1, 2):
                        h = sum(simulate(tr, entry, entries, ways, bitmap)[0] for tr in traces)
                        print(f"  {entry:>6} {entries:>8} {ways:>5} {kb + (8 if bitmap else 0):>7.1f} {'yes' if bitmap else 'no':>7} {100 * h / steps:>15.1f}")
        # the ceiling: an unbounded cache of nodes of any size (only first visits miss), with and without the bitmap
        print(f"  ceiling (unbounded cache, any node size): {100 * (steps - firsts) / steps:.1f} % without the bitmap, 100.0 % with it")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
