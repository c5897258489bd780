"""Simulation: how often could the comprop decoder have the NEXT step's order-2 node already in flight?

Not a test (pytest does not collect it) and not product code. The decoder's step is one dependent HBM fetch (the node of the
16-bit context, known only when the previous symbol is) + ~180 ns of work. A small table in LDS "context -> the symbol that
followed it last time" lets a step issue, together with the real fetch for step t+1, a SPECULATIVE fetch for step t+2 (context
= {s_t, guess of s_t+1}); if the guess holds, step t+2 finds its node already there. This script measures, on the oracle's
symbol trace (tools/sim_node_cache.py's streams), how often the guess holds per step, by table size, and what a fetch round
yields at speculation depth 1-3.

    python tools/sim_guess.py [blocks per stream] > profiles/r04v_guess_sim.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from sim_node_cache import events_of_block, stream_blocks       # noqa: E402


def run(ev, bits, depth):
    """returns (steps, steps whose node was in flight one fetch early at depth 1, .. at `depth`)"""
    mask = (1 << bits) - 1
    T = np.zeros(1 << bits, np.int16) - 1

    def slot(c16):
        return ((c16 * 0x9E37) >> 3) & mask if bits < 16 else c16

    n = len(ev)
    covered = np.zeros(n, np.uint8)                 # 1 = this step's fetch was issued speculatively and right
    rounds = 0
    i = 0
    ctxs = [e[0] & 0xFFFF for e in ev]
    syms = [e[1] for e in ev]
    # the table learns in decode order; a guess made at step i only uses what steps < = i taught it
    learned_upto = -1

    def learn(upto):
        nonlocal learned_upto
        for j in range(learned_upto + 1, upto + 1):
            T[slot(ctxs[j])] = syms[j]
        learned_upto = upto

    while i < n:
        # a fetch round: step i's node is fetched for real (issued when step i-1's symbol was known); with it, up to `depth`
        # guessed successors
        rounds += 1
        learn(i - 1)
        c = ctxs[i]
        full = ev[i][0]
        k = 1
        while k <= depth and i + k < n:
            g = int(T[slot(c)])
            if g < 0:
                break
            full = ((full << 8) | g) & 0xFFFFFFFF
            c = full & 0xFFFF
            if ev[i + k][0] != full or syms[i + k - 1] != g:
                break
            k += 1
        i += k
    return n, rounds


def main():
    nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    print("speculative node fetch for the comprop decoder, simulated on the oracle's symbol trace (tools/sim_guess.py)")
    print("guess table: context (16 bits, hashed to the table) -> symbol that followed it last; steps per fetch round = how many")
    print("steps one HBM round trip serves when the guessed successors' nodes are fetched beside the real one")
    for name in ("bench", "hard", "config3", "bench_raw"):
        o, blocks = stream_blocks(name, nblk)
        evs = [events_of_block(o, b) for b in blocks]
        steps = sum(map(len, evs))
        print(f"\nstream {name}: {len(blocks)} blocks, {steps // len(blocks)} steps per block")
        print(f"  {'table':>12} {'LDS KB':>7} {'depth':>6} {'steps per round':>16} {'step at 393+180 ns':>20}")
        for bits in (12, 13, 14, 16):
            for depth in (1, 2, 3):
                n = r = 0
                for ev in evs:
                    a, b = run(ev, bits, depth)
                    n += a; r += b
                spr = n / r
                print(f"  {1 << bits:>12} {(1 << bits) / 1024:>7.0f} {depth:>6} {spr:>16.3f} {393 / spr + 180:>17.0f} ns")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
