"""Seeded synthetic byte streams shaped like the corpora BASELINE.json names (none are on disk).

`enwik_like(n, seed)` follows SURVEY.md §8d: Zipf(1.1) over a 50 000-word synthetic lowercase
vocabulary (lengths 2-12), sentence case after ". ", punctuation from " ,.:;", about 3 % XML-ish
tags (<page>..</page>, [[..]]), a newline roughly every 80 characters. seed 8 / n = 10^8 stands in
for enwik8, seed 9 / n = 10^9 for enwik9. `markov2(n, block_index)` is config 5's order-2 Markov
block. Everything is numpy-vectorised and deterministic (splitmix64), so hashes of oracle output
over these streams are stable fixtures.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def splitmix(seed: int, count: int, start: int = 0) -> np.ndarray:
    """Outputs start+1 .. start+count of splitmix64 with initial state `seed`."""
    idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


_VOCAB_WORDS = 50000
_VOCAB_CACHE = {}


def _vocab(seed: int):
    if seed in _VOCAB_CACHE:
        return _VOCAB_CACHE[seed]
    r = splitmix(seed * 0x1234567 + 99, _VOCAB_WORDS * 13)
    lens = (r[:_VOCAB_WORDS] % np.uint64(11)).astype(np.int64) + 2
    letters = np.frombuffer(b"etaoinshrdlucmfwypvbgkjqxz", dtype=np.uint8)
    u = r[_VOCAB_WORDS:].astype(np.float64) / 2.0 ** 64
    chars = letters[(u * u * 26).astype(np.int64)].reshape(_VOCAB_WORDS, 12)
    w = 1.0 / np.arange(1, _VOCAB_WORDS + 1, dtype=np.float64) ** 1.1
    cdf = np.cumsum(w / w.sum())
    _VOCAB_CACHE[seed] = (chars, lens, cdf)
    return _VOCAB_CACHE[seed]


_PAGE_OPEN = np.frombuffer(b"<page>", dtype=np.uint8)
_PAGE_CLOSE = np.frombuffer(b"</page>", dtype=np.uint8)


_HARD = 1 << 20                      # seeds >= this select the harder corpus (enwik_hard): a vocabulary of its own
_HARD_WORDS = 262144
_HARD_WIDTH = 28
_MARKUP = [b"<ref>", b"</ref>", b"\x27\x27", b"\x27\x27\x27", b"==", b"&amp;", b"&quot;", b"<br />", b"{|", b"|}", b"|-", b"#REDIRECT", b"* ",
           b"===", b"&nbsp;", b"<!--", b"-->"]


def _hard_vocab(seed: int):
    """Vocabulary of the harder corpus: 262 144 words under a flatter Zipf law (exponent 1.0: a dictionary of 25 000 words
    covers far less of the text than it does of enwik_like's) and, behind them, 8 192 strings no dictionary word can match:
    numbers, dates, URLs, wiki markup, table cells, capitalised names."""
    if seed in _VOCAB_CACHE:
        return _VOCAB_CACHE[seed]
    V, P, W = _HARD_WORDS, 8192, _HARD_WIDTH
    r = splitmix(seed * 0x1234567 + 99, V * 15 + P * 40)
    lens = np.empty(V + P, dtype=np.int64)
    lens[:V] = (r[:V] % np.uint64(13)).astype(np.int64) + 2
    letters = np.frombuffer(b"etaoinshrdlucmfwypvbgkjqxz", dtype=np.uint8)
    u = r[V:V + V * 14].astype(np.float64) / 2.0 ** 64
    chars = np.full((V + P, W), 32, dtype=np.uint8)
    chars[:V, :14] = letters[(u * u * 26).astype(np.int64)].reshape(V, 14)
    rs = r[V * 15:].reshape(P, 40)

    def word(i, j):
        w = int(rs[i, j] % np.uint64(4000))
        return bytes(chars[w, :lens[w]])

    for i in range(P):
        kind, a, b = int(rs[i, 0] % np.uint64(100)), int(rs[i, 1] >> np.uint64(8)) & 0x7fffffff, int(rs[i, 2] >> np.uint64(8)) & 0x7fffffff
        if kind < 10:                                      # years
            t = str(1000 + a % 1025).encode()
        elif kind < 16:                                    # decimals
            t = ("%d.%d" % (a % 1000, b % 100)).encode()
        elif kind < 30:                                    # numbers with thousands separators
            t = "{:,}".format(a % (10 ** (1 + b % 8))).encode()
        elif kind < 36:                                    # dates
            t = ("%d-%02d-%02d" % (1800 + a % 225, 1 + b % 12, 1 + (a >> 8) % 28)).encode()
        elif kind < 40:
            t = ("ISBN %d-%d-%d" % (a % 10, b % 100000, a % 1000)).encode()
        elif kind < 48:                                    # URLs
            t = b"http://www." + word(i, 3) + (b".org/", b".com/", b".net/wiki/")[a % 3] + word(i, 4)
        elif kind < 58:                                    # links and templates
            t = b"[[" + word(i, 3) + b"|" + word(i, 4) + b"]]" if a % 2 else b"{{" + word(i, 3) + b"|" + word(i, 4)[:3] + b"=" + str(b % 100).encode() + b"}}"
        elif kind < 64:
            t = b"<ref name=" + word(i, 3) + b"/>" if a % 2 else b"[[Category:" + word(i, 3) + b"]]"
        elif kind < 76:                                    # bare markup
            t = _MARKUP[a % len(_MARKUP)]
        elif kind < 86:                                    # table cells
            t = ("| %d || %d.%d || %d" % (a % 500, b % 90, a % 10, b % 7000)).encode()
        else:                                              # capitalised names (not sentence starts)
            t = word(i, 3).capitalize()
        t = t[:W]
        chars[V + i, :len(t)] = np.frombuffer(t, dtype=np.uint8)
        lens[V + i] = len(t)
    w = 1.0 / np.arange(1, V + 1, dtype=np.float64) ** 1.0
    cdf = np.cumsum(w / w.sum())
    _VOCAB_CACHE[seed] = (chars, lens, cdf)
    return _VOCAB_CACHE[seed]


def _enwik_tokens(seed: int, word0: int, k: int):
    """Words word0 .. word0 + k - 1 of the stream: vocabulary ids, token kinds, separators and token lengths."""
    r = splitmix(seed, 2 * k, start=2 * word0)
    if seed >= _HARD:
        _, lens, cdf = _hard_vocab(seed)
        ids = np.searchsorted(cdf, r[:k].astype(np.float64) / 2.0 ** 64).clip(0, _HARD_WORDS - 1)
        pick = ((r[k:] >> np.uint64(20)) % np.uint64(1000)).astype(np.int64)     # 28 % of the tokens come from the 8 192 other strings
        ids = np.where(pick < 280, _HARD_WORDS + ((r[k:] >> np.uint64(34)) % np.uint64(8192)).astype(np.int64), ids)
    else:
        _, lens, cdf = _vocab(seed)
        ids = np.searchsorted(cdf, r[:k].astype(np.float64) / 2.0 ** 64).clip(0, _VOCAB_WORDS - 1)
    ctl = (r[k:] % np.uint64(1000)).astype(np.int64)
    wl = lens[ids]
    kind = np.where(ctl < 15, 1, np.where(ctl < 30, 2, 0))              # 1 <page>, 2 [[ ]]
    pre = np.where(kind == 1, 6, np.where(kind == 2, 2, 0))
    post = np.where(kind == 1, 7, np.where(kind == 2, 2, 0))
    sep = np.select([ctl >= 900, ctl >= 820, ctl >= 805, ctl >= 790], [1, 2, 3, 4], 0)   # . , ; : space
    sep_len = np.where(sep == 0, 1, 2)
    tok = pre + wl + post + sep_len
    return ids, wl, kind, pre, post, sep, tok


def _enwik_chunk(seed: int, word0: int, k: int, col_carry: int, cap_carry: bool):
    """The bytes of words word0 .. word0 + k - 1, given the column and the sentence state the words in front left.
    Returns (bytes, column carried on, sentence state carried on)."""
    chars = (_hard_vocab(seed) if seed >= _HARD else _vocab(seed))[0]
    ids, wl, kind, pre, post, sep, tok = _enwik_tokens(seed, word0, k)
    cum = np.cumsum(tok) + col_carry
    nl = (cum // 80) > ((cum - tok) // 80)
    total = tok + nl
    start = np.cumsum(total) - total
    size = int(start[-1] + total[-1])
    buf = np.full(size, 32, dtype=np.uint8)
    # word letters
    widx = np.repeat(np.arange(k), wl)
    woff = np.arange(int(wl.sum())) - np.repeat(np.cumsum(wl) - wl, wl)
    buf[(start + pre)[widx] + woff] = chars[ids[widx], woff]
    # sentence case: first letter of a plain word following ". "
    cap = np.empty(k, dtype=bool)
    cap[0] = cap_carry
    cap[1:] = sep[:-1] == 1
    capw = cap & (kind == 0)
    if seed >= _HARD:
        capw &= ids < _HARD_WORDS                    # (numbers, markup and names stay as they are)
    buf[(start + pre)[capw]] -= 32
    # tags
    for j in range(6):
        buf[start[kind == 1] + j] = _PAGE_OPEN[j]
    for j in range(7):
        buf[(start + pre + wl)[kind == 1] + j] = _PAGE_CLOSE[j]
    buf[start[kind == 2]] = ord("[")
    buf[start[kind == 2] + 1] = ord("[")
    buf[(start + pre + wl)[kind == 2]] = ord("]")
    buf[(start + pre + wl)[kind == 2] + 1] = ord("]")
    # separators
    sp = start + pre + wl + post
    punct = np.array([32, ord("."), ord(","), ord(";"), ord(":")], dtype=np.uint8)
    buf[sp] = punct[sep]
    # (second separator byte is already a space); newline
    buf[(start + tok)[nl]] = 10
    return buf, int(cum[-1] % 80), bool(sep[-1] == 1)


def enwik_like(n: int, seed: int = 8, chunk_words: int = 1 << 20) -> np.ndarray:
    """n bytes of enwik-shaped text as a uint8 array."""
    out = np.empty(n + 64, dtype=np.uint8)
    filled = 0
    word0 = 0
    col_carry = 0
    cap_carry = True
    while filled < n:
        buf, col_carry, cap_carry = _enwik_chunk(seed, word0, chunk_words, col_carry, cap_carry)
        take = min(int(buf.size), n - filled)
        out[filled:filled + take] = buf[:take]
        filled += take
        word0 += chunk_words
    return out[:n]


def enwik_hard(n: int, seed: int = 8, chunk_words: int = 1 << 20) -> np.ndarray:
    """n bytes of the HARDER enwik-shaped stream: the same sentence / tag / line structure as enwik_like, but a vocabulary of
    262 144 words under Zipf(1.0) and more than one token in four a number, date, URL, piece of wiki markup, table cell or capitalised
    name — text the static dictionary covers far less of. A second opinion on how far real enwik8 may sit from the headline
    corpus; the headline stays enwik_like(1e8, seed 8)."""
    return enwik_like(n, _HARD + seed, chunk_words)


def enwik_hard_to_file(path: str, n: int, seed: int = 8, workers: int = 0, chunk_words: int = 1 << 20) -> None:
    enwik_like_to_file(path, n, _HARD + seed, workers, chunk_words)


def _enwik_stats_job(a):
    seed, word0, k = a
    tok = _enwik_tokens(seed, word0, k)
    return int(tok[6].sum()), bool(tok[5][-1] == 1)


def _enwik_fill_job(a):
    seed, word0, k, col, cap, path, n, at = a
    buf = _enwik_chunk(seed, word0, k, col, cap)[0]
    take = min(int(buf.size), n - at)
    if take > 0:
        np.memmap(path, dtype=np.uint8, mode="r+", shape=(n,))[at:at + take] = buf[:take]
    return take


def enwik_like_to_file(path: str, n: int, seed: int = 8, workers: int = 0, chunk_words: int = 1 << 20) -> None:
    """The same n bytes as enwik_like(n, seed), written to `path` by a pool of processes. What a chunk of words needs
    from the words in front of it — the output column modulo 80 and whether a sentence just ended — follows from the
    chunks' token-length sums alone (every token is shorter than a line, so a chunk of S token bytes starting in column
    c holds (S + c) // 80 newlines), which a first parallel pass computes; the chunks are then generated independently
    and written at their offsets."""
    import multiprocessing as mp
    import os
    workers = workers or max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    (_hard_vocab if seed >= _HARD else _vocab)(seed)   # built once, inherited by the forked workers
    with open(path, "wb") as f:
        f.truncate(n)
    if n == 0:
        return
    with mp.get_context("fork").Pool(workers) as pool:
        plan, at, col, cap, word0 = [], 0, 0, True, 0
        while at < n:
            wave = [(seed, word0 + j * chunk_words, chunk_words) for j in range(workers * 2)]
            for (_, w0, k), (s_tok, last_period) in zip(wave, pool.map(_enwik_stats_job, wave)):
                if at >= n:
                    break
                plan.append((seed, w0, k, col, cap, path, n, at))
                at += s_tok + (s_tok + col) // 80
                col = (s_tok + col) % 80
                cap = last_period
            word0 += len(wave) * chunk_words
        pool.map(_enwik_fill_job, plan)


def _mix(a, b, i):
    with np.errstate(over="ignore"):
        x = (a.astype(np.uint32) * np.uint32(0x10001) + b.astype(np.uint32) * np.uint32(0x101)
             + i.astype(np.uint32) * np.uint32(0x9E3779B1) + np.uint32(0x7F4A7C15))
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x2C1B3C6D)
        x ^= x >> np.uint32(12)
    return (x & np.uint32(0xFF)).astype(np.uint8)


_SUCC = None


def _succ_table():
    """succ[a, b, i] = i-th successor byte of context (a, b)."""
    global _SUCC
    if _SUCC is None:
        a, b, i = np.meshgrid(np.arange(256), np.arange(256), np.arange(8), indexing="ij")
        _SUCC = _mix(a, b, i)
    return _SUCC


def markov2(n: int, block_index: int = 0) -> np.ndarray:
    """One block of config 5's order-2 Markov stream: seed = golden ^ block_index, successor i of
    context (a,b) with probability 1/2,1/4,...,1/128,1/128; first two bytes = low bytes of the seed."""
    seed = 0x9E3779B97F4A7C15 ^ block_index
    lb = (splitmix(seed, n) & np.uint64(0xFF)).astype(np.int64)
    choice = np.select([lb < 128, lb < 192, lb < 224, lb < 240, lb < 248, lb < 252, lb < 254],
                       [0, 1, 2, 3, 4, 5, 6], 7).tolist()
    succ = _succ_table()
    out = bytearray(n)
    if n > 0:
        out[0] = seed & 0xFF
    if n > 1:
        out[1] = (seed >> 8) & 0xFF
    if n > 2:
        a, b = out[0], out[1]
        flat = succ.reshape(-1).tolist()
        for k in range(2, n):
            c = flat[(a << 11) | (b << 3) | choice[k]]
            out[k] = c
            a, b = b, c
    return np.frombuffer(bytes(out), dtype=np.uint8)


def markov2_blocks(nblocks: int, first_index: int = 0, block: int = 65536, device="cpu"):
    """`nblocks` consecutive blocks of config 5's stream (block indices first_index ..), as a torch uint8 tensor of
    shape (nblocks, block) on `device` — the same bytes as markov2(block, index) per row, generated for all blocks at
    once (one step of the chain per iteration, vectorised across blocks), so that a GiB takes seconds on the GPU."""
    import torch
    dev = torch.device(device)
    M64 = (1 << 64) - 1

    def s64(v):                                  # python int (mod 2^64) -> the int64 with the same bits
        v &= M64
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(x, k):                               # logical shift right on int64 tensors
        return (x >> k) & ((1 << (64 - k)) - 1)

    idx = torch.arange(first_index, first_index + nblocks, dtype=torch.int64, device=dev)
    seed = torch.full_like(idx, s64(0x9E3779B97F4A7C15)) ^ idx                    # golden ^ block_index
    out = torch.empty((nblocks, block), dtype=torch.uint8, device=dev)
    if block > 0:
        out[:, 0] = (seed & 0xFF).to(torch.uint8)
    if block > 1:
        out[:, 1] = ((seed >> 8) & 0xFF).to(torch.uint8)
    succ = torch.from_numpy(_succ_table().reshape(-1).astype(np.int64)).to(dev)
    thr = torch.tensor([128, 192, 224, 240, 248, 252, 254], dtype=torch.int64, device=dev)
    a = out[:, 0].to(torch.int64) if block > 0 else None
    b = out[:, 1].to(torch.int64) if block > 1 else None
    gold = s64(0x9E3779B97F4A7C15)
    m1, m2 = s64(0xBF58476D1CE4E5B9), s64(0x94D049BB133111EB)
    STEP = 4096                                  # splitmix outputs are computed a slab of positions at a time
    for k0 in range(2, block, STEP):
        k1 = min(block, k0 + STEP)
        pos = torch.arange(k0 + 1, k1 + 1, dtype=torch.int64, device=dev)          # output index k uses splitmix draw k + 1
        z = seed[:, None] + pos[None, :] * gold
        z = (z ^ lsr(z, 30)) * m1
        z = (z ^ lsr(z, 27)) * m2
        lb = (z ^ lsr(z, 31)) & 0xFF
        choice = (lb[:, :, None] >= thr[None, None, :]).sum(dim=2)                 # 0..7
        for j in range(k1 - k0):
            c = succ[(a << 11) | (b << 3) | choice[:, j]]
            out[:, k0 + j] = c.to(torch.uint8)
            a, b = b, c
    return out
