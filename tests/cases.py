"""Seeded input families shared by the parity tests (same classes the survey's fuzz used)."""
import random

EDGE_SIZES = list(range(0, 70)) + [95, 96, 97, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 1000, 4095, 4096, 4097,
                                   8191, 8192, 8193, 8200, 12288, 16384, 20000, 40000, 65535, 65536, 65537, 65538, 70000,
                                   131072, 131073]
WORDS = [b"the ", b"quick ", b"brown ", b"fox ", b"jumps ", b"over ", b"lazy ", b"dog ", b"\n"]


def family(kind, n, rnd):
    if kind == "random":
        return bytes(rnd.getrandbits(8) for _ in range(n))
    if kind == "two":
        return bytes(rnd.choice(b"ab") for _ in range(n))
    if kind == "run":
        return bytes([7]) * n
    if kind == "words":
        return b"".join(rnd.choice(WORDS) for _ in range(n // 3 + 2))[:n]
    if kind == "lz":
        out = bytearray()
        while len(out) < n:
            if len(out) > 10 and rnd.random() < 0.5:
                off = rnd.randint(1, min(len(out), 9000))
                for _ in range(rnd.randint(3, 600)):
                    out.append(out[-off])
            else:
                out.append(rnd.getrandbits(8))
        return bytes(out[:n])
    raise ValueError(kind)


KINDS = ["random", "two", "run", "words", "lz"]


def edge_cases(seed=1, sizes=EDGE_SIZES, kinds=KINDS):
    rnd = random.Random(seed)
    return [family(k, n, rnd) for n in sizes for k in kinds]


def mixed_buffer(seed=2):
    """text + 100 000 zeros + 70 000 random + 'ab'*40000 + text (lagging fill, >64 KiB match, XH fallback)."""
    rnd = random.Random(seed)
    return (b"some text here " * 7000 + bytes(100000) + bytes(rnd.getrandbits(8) for _ in range(70000))
            + b"ab" * 40000 + b"more text over there " * 3000)


def decode_streams(fmt, compress, seed=5, n_corrupt=60):
    """(stream, capacity) pairs for the decompressors: valid streams with exact / larger / smaller capacities, streams
    with terminators and trailing bytes, truncated streams, concatenated streams (short chunks in the middle for LZNT1) and
    randomly corrupted streams. ``compress(data) -> bytes`` is the checker's encoder."""
    rnd = random.Random(seed)
    plain = [family(k, n, rnd) for n in (0, 1, 2, 3, 17, 100, 4095, 4096, 4097, 8192, 12288 + 5, 70000) for k in ("words", "run", "random", "lz")]
    plain.append(mixed_buffer()[60000:60000 + 150000])
    out = []
    comps = [(d, compress(d)) for d in plain]
    for d, c in comps:
        for cap in sorted({len(d), len(d) + 1, len(d) + 5000, max(0, len(d) - 1), len(d) // 2, 0}):
            out.append((c, cap))
        out.append((c + b"\0\0", len(d)))
        out.append((c + b"\0\0", len(d) + 1))
        out.append((c + b"\0\0\0", len(d) + 10))
        out.append((c + b"\0", len(d) + 10))
        out.append((c + b"\x07", len(d) + 10))
        for k in sorted({1, 2, 3, len(c) // 3, len(c) // 2, len(c) - 2, len(c) - 1}):
            if 0 < k < len(c):
                out.append((c[:k], len(d) + 10))
    for _ in range(12):                                      # concatenations
        (d1, c1), (d2, c2), (d3, c3) = rnd.choice(comps), rnd.choice(comps), rnd.choice(comps)
        out.append((c1 + c2 + c3, len(d1) + len(d2) + len(d3)))
        out.append((c1 + c2 + c3, len(d1) + len(d2) + len(d3) + 4096))
        out.append((c1 + c2, len(d1) + len(d2) // 2))
    big = [x for x in comps if len(x[1]) > 200]
    for _ in range(n_corrupt):                               # corruptions
        d, c = rnd.choice(big)
        b = bytearray(c)
        for _ in range(rnd.choice((1, 1, 2, 5))):
            i = rnd.randrange(len(b))
            b[i] = rnd.choice((b[i] ^ (1 << rnd.randrange(8)), rnd.getrandbits(8), 0, 0xFF))
        out.append((bytes(b), len(d) + rnd.choice((0, 0, 100, 5000))))
    return out


def huff_histograms(n=240, seed=11):
    """512-bin symbol histograms for the Huffman builder (HuffmanEncoder<15,512>::CreateCodes): sparse / tiny counts (ties everywhere),
    geometric counts (trees deeper than 15: the rescale loop runs 1-3 times), flat literal-heavy chunks with EOS, Zipf-shaped counts
    with a match-symbol tail like real chunks, all-zero but one, all equal."""
    rnd = random.Random(seed)
    out = []
    for case in range(n):
        kind = case % 8
        if kind == 0:
            c = [rnd.randint(0, 3) * rnd.randint(0, 300) for _ in range(512)]
        elif kind == 1:
            c = [int(2 ** rnd.uniform(0, 16)) if rnd.random() < 0.3 else 0 for _ in range(512)]
        elif kind == 2:
            c = [rnd.randint(200, 300) for _ in range(256)] + [0] * 256
            c[256] = 1
        elif kind == 3:
            c = [1 if rnd.random() < 0.05 else 0 for _ in range(512)]
        elif kind == 4:
            c = [int(40000 / (1 + i) ** rnd.uniform(0.8, 1.6)) for i in range(256)]
            rnd.shuffle(c)
            c += [int(3000 / (1 + (i % 16)) / (1 + i // 16)) if rnd.random() < 0.7 else 0 for i in range(256)]
        elif kind == 5:
            c = [0] * 512
            c[rnd.randrange(512)] = rnd.randint(1, 65536)
        elif kind == 6:
            v = rnd.randint(0, 5)
            c = [v] * 512
        else:
            c = [int(1.6 ** (i % 24)) if i % 3 == 0 else 0 for i in range(512)]         # Fibonacci-like growth: very deep tree
            rnd.shuffle(c)
        if sum(c) == 0 and kind != 6:
            c[0] = 1
        out.append(c)
    return out


def periodic_with_mutations(n=65536, period=3756, seed=9, gap=(150, 400)):
    """A random block of `period` bytes repeated, one byte changed every 150-400 positions: every position finds a match at distance `period`
    (or twice that) that runs to the next change -- hundreds of matches of ONE distance, each longer than the 112 bytes a lane of the lazy Xpress
    finder extends by itself, extended by all waves of a block at once (the shape of the real files -- 64 KiB pieces of GPU code tables --
    on which its long-match cache gave wave-divergent answers; csrc/xpress_lazy.hip)."""
    rnd = random.Random(seed)
    base = bytes(rnd.getrandbits(8) for _ in range(period))
    out = bytearray((base * (n // period + 1))[:n])
    p = period + rnd.randint(*gap)
    while p < n:
        out[p] = (out[p] + 1 + rnd.getrandbits(7)) & 0xFF
        p += rnd.randint(*gap)
    return bytes(out)


def few_distances(n=65536, dists=(3756, 7426, 3704, 486, 524), seed=11, run=(120, 320)):
    """An LZ-generated stream whose copies come from a handful of distances only and run 120-320 bytes each, one fresh byte between them: the
    greedy parse is long matches of alternating distances (what 64 KiB pieces of the GPU code tables in the image look like), so the blocks of the
    lazy Xpress finder extend long matches of several distances at once and its 4-entry long-match cache is written and read all the time."""
    rnd = random.Random(seed)
    out = bytearray(rnd.getrandbits(8) for _ in range(max(dists) + 64))
    while len(out) < n:
        d = rnd.choice(dists)
        for _ in range(rnd.randint(*run)):
            out.append(out[-d])
        out.append(rnd.getrandbits(8))
    return bytes(out[:n])
