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
