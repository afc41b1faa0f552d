"""CPU: tools/real_corpus.py -- the deterministic list of real files the GPU parity tests and `bench.py --real-files` run on.
No compression here: the selection must be reproducible, respect its budget and size bounds, take one path per inode, and a read deadline must cut
the corpus short instead of running on."""
import os

import numpy as np


def test_selection_is_deterministic_bounded_and_diverse():
    from tools import real_corpus
    a = real_corpus.select(96 << 20)
    b = real_corpus.select(96 << 20)
    assert a == b and len(a) >= 12
    sizes = [s for _, s in a]
    assert sum(sizes) <= (96 << 20) and all(real_corpus.MIN_SIZE <= s <= real_corpus.MAX_SIZE for s in sizes)
    assert len({os.path.realpath(p) for p, _ in a}) == len(a)                      # one path per file
    roots = {p.split(os.sep)[1] + "/" + p.split(os.sep)[2] for p, _ in a}
    assert len(roots) >= 3, roots                                                  # text-like roots AND the big binary trees
    bigger = real_corpus.select(128 << 20)
    assert sum(s for _, s in bigger) > sum(sizes)


def test_corpus_layout_units_and_manifest():
    from tools import real_corpus
    c = real_corpus.RealCorpus(48 << 20)
    assert len(c.paths) == len(c.off) == len(c.len) >= 12 and not c.cut_short
    assert all(int(o) % 16 == 0 for o in c.off) and int(c.off[-1] + c.len[-1]) <= c.blob.size
    for p, o, l in list(zip(c.paths, c.off, c.len))[:5]:                            # the array holds the files' bytes
        with open(p, "rb") as f:
            assert c.blob[int(o):int(o) + int(l)].tobytes() == f.read()
    uo, ul, idx = c.units(65536)
    assert int(ul.sum()) == c.total and int(ul.max()) <= 65536 and len(uo) == len(idx)
    for i in (0, len(c.paths) // 2, len(c.paths) - 1):                              # units of a file tile it exactly
        sel = np.nonzero(idx == i)[0]
        assert int(uo[sel[0]]) == int(c.off[i]) and int(ul[sel].sum()) == int(c.len[i])
    kinds = c.kinds()
    assert sum(kinds.values()) == c.total and len(c.kind_of_files()) == len(c.paths)
    m = c.manifest()
    assert len(m) == len(c.paths) and all(len(r["sha256"]) == 64 for r in m)


def test_read_deadline_cuts_the_corpus_short(tmp_path):
    from tools import real_corpus
    c = real_corpus.RealCorpus(64 << 20, deadline_s=0.0)                            # the deadline has passed before the first file starts
    assert c.cut_short and c.total == 0 and len(c.paths) == 0
    for i in range(3):                                                              # a data directory: every regular file of it, by name
        (tmp_path / ("f%d.bin" % i)).write_bytes(bytes([i]) * (1000 + i))
    d = real_corpus.RealCorpus(1 << 20, data_dir=str(tmp_path))
    assert [os.path.basename(p) for p in d.paths] == ["f0.bin", "f1.bin", "f2.bin"] and d.total == 3003
