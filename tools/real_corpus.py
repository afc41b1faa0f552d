#!/usr/bin/env python3
"""A corpus of REAL files that this repository did not write -- bench / test INPUT only.

The reference's own accuracy test walks a directory of real files (/root/reference/test/test_accuracy.py:63-92; the corpora of
/root/reference/README.md:7-14). Silesia is not in the image and there is no network, but the image itself holds gigabytes of real ELF
objects, static archives, GPU code objects, Python source / bytecode and text. `select()` picks a DETERMINISTIC list of them:

* roots, in this order: /usr/share, the Python standard library, /usr/bin, /usr/lib/x86_64-linux-gnu, PyTorch's lib directory,
  /opt/rocm/lib -- each with its own byte budget, so that the big binary trees cannot crowd out text;
* regular files (no symbolic links, one path per inode) of 64 KiB .. 128 MiB;
* inside a root the largest file that fits 40 % of the root's budget goes first (long units: whole-file LZNT1 / Xpress+Huffman), then
  the candidates ordered by the MD5 of their path (deterministic, and not "the alphabetically first directory"), taken while they fit;
* `SILESIA_DIR` (or `--data-dir`): every regular file of that directory instead, sorted by name, no size filter.

The manifest (path, size, SHA-256) is what a run prints, so that two runs can be compared. Nothing here touches /root/reference.

    python tools/real_corpus.py [--max-mb 1280] [--data-dir DIR] [--sha]
"""
import argparse
import hashlib
import importlib.util
import json
import os
import sys

import numpy as np

MIN_SIZE = 64 << 10
MAX_SIZE = 128 << 20
DEFAULT_MAX_BYTES = 1100 << 20


def _torch_lib():
    spec = importlib.util.find_spec("torch")                  # (no import: the package's location is enough)
    return os.path.join(os.path.dirname(spec.origin), "lib") if spec and spec.origin else None


def default_roots():
    """(root, share of the byte budget). Text-like roots first: they are small and taken whole."""
    import sysconfig
    roots = [("/usr/share", 0.04), (sysconfig.get_paths()["stdlib"], 0.04), ("/usr/bin", 0.07), ("/usr/lib/x86_64-linux-gnu", 0.25),
             (_torch_lib(), 0.30), ("/opt/rocm/lib", 0.30)]
    return [(r, s) for r, s in roots if r and os.path.isdir(r)]


def _candidates(root, lo, hi):
    seen, out = set(), []
    for dp, dn, fn in os.walk(root):
        dn.sort()
        for f in fn:
            p = os.path.join(dp, f)
            try:
                st = os.lstat(p)
            except OSError:
                continue
            import stat
            if not stat.S_ISREG(st.st_mode) or not (lo <= st.st_size <= hi) or (st.st_dev, st.st_ino) in seen or not os.access(p, os.R_OK):
                continue
            seen.add((st.st_dev, st.st_ino))
            out.append((p, st.st_size))
    out.sort(key=lambda e: hashlib.md5(e[0].encode("utf-8", "surrogateescape")).digest())
    return out


def select(max_bytes=DEFAULT_MAX_BYTES, data_dir=None):
    """-> [(path, size)] in the order the units of a job are built from."""
    data_dir = data_dir or os.environ.get("SILESIA_DIR") or os.environ.get("MSCOMP_AMD_DATA_DIR")
    if data_dir:
        files = [(os.path.join(data_dir, f), os.path.getsize(os.path.join(data_dir, f))) for f in sorted(os.listdir(data_dir))
                 if os.path.isfile(os.path.join(data_dir, f))]
        return [e for e in files if e[1] > 0]
    picked, carry = [], 0
    for root, share in default_roots():
        budget = int(max_bytes * share) + carry
        used = 0
        cand = _candidates(root, MIN_SIZE, MAX_SIZE)
        big = max([e for e in cand if e[1] <= budget * 0.4], key=lambda e: (e[1], e[0]), default=None)   # one LONG unit per root first
        for p, s in ([big] if big else []) + [e for e in cand if e != big]:
            if used + s <= budget:
                picked.append((p, s)); used += s
        carry = budget - used                                  # what a small root leaves goes to the next one
    return picked


class RealCorpus:
    """The selected files in ONE numpy uint8 array (every file starts 16-byte aligned) + offsets / lengths / names."""

    def __init__(self, max_bytes=DEFAULT_MAX_BYTES, data_dir=None, deadline_s=None):
        """deadline_s: stop STARTING new files that many seconds after the first (a fresh box pages its image in lazily, 4-12 MB/s in total: a test run
        must not spend its whole time limit reading); the files read by then are the corpus, `self.cut_short` says so."""
        import time
        sel = select(max_bytes, data_dir)
        t_end = None if deadline_s is None else time.monotonic() + float(deadline_s)
        self.cut_short = False
        self.source = ("dir:" + data_dir) if data_dir else ("dir:" + os.environ["SILESIA_DIR"] if os.environ.get("SILESIA_DIR") else "image files")
        offs, pos, kept = [], 0, []
        for p, s in sel:
            offs.append(pos); pos += (s + 15) // 16 * 16; kept.append((p, s))
        self.blob = np.zeros(pos + 64, dtype=np.uint8)
        view = memoryview(self.blob)

        def read(job):                                         # (a fresh box pages the image in lazily: one reader gets ~5 MB/s, so many read at once)
            (p, s), o = job
            if t_end is not None and time.monotonic() > t_end:
                self.cut_short = True
                return -1
            try:
                with open(p, "rb") as f:
                    return f.readinto(view[o:o + s])
            except OSError:
                return -1
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(int(os.environ.get("MSCOMP_AMD_REAL_READERS", "48"))) as ex:
            got = list(ex.map(read, zip(kept, offs)))
        self.paths, lens, offs2 = [], [], []
        for (p, s), o, n in zip(kept, offs, got):
            if n < 0:
                continue
            if n != s:                                         # a file that changed under us: take what was read
                self.blob[o + n:o + s] = 0
            self.paths.append(p); lens.append(n); offs2.append(o)
        self.off = np.array(offs2, dtype=np.uint64)
        self.len = np.array(lens, dtype=np.uint64)
        self.total = int(self.len.sum())

    def units(self, unit=None):
        """(offsets, lengths, file index of every unit): whole files, or every file cut into independent `unit`-byte pieces."""
        if unit is None:
            return self.off, self.len, np.arange(len(self.off))
        offs, lens, idx = [], [], []
        for i, (o, l) in enumerate(zip(self.off, self.len)):
            s = np.arange(0, int(l), unit, dtype=np.uint64)
            offs.append(s + o); lens.append(np.minimum(unit, int(l) - s).astype(np.uint64)); idx.append(np.full(len(s), i))
        return np.concatenate(offs), np.concatenate(lens), np.concatenate(idx)

    def manifest(self, sha=True):
        rows = []
        for p, o, l in zip(self.paths, self.off, self.len):
            row = {"path": p, "size": int(l)}
            if sha:
                row["sha256"] = hashlib.sha256(self.blob[int(o):int(o) + int(l)].tobytes()).hexdigest()
            rows.append(row)
        return rows

    def kind_of_files(self):
        """the kind of every file, in file order (the classes of kinds())"""
        out = []
        for p, o in zip(self.paths, self.off):
            head = self.blob[int(o):int(o) + 8].tobytes()
            if head[:4] == b"\x7fELF":
                out.append("elf")
            elif head[:8] == b"!<arch>\n":
                out.append("ar")
            elif p.endswith((".pyc",)):
                out.append("pyc")
            elif p.endswith((".py", ".txt", ".xml", ".json", ".html", ".js", ".pl", ".pm", ".h", ".hpp", ".ids", ".css", ".rst", ".md", ".cmake", ".yaml")):
                out.append("text")
            elif p.endswith((".dat", ".co", ".hsaco", ".kdb", ".db", ".bc")):
                out.append("gpu code / tables")
            else:
                out.append("other")
        return out

    def kinds(self):
        """bytes per file kind (by suffix / magic): what the corpus is made of"""
        k = {}
        for kind, l in zip(self.kind_of_files(), self.len):
            k[kind] = k.get(kind, 0) + int(l)
        return k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-mb", type=int, default=DEFAULT_MAX_BYTES >> 20)
    ap.add_argument("--data-dir", default=None)
    ap.add_argument("--sha", action="store_true", help="SHA-256 of every file (reads them all)")
    args = ap.parse_args()
    if args.sha:
        c = RealCorpus(args.max_mb << 20, args.data_dir)
        doc = {"files": len(c.paths), "bytes": c.total, "kinds": c.kinds(), "manifest": c.manifest()}
    else:
        sel = select(args.max_mb << 20, args.data_dir)
        doc = {"files": len(sel), "bytes": sum(s for _, s in sel), "manifest": [{"path": p, "size": s} for p, s in sel]}
    json.dump(doc, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
