"""bench.py's host-side pieces that need no GPU: the cpu_baseline leg over the leg's own unit list and the host-reference digests the
parity gate uses on data that is not the synthetic corpus (checker code: oracle/ -- allowed in bench.py's cpu_baseline / gate only)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from ms_compress_amd import corpus             # noqa: E402
from oracle import loader                      # noqa: E402


class SmallCorpus:
    """12 short members laid out like bench.Corpus (no device side)"""

    def __init__(self):
        self.files = [corpus.file_bytes(i, 150_000 + 7_001 * i) for i in range(12)]
        self.flen = np.array([len(f) for f in self.files], np.uint64)
        self.foff = np.zeros(12, np.uint64); self.foff[1:] = np.cumsum(self.flen)[:-1]
        self.total = int(self.flen.sum())
        self._blob = np.concatenate(self.files)

    def blob(self):
        return self._blob


def _compress(fmt, data):
    fn = loader.ref_compress if loader.load_ref() is not None else loader.oracle_compress
    st, out = fn(fmt, data)
    assert st == 0
    return out


def test_host_reference_digests_match_one_shot_calls():
    cor = SmallCorpus()
    for fmt in (2, 3, 4):
        got = bench._host_reference_digests(fmt, cor)
        assert len(got) == 12
        for i in (0, 5, 11):
            f = cor.files[i].tobytes()
            out = _compress(fmt, f) if fmt != 3 else b"".join(_compress(3, f[o:o + 65536]) for o in range(0, len(f), 65536))
            assert got[i] == (len(out), hashlib.sha256(out).hexdigest()), (fmt, i)


def test_cpu_baseline_runs_the_legs_own_units():
    cor = SmallCorpus()
    for fmt, words in ((2, "whole-file units"), (3, "independent 64 KiB units"), (4, "whole-file units")):
        r = bench.cpu_baseline(fmt, cor, budget_s=0.5)
        assert r["value"] > 0 and r["single_thread"]["value"] > 0 and r["cores"] >= 1 and words in r["sample"], r
        assert r["balanced_value"] >= r["value"] * 0.5, r          # (bytes x threads / thread-seconds inside ms_compress: never far below the one-pass figure)
    r = bench.cpu_baseline(2, cor, budget_s=0.5, sa_dict=True)
    assert r is None or (r["value"] > 0 and r["kind"] == "reference")


def test_the_printed_line_stays_short_and_parseable():
    """bench.short_line on the whole document of the last profiled run (profiles/r06_bench_extra.json): what rank 0 prints must stay under 4 KB
    (the driver's record keeps the last 8 KB of stdout: round 4's 20.8 KB line went unparsed), carry the contract's keys and hold no prose."""
    import json
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_extra.json")))
    line = bench.short_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 4096 and "extra" not in line
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert set(line["roofline"]) == {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_per_launch", "launches_per_step", "algorithmic_bytes_per_launch"}
    assert 0 < line["roofline"]["frac"] < 1 and line["cpu_baseline"]["balanced_value"] > 0
    assert "configs[4]" in line["config"]["workload"]

    def walk(x):
        if isinstance(x, dict):
            for v in x.values():
                walk(v)
        elif isinstance(x, str):
            assert len(x) <= 200
    walk(line)
