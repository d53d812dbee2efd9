"""The multi-GPU path's host logic on CPUs: two processes (gloo), block-range shards, one collective
of the verify counters.  Per-shard verification is done by the CPU oracle here (the GPU engine takes
its place on GPUs; bench.py --gpus N runs that)."""
import os
import socket
import subprocess
import sys

import numpy as np

from terarkdb_amd.sharding import shard_range

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle_lib as O
from terarkdb_amd.sharding import shard_range, allreduce_totals, gather_verify_counts
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n, blen = 1003, 4096
corpus = O.splitmix(n * blen, 55)                      # every rank can regenerate the corpus from the seed
stored = O.sst_trailer_batch(corpus, blen, blen, n, 0)
bad = [17, 500, 501, 1002]
for b in bad:
    corpus[b * blen + 9] ^= 0x80
lo, hi = shard_range(n, rank, world)
mine = O.sst_trailer_batch(corpus[lo * blen: hi * blen], blen, blen, hi - lo, 0)
wrong = np.nonzero(mine != stored[lo:hi])[0]
t = allreduce_totals(len(wrong), hi - lo, int(wrong[0]) if len(wrong) else None, lo)
assert (t.mismatches, t.checked, t.first_bad) == (len(bad), n, 17), t
# the same exchange straight from an engine-style counter block {mismatches, first bad local or ~0, checked}
block = torch.tensor([len(wrong), int(wrong[0]) if len(wrong) else -1, hi - lo], dtype=torch.int64)
spans = [shard_range(n, r, world) for r in range(world)]
g = gather_verify_counts(block, lambda r: spans[r][0])
assert (g.mismatches, g.checked, g.first_bad) == (len(bad), n, 17), g
clean = gather_verify_counts(torch.tensor([0, -1, hi - lo], dtype=torch.int64), lambda r: spans[r][0])
assert (clean.mismatches, clean.checked, clean.first_bad) == (0, n, -1), clean
print("rank", rank, "ok", t)
dist.destroy_process_group()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 8, 1003, 131072):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_gloo_allreduce_of_verify_counters():
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_sixty_four_gib_corpus_shards_own_whole_gibs():
    """bench.py --gpus N: one 64 GiB corpus, rank r owns shard_range(8 388 608, r, N); at N = 2, 4, 8 every GiB
    (the unit the trailer folds of tests/golden/corpus64_folds.json are kept in) belongs whole to one rank, and
    the seeded corruptions are counted exactly once."""
    total, per_gib = 64 * 131072, 131072
    bad = np.sort(np.random.default_rng(1000).choice(total, 5 * 64, replace=False))
    for world in (2, 4, 8):
        owned, found = [], 0
        for r in range(world):
            lo, hi = shard_range(total, r, world)
            assert lo % per_gib == 0 and hi % per_gib == 0 and (hi - lo) * 8192 == (64 // world) << 30
            owned += list(range(lo // per_gib, hi // per_gib))
            found += int(((bad >= lo) & (bad < hi)).sum())
        assert owned == list(range(64)) and found == len(bad)
