"""Writes tests/golden/corpus64_folds.json: for every GiB g of the 64 GiB verification corpus (the splitmix64
stream with state 301 cut into 8 KiB blocks, SST trailers with type byte 0) the XOR of its 131 072 masked
CRCs and a few individual trailers, computed by the CPU oracle (SSE4.2 path, checked block by block against its
table path on the first GiB).  GiB 0 must reproduce the value captured from a build of the reference.

    python tools/make_corpus_folds.py          (about two minutes on 8 cores)
"""
import json, os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O

GOLDEN = 0x9E3779B97F4A7C15
BLEN, PER_GIB = 8192, 131072
WORDS_PER_GIB = (1 << 30) // 8
NTHREADS = 8


def gib_trailers(g):
    state = (301 + GOLDEN * WORDS_PER_GIB * g) & (2**64 - 1)
    buf = O.splitmix(1 << 30, state)
    out = np.empty(PER_GIB, np.uint32)
    per = PER_GIB // NTHREADS

    def work(t):
        out[t * per:(t + 1) * per] = O.sst_trailer_batch(buf[t * per * BLEN:(t + 1) * per * BLEN], BLEN, BLEN, per, 0, hw=True)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(NTHREADS)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    if g == 0:
        assert (O.sst_trailer_batch(buf[:4096 * BLEN], BLEN, BLEN, 4096, 0, hw=False) == out[:4096]).all()
    return out


def main():
    rng = np.random.default_rng(64)
    folds, samples = [], {}
    for g in range(64):
        tr = gib_trailers(g)
        folds.append(f"{int(np.bitwise_xor.reduce(tr)):08x}")
        for k in sorted(set([0, PER_GIB - 1] + rng.integers(0, PER_GIB, 6).tolist())):
            samples[str(g * PER_GIB + k)] = f"{int(tr[k]):08x}"
        print(g, folds[-1], flush=True)
    assert folds[0] == "0744a84e", folds[0]
    doc = {"what": "64 GiB corpus: splitmix64 state 301, 8 KiB blocks, trailer = Mask(crc32c(block || 0x00))",
           "made_by": "tools/make_corpus_folds.py (CPU oracle)", "block_len": BLEN, "blocks_per_gib": PER_GIB,
           "xor_of_masked_per_gib": folds, "trailer_of_block": samples}
    json.dump(doc, open(os.path.join(ROOT, "tests", "golden", "corpus64_folds.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
