"""Synthetic corpora generated directly in device memory (benchmarks and tests)."""


def splitmix_dev(nbytes, state=301):
    """The splitmix64 byte stream the benchmark configurations are defined on: state `state`, the golden
    gamma 0x9e3779b97f4a7c15 added before each 8-byte little-endian word, mixers 0xbf58476d1ce4e5b9 and
    0x94d049bb133111eb with shifts 30 / 27 / 31.  Returns a uint8 tensor on the current GPU."""
    import torch
    assert nbytes % 8 == 0
    n = nbytes // 8
    idx = torch.arange(1, n + 1, dtype=torch.int64, device="cuda")
    golden = -7046029254386353131  # 0x9e3779b97f4a7c15 as int64
    z = idx * golden + state

    def lsr(x, k):  # logical shift right on int64
        return (x >> k) & ((1 << (64 - k)) - 1)

    z = (z ^ lsr(z, 30)) * -4658895280553007687  # 0xbf58476d1ce4e5b9
    z = (z ^ lsr(z, 27)) * -7723592293110705685  # 0x94d049bb133111eb
    z = z ^ lsr(z, 31)
    return z.view(torch.uint8)


def splitmix_fill(buf, state=301, first_word=0, chunk_bytes=256 << 20):
    """Fills buf (uint8 device tensor, a multiple of 8 bytes) with the words first_word, first_word + 1, ...
    of that stream, a chunk at a time so that temporaries stay small next to a shard of tens of GiB."""
    import torch
    golden = -7046029254386353131
    words = buf.view(torch.int64)
    step = chunk_bytes // 8

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    for w0 in range(0, words.numel(), step):
        n = min(step, words.numel() - w0)
        lo = first_word + w0 + 1
        # index * golden wraps modulo 2^64 exactly like the generator's running sum
        z = torch.arange(lo, lo + n, dtype=torch.int64, device=buf.device) * golden + state
        z = (z ^ lsr(z, 30)) * -4658895280553007687
        z = (z ^ lsr(z, 27)) * -7723592293110705685
        words[w0:w0 + n] = z ^ lsr(z, 31)
    return buf
