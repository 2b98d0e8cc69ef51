"""Deterministic synthetic corpora: seeded Zipf word text,
incompressible bytes, highly repetitive bytes, x86-like call sequences."""
import numpy as np


def _splitmix(seed):
    return np.random.Generator(np.random.PCG64(seed))


def text(nbytes, seed=0x5EED0001, vocab=4096):
    """Zipf-distributed words from a seeded vocabulary; about one space in twelve is a
    newline (lines of roughly 40-100 characters).  Vectorised: ~100 MB/s."""
    rng = _splitmix(seed)
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    W = 12
    lens = rng.integers(2, 11, size=vocab)
    table = np.zeros((vocab, W), dtype=np.uint8)
    for i, l in enumerate(lens):
        # letter frequencies fall off with rank, like natural text
        table[i, :l] = letters[np.minimum(rng.geometric(0.12, size=int(l)) - 1, 25)]
        table[i, l] = 0x20
    ranks = np.arange(1, vocab + 1, dtype=np.float64)
    p = 1.0 / ranks
    p /= p.sum()
    cdf = np.cumsum(p)
    nwords = nbytes // 5 + 16
    idx = np.searchsorted(cdf, rng.random(nwords)).clip(0, vocab - 1)
    mat = table[idx]
    nl = rng.random(nwords) < (1.0 / 12.0)
    sp = lens[idx]
    mat[np.nonzero(nl)[0], sp[nl]] = 0x0A
    flat = mat.reshape(-1)
    flat = flat[flat != 0]
    while flat.size < nbytes:
        flat = np.concatenate([flat, flat])
    return flat[:nbytes].tobytes()


def random_bytes(nbytes, seed=1):
    return _splitmix(seed).integers(0, 256, size=nbytes, dtype=np.uint8).tobytes()


def repetitive(nbytes, seed=2):
    rng = _splitmix(seed)
    out = bytearray()
    while len(out) < nbytes:
        run = int(rng.integers(1, 2000))
        if rng.random() < 0.5:
            out += bytes([int(rng.integers(0, 256))]) * run
        else:
            pat = bytes(rng.integers(0, 256, size=int(rng.integers(1, 9)), dtype=np.uint8))
            out += pat * (run // len(pat) + 1)
    return bytes(out[:nbytes])


def x86ish(nbytes, seed=3):
    """Text interleaved with E8/E9 rel32 sequences so that the BCJ filter has work."""
    rng = _splitmix(seed)
    base = bytearray(text(nbytes, seed=seed + 100))
    pos = 0
    while pos + 8 < nbytes:
        pos += int(rng.integers(8, 200))
        if pos + 5 >= nbytes:
            break
        base[pos] = 0xE8 if rng.random() < 0.7 else 0xE9
        tgt = int(rng.integers(0, 1 << 20))
        rel = (tgt - pos) & 0xFFFFFFFF if rng.random() < 0.8 else int(rng.integers(0, 1 << 32))
        base[pos + 1:pos + 5] = rel.to_bytes(4, "little")
    return bytes(base)


def mixed(nbytes, seed=4):
    third = nbytes // 3
    return text(third, seed) + random_bytes(third, seed + 1) + repetitive(nbytes - 2 * third, seed + 2)
