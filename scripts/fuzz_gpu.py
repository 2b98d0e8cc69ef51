#!/usr/bin/env python3
"""Developer check, run on the GPU box: many more damaged streams (deflate family, bzip2 blocks, raw LZMA2 units, xz blocks behind filters) than the test suite holds, each decoded by the
engine (through the C ABI, as the tests do) and by the oracle, compared on status, detail and bytes.  Seeds are
printed so that a failure can be replayed.  usage: python3 scripts/fuzz_gpu.py [cases] [seed]"""
import bz2
import os
import random
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus  # noqa: E402
import test_gpu_bzip2 as tb  # noqa: E402
import test_gpu_deflate as td  # noqa: E402
from decompress_amd import _lib  # noqa: E402
from oracle import oracle  # noqa: E402


def damage(blob, rng):
    kind = rng.randrange(4)
    if kind == 0 and len(blob) > 2:
        return blob[:rng.randrange(1, len(blob))]
    b = bytearray(blob)
    # (most of the damage near the front: headers, tables, selectors)
    hi = len(b) if kind == 1 else min(len(b), 200)
    for _ in range(rng.choice([1, 1, 2, 4])):
        b[rng.randrange(0, hi)] ^= 1 << rng.randrange(8)
    return bytes(b)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    eng = _lib.Engine(0)
    gens = [corpus.text, corpus.random_bytes, corpus.repetitive, corpus.mixed]
    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]
    seen = {}
    done = 0
    while done < cases:
        items = []
        for _ in range(64):
            n = rng.choice([30, 300, 3000, 9000, 40000, 120000])
            d = gens[rng.randrange(4)](n, seed=rng.randrange(1 << 30))
            fmt = rng.randrange(3)
            blob = td.compress(fmt, d, rng.choice([1, 4, 6, 9]), rng.choice(strategies))
            items.append((fmt, damage(blob, rng), len(d) + rng.choice([0, 8, 600])))
        td.check_items(eng, items, rng.randrange(1 << 30))
        for it in items[::8]:
            st = td.oracle_unit(it[0], it[1], it[2])[0].status
            seen[st] = seen.get(st, 0) + 1
        done += 64
        if done % 640 == 0:
            print(f"deflate family: {done} cases", flush=True)
    print("deflate family statuses (every eighth case):", dict(sorted(seen.items())), flush=True)
    # bzip2: one block per case
    seen = {}
    for t in range(cases // 16):
        n = rng.choice([2000, 30000, 120000, 400000])
        d = gens[rng.randrange(4)](n, seed=rng.randrange(1 << 30))
        level = rng.choice([1, 9])
        data = damage(bz2.compress(d, level), rng)
        if len(data) < 5:
            continue
        u = _lib.Unit(in_off=4, in_limit=len(data), out_off=_lib.OUT_AUTO, kind=_lib.UNIT_BZIP2_BLOCK, param0=level)
        out, res = eng.decode_batch(data, [u], 1 << 20)
        er, eo = oracle.bzip2_block(data, 32, level, cap=1 << 20)
        assert res[0].status == er.status, ("bzip2", seed, t, res[0].status, er.status)
        if er.status == 0:
            assert out.raw[:res[0].out_len] == eo, ("bzip2", seed, t)
        else:
            assert res[0].detail == er.detail, ("bzip2", seed, t, res[0].status, res[0].detail, er.detail)
        seen[er.status] = seen.get(er.status, 0) + 1
        if (t + 1) % 50 == 0:
            print(f"bzip2: {t + 1} cases", flush=True)
    print("bzip2 statuses:", dict(sorted(seen.items())), flush=True)
    # raw LZMA2 units, batches of 64, lc / lp / pb and dictionary sizes varied
    import lzma
    seen = {}
    for t in range(cases // 64):
        blobs, dicts, caps = [], [], []
        for _ in range(64):
            n = rng.choice([100, 3000, 40000, 150000])
            d = gens[rng.randrange(4)](n, seed=rng.randrange(1 << 30))
            lc, lp = rng.choice([(3, 0), (0, 0), (1, 2), (0, 4), (4, 0)])
            ds = rng.choice([4096, 1 << 16, 1 << 20])
            f = [{"id": lzma.FILTER_LZMA2, "preset": rng.choice([0, 2, 6]), "dict_size": ds, "lc": lc, "lp": lp, "pb": rng.choice([0, 2, 4])}]
            blob = lzma.compress(d, lzma.FORMAT_RAW, filters=f)
            blobs.append(damage(blob, rng) if rng.randrange(8) else blob)
            dicts.append(ds if rng.randrange(4) else 4096)
            caps.append(n + rng.choice([0, 16, 5000]))
        data = b"".join(blobs)
        units, off, ooff = [], 0, 0
        for bl, ds, cap in zip(blobs, dicts, caps):
            units.append(_lib.Unit(in_off=off, in_limit=off + len(bl), out_off=ooff, out_cap=cap, kind=_lib.UNIT_LZMA2, param0=ds))
            off += len(bl)
            ooff += cap + 8
        out, res = eng.decode_batch(data, units, ooff)
        for i, (u, r, bl) in enumerate(zip(units, res, blobs)):
            er, eo = oracle.lzma2(bl, u.param0, cap=u.out_cap)
            assert r.status == er.status, ("lzma2", seed, t, i, r.status, er.status, r.detail, er.detail)
            if er.status == 0:
                assert out.raw[u.out_off:u.out_off + r.out_len] == eo, ("lzma2", seed, t, i)
                assert r.in_end_bit - 8 * u.in_off == er.in_end_bit, ("lzma2", seed, t, i)
            else:
                assert r.detail == er.detail, ("lzma2", seed, t, i, r.status, r.detail, er.detail)
            seen[er.status] = seen.get(er.status, 0) + 1
        if (t + 1) % 10 == 0:
            print(f"lzma2: {64 * (t + 1)} cases", flush=True)
    print("lzma2 statuses:", dict(sorted(seen.items())), flush=True)
    # xz blocks behind filter chains (x86 BCJ, delta), damaged in the payload, padding or check
    import xzutil
    from test_split import split
    seen = {}
    for t in range(cases // 16):
        n = rng.choice([700, 5000, 70000, 300000])
        d = rng.choice([corpus.x86ish, corpus.mixed, corpus.text])(n, seed=rng.randrange(1 << 30))
        filters = rng.choice([[{"id": lzma.FILTER_X86}, {"id": lzma.FILTER_LZMA2, "preset": 3}],
                              [{"id": lzma.FILTER_DELTA, "dist": rng.choice([1, 4, 256])}, {"id": lzma.FILTER_LZMA2, "preset": 1}],
                              [{"id": lzma.FILTER_X86}, {"id": lzma.FILTER_DELTA, "dist": 3}, {"id": lzma.FILTER_LZMA2, "preset": 2}],
                              [{"id": lzma.FILTER_LZMA2, "preset": 6}]])
        base = xzutil.xz_py(d, rng.choice([lzma.CHECK_NONE, lzma.CHECK_CRC32, lzma.CHECK_CRC64, lzma.CHECK_SHA256]), filters)
        units0, _ = split(_lib.FMT_XZ, base)
        if len(units0) != 1:
            continue
        b = bytearray(base)
        if rng.randrange(6):
            for _ in range(rng.choice([1, 2])):
                b[rng.randrange(units0[0].in_off, len(b) - 24)] ^= 1 << rng.randrange(8)
        data = bytes(b)
        u = units0[0]
        out, res = eng.decode_batch(data, [u], u.out_cap + 16)
        er, eo = oracle.decompress(8, data, True, cap=len(d) + 16)
        assert res[0].status == er.status, ("xz", seed, t, res[0].status, er.status, res[0].detail, er.detail)
        if er.status == 0:
            assert out.raw[:res[0].out_len] == eo, ("xz", seed, t)
        else:
            assert res[0].detail == er.detail, ("xz", seed, t, res[0].status, res[0].detail, er.detail)
        seen[er.status] = seen.get(er.status, 0) + 1
        if (t + 1) % 100 == 0:
            print(f"xz: {t + 1} cases", flush=True)
    print("xz statuses:", dict(sorted(seen.items())), flush=True)
    eng.close()
    print("ok")


if __name__ == "__main__":
    main()
