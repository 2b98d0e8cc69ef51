import bz2, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import corpus
from decompress_amd import _lib
eng = _lib.Engine(0)
for n in (1, 10, 100, 1000, 5000, 50000, 120000, 400000):
    d = corpus.text(n, seed=n)
    z = bz2.compress(d, 1)
    got, res, info = eng.decode_stream(3, z, True)
    ok = got == d
    print(n, res.status, len(got), ok)
    if not ok and res.status == 0:
        for i, (a, b) in enumerate(zip(got, d)):
            if a != b:
                print("  first mismatch at", i); break
