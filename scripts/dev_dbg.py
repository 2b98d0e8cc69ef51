import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from decompress_amd import _lib
eng = _lib.Engine(0)
z = bytes([120, 156, 99, 84, 100, 7, 0, 0, 79, 0, 42])
g = bytes([31, 139, 8, 8, 0, 0, 0, 0, 0, 0, 102, 111, 111, 0, 99, 84, 100, 7, 0, 101, 51, 120, 236, 3, 0, 0, 0])
blobs = [(1, z), (2, g), (0, z[2:7])]
data = b"".join(b for _, b in blobs)
units = []; off = 0
for i, (k, b) in enumerate(blobs):
    units.append(_lib.Unit(in_off=off, in_limit=off + len(b), out_off=32 * i, out_cap=16, kind=k)); off += len(b)
out, res = eng.decode_batch(data, units, 128)
for r in res:
    print(r.status, r.detail, r.in_end_bit, r.out_len, bytes(r.check[:8]).hex(), r.check_len, r.aux)
print(out.raw[:96])
