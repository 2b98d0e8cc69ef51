import sys, os, lzma
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus, xzutil
from decompress_amd import _lib
from test_split import split
eng = _lib.Engine(0)
d = corpus.text(3000, seed=1)
data = xzutil.xz_py(d, lzma.CHECK_CRC32, [{"id": lzma.FILTER_LZMA2, "preset": 0}])
units, info = split(_lib.FMT_XZ, data)
u = units[0]
print("unit", u.in_off, u.in_limit, u.out_off, u.out_cap, u.kind, u.flags, u.param0, u.param1, u.filters, data[u.in_off:u.in_off+8].hex())
out, res = eng.decode_batch(data, units, u.out_cap + 16)
r = res[0]
print("res", r.status, r.detail, r.in_end_bit, r.out_len, r.aux, bytes(r.check[:8]).hex(), out.raw[:16])
raw = lzma.compress(d, lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "preset": 0, "dict_size": 1 << 16}])
u2 = _lib.Unit(in_off=0, in_limit=len(raw), out_off=0, out_cap=len(d), kind=_lib.UNIT_LZMA2, param0=1 << 16)
out, res = eng.decode_batch(raw, [u2], len(d) + 16)
r = res[0]
print("raw", raw[:8].hex(), r.status, r.detail, r.in_end_bit, r.out_len, out.raw[:16], d[:16])
