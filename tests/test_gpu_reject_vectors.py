"""GPU parity on the hand-built vectors of tests/vectors_reject.py: the HIP path gives every vector the status
AND the detail the oracle gives (and, for accepted streams, the same bytes), through the C ABI.  The set of
statuses the kernels returned must cover the enum of include/semz_status.h minus the codes that no input can
provoke (vectors_reject.UNREACHABLE) and the engine-level ones that have their own tests."""
import os
import re

import pytest

import reject_runner
import vectors_reject as V

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEEN = set()


@pytest.mark.parametrize("i", range(len(V.VECTORS)), ids=lambda i: V.VECTORS[i]["name"][:60])
def test_gpu_matches_oracle(engine, i):
    v = V.VECTORS[i]
    want = reject_runner.oracle_run(v)
    got = reject_runner.gpu_run(engine, v)
    assert got[0] == want[0] == v["status"], (v["name"], got[0], want[0])
    if got[0] != 0:
        assert got[1] == want[1], (v["name"], "detail", got[1], want[1])
        assert got[2][:len(want[2])] == want[2][:len(got[2])]
    else:
        assert got[2] == want[2]
        if v["out"] is not None:
            assert got[2] == v["out"]
    SEEN.add(got[0])


def test_statuses_seen_cover_the_enum():
    text = open(os.path.join(ROOT, "include", "semz_status.h")).read()
    enum = {int(n) for n in re.findall(r"SEMZ_\w+ = (\d+)", text)}
    missing = enum - SEEN - set(V.UNREACHABLE) - set(V.ENGINE_LEVEL)
    assert not missing, sorted(missing)
