"""CPU: the Philox4x32-10 restatement (oracle/philox_oracle.py) against the published Random123 known-answer vectors, and
the properties of the stream layout the HIP kernel is compared with on the GPU (tests/test_hip_philox.py)."""
import json
import os

import numpy as np

from oracle import philox_oracle as po

KAT = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "philox_kat.json")))


def _hex(words):
    return [int(w, 16) for w in words]


def test_restatement_matches_random123_known_answers():
    assert KAT["generator"] == "philox4x32" and KAT["rounds"] == 10 and len(KAT["vectors"]) == 3
    for v in KAT["vectors"]:
        out = po.philox4x32_10(_hex(v["counter"]), _hex(v["key"]))
        assert [int(x) for x in out] == _hex(v["expected"])
    # vectorised form == one block at a time
    ctr = np.array([_hex(v["counter"]) for v in KAT["vectors"]], dtype=np.uint64)
    key = np.array([_hex(v["key"]) for v in KAT["vectors"]], dtype=np.uint64)
    exp = np.array([_hex(v["expected"]) for v in KAT["vectors"]], dtype=np.uint32)
    assert np.array_equal(po.philox4x32_10(ctr, key), exp)


def test_stream_layout_and_box_muller():
    seed = 0x0123456789ABCDEF
    u = po.stream_uint32(seed, 3, 64)
    # group g of stream s = block (g, 0, s, tag) under key (seed lo, seed hi)
    blk = po.philox4x32_10([5, 0, 3, po.STREAM_TAG], [seed & 0xFFFFFFFF, seed >> 32])
    assert np.array_equal(u[20:24], blk)
    assert not np.array_equal(u, po.stream_uint32(seed, 4, 64)) and not np.array_equal(u, po.stream_uint32(seed + 1, 3, 64))
    # Box-Muller: the all-ones word gives u1 = 1 -> r = 0; the zero word gives u1 = 2^-24 -> r = sqrt(48 ln 2) at angle 0
    z = po.normals_from_uint32(np.array([0xFFFFFFFF, 0, 0, 0], dtype=np.uint32))
    assert z[0] == 0.0 and z[1] == 0.0
    assert abs(z[2] - np.sqrt(48 * np.log(2.0))) < 1e-12 and z[3] == 0.0
    x = po.normal_stream(seed, 7, 1 << 18)
    assert abs(x.mean()) < 6e-3 and abs(x.var() - 1) < 1e-2 and abs((x ** 4).mean() - 3) < 8e-2
