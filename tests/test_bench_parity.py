"""CPU: bench.py's parity_in_run checker itself — it must accept the oracle's own results laid out like the GPU outputs
(padded to max_keypoints, counts un-clamped, a clipped capacity included) and reject a wrong count, one wrong keypoint
word, one wrong descriptor word, a wrong keypoint ORDER and a count that misses the reference's pin."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _snap(results, max_kp, words=8):
    n = len(results)
    kp = np.zeros((n, max_kp), np.uint32)
    desc = np.zeros((n, max_kp, words), np.uint32)
    counts = np.zeros(n, np.uint32)
    for i, (k, d) in enumerate(results):
        m = min(len(k), max_kp)
        kp[i, :m], desc[i, :m], counts[i] = k[:m], d[:m], len(k)
    return kp, desc, counts


def test_parity_in_run_accepts_the_oracle_and_rejects_every_kind_of_mismatch(orc):
    import bench
    from pislam_amd import synth
    levels = synth.level_table(320, 240, 4)
    host = synth.make_batch(7700, 3, w0=320, h0=240, nlevels=4, levels=levels, nshapes=30)
    res = [orc.pyramid4(p, levels, cap=bench.CPU_CAP)[:2] for p in host]
    assert min(len(k) for k, _ in res) > 20
    for max_kp in (2048, 64):                               # (64: fewer slots than keypoints — only the stored ones are compared)
        lanes = [_snap(res, max_kp), _snap(res[:2], max_kp)]
        out = bench.parity_in_run(lanes, res, [host, host], levels, max_kp, 0, 5)
        assert out["ok"] and out["pyramids"] == 3 and out["pyramids_other_lanes"] == 2, out
        assert out["keypoints"] == sum(min(len(k), max_kp) for k, _ in res) + sum(min(len(k), max_kp) for k, _ in res[:2])

    def broken(mutate, lane=0):
        lanes = [_snap(res, 2048), _snap(res[:2], 2048)]
        mutate(lanes[lane])
        out = bench.parity_in_run(lanes, res, [host, host], levels, 2048, 0, 5)
        assert not out["ok"] and out.get("first_mismatch"), out
        return out["first_mismatch"]

    assert "keypoints, oracle" in broken(lambda s: s[2].__setitem__(1, s[2][1] + 1))
    assert "keypoint words" in broken(lambda s: s[0].__setitem__((2, 5), s[0][2, 5] ^ 1))
    assert "descriptor words" in broken(lambda s: s[1].__setitem__((0, 3, 7), s[1][0, 3, 7] ^ 0x80000000))

    def swap(s):                                             # right set of keypoints, wrong order
        s[0][0, [0, 1]] = s[0][0, [1, 0]]
        s[1][0, [0, 1]] = s[1][0, [1, 0]]
    assert "keypoint words" in broken(swap)
    assert "lane 1" in broken(lambda s: s[0].__setitem__((1, 0), 0), lane=1)                # the other lanes are checked too
    out = bench.parity_in_run([_snap(res, 2048)], res, [host], levels, 2048, 0, 5, pins=len(res[0][0]) + 1)
    assert not out["ok"] and "pin" in out["first_mismatch"]
    assert not bench.parity_in_run([_snap(res, 2048)], [], [host], levels, 2048, 0, 5)["ok"]   # nothing compared is not a pass
