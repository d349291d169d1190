import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# SHA-256 prefixes of the reference's outputs on demo/input.png, recorded in SURVEY.md §8c
SURVEY_PINS = {
    "img": "0f7c28c31d3466be", "det": "205ff4f820fd568c", "score": "7f0f15bfc5110f4c",
    "kp": "03f731507fe64358", "kp_bucket43": "feb6071b7035e57d", "centroids": "6e06b93c6186852c",
    "angles": "5c511df056e40160", "desc": "86199421e6b3a298",
}
DEMO_LEVELS = [(640, 480, 0), (533, 400, 480), (444, 333, 880), (370, 278, 1213), (309, 231, 1491),
               (257, 193, 1722), (214, 161, 1915), (179, 134, 2076)]     # demo/demo.cpp:38-47


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def sha16(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


@pytest.fixture(scope="session")
def demo():
    """The reference's demo pyramid + its golden outputs (tests/golden/make_golden.py)."""
    z = np.load(os.path.join(GOLDEN, "demo_pyramid.npz"))
    d = {k: z[k] for k in z.files}
    img = d["img"]
    det = np.zeros(img.size, np.uint8)
    det[d["det_nonzero"]] = 0xFF
    score = np.zeros(img.size, np.uint8)
    score[d["score_idx"]] = d["score_val"]
    d["det"] = det.reshape(img.shape)
    d["score"] = score.reshape(img.shape)
    d["levels"] = DEMO_LEVELS
    return d


@pytest.fixture(scope="session")
def synth_small():
    z = np.load(os.path.join(GOLDEN, "synth_small.npz"))
    d = {k: z[k] for k in z.files}
    d["levels"] = [tuple(int(v) for v in r) for r in d["levels"]]
    return d


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pislam_amd import frontend
    return frontend.default_context()
