"""Descriptor matching.  The reference ships no matcher, so the semantics are the
library's own (include/pislam_hip.h); the oracle's orc_match_hamming defines them and is itself checked
here against an independent numpy formulation."""
import numpy as np
import pytest


def numpy_match(q, t):
    """Independent formulation: full distance matrix by bit unpacking, stable argsort."""
    nq, nt = len(q), len(t)
    idx = np.full(nq, -1, np.int32)
    dist = np.full(nq, 0xFFFFFFFF, np.uint32)
    dist2 = np.full(nq, 0xFFFFFFFF, np.uint32)
    if nq == 0 or nt == 0:
        return idx, dist, dist2
    qb = np.unpackbits(q.view(np.uint8), axis=1).astype(np.int32)
    tb = np.unpackbits(t.view(np.uint8), axis=1).astype(np.int32)
    d = qb.sum(1)[:, None] + tb.sum(1)[None, :] - 2 * qb @ tb.T
    order = np.argsort(d, axis=1, kind="stable")
    idx[:] = order[:, 0]
    dist[:] = d[np.arange(nq), order[:, 0]]
    if nt > 1:
        dist2[:] = d[np.arange(nq), order[:, 1]]
    return idx, dist, dist2


def make_sets(rng, nq, nt, words, dup=True):
    t = rng.integers(0, 2**32, size=(nt, words), dtype=np.uint64).astype(np.uint32)
    q = rng.integers(0, 2**32, size=(nq, words), dtype=np.uint64).astype(np.uint32)
    if dup and nt >= 4 and nq >= 4:
        t[nt // 2] = t[1]                     # duplicate train descriptors: ties -> smallest index
        q[0] = t[1]                           # exact match with a tie
        q[1] = t[nt - 1] ^ np.uint32(1)       # distance 1
        q[2] = ~t[0]                          # maximal distance to t[0]
    return q, t


@pytest.mark.parametrize("words", [1, 2, 4, 8])
@pytest.mark.parametrize("nq,nt", [(0, 5), (5, 0), (1, 1), (7, 2), (300, 257), (64, 1000)])
def test_oracle_matcher_vs_numpy(orc, words, nq, nt):
    rng = np.random.default_rng(words * 1000 + nq * 7 + nt)
    q, t = make_sets(rng, nq, nt, words)
    got = orc.match_hamming(q.reshape(nq, words), t.reshape(nt, words))
    exp = numpy_match(q.reshape(nq, words), t.reshape(nt, words))
    for g, e in zip(got, exp):
        assert (g == e).all()


@pytest.fixture(params=[1, 0], ids=["mfma", "valu"])
def match_kernel(request, gpu_ctx):
    """Both matcher kernels: the matrix-core one (default) and the VALU popcount one."""
    gpu_ctx.set_option("match_mfma", request.param)
    yield request.param
    gpu_ctx.set_option("match_mfma", 1)


@pytest.mark.gpu
@pytest.mark.parametrize("words", [1, 2, 4, 8])
@pytest.mark.parametrize("nq,nt", [(0, 5), (5, 0), (1, 1), (7, 2), (31, 32), (33, 33), (129, 64), (300, 257), (64, 1000),
                                   (1025, 3000)])
def test_gpu_matcher_vs_oracle(gpu_ctx, orc, match_kernel, words, nq, nt):
    from pislam_amd import frontend
    rng = np.random.default_rng(words * 1000 + nq * 7 + nt + 1)
    q, t = make_sets(rng, nq, nt, words)
    got = frontend.matchHamming(q.reshape(nq, words), t.reshape(nt, words), ctx=gpu_ctx)
    exp = orc.match_hamming(q.reshape(nq, words), t.reshape(nt, words))
    for g, e in zip(got, exp):
        assert (g == e).all()


@pytest.mark.gpu
def test_gpu_matcher_structured_bits(gpu_ctx, orc, match_kernel):
    """Descriptors with one bit set / cleared per position (every bit position of every word reaches its own
    byte lane of the matrix-core fragments), sparse and dense sets, many exact ties."""
    from pislam_amd import frontend
    for words in (1, 2, 4, 8):
        K = 32 * words
        one = np.zeros((K, words), np.uint32)
        for k in range(K):
            one[k, k // 32] = np.uint32(1) << np.uint32(k % 32)
        sets = [one, ~one, np.concatenate([one, ~one, one[::-1]]), np.zeros((40, words), np.uint32),
                np.full((70, words), 0xFFFFFFFF, np.uint32)]
        for q in sets:
            for t in sets:
                got = frontend.matchHamming(q, t, ctx=gpu_ctx)
                exp = orc.match_hamming(q, t)
                for g, e in zip(got, exp):
                    assert (g == e).all(), (words, len(q), len(t))


@pytest.mark.gpu
def test_gpu_matcher_rejects_bad_arguments(gpu_ctx):
    from pislam_amd.capi import PislamError
    from pislam_amd import frontend
    q = np.zeros((4, 3), np.uint32)
    with pytest.raises(PislamError):
        frontend.matchHamming(q, q, ctx=gpu_ctx)                      # words = 3
    big = np.zeros((65536, 1), np.uint32)
    with pytest.raises(PislamError):
        frontend.matchHamming(np.zeros((4, 1), np.uint32), big, ctx=gpu_ctx)


@pytest.mark.gpu
def test_gpu_batch_matcher_on_frontend_outputs(gpu_ctx, orc, match_kernel):
    """Consecutive synthetic frames through the front-end, then matched pairwise on the device with the
    front-end's own [batch][max_kp][words] / counts arrays (ragged, one pair with an empty side)."""
    import torch
    from pislam_amd import synth
    from pislam_amd.frontend import OrbFrontend, matchHammingBatch
    levels = synth.level_table()
    B = 5
    pyr = synth.make_batch(40, B + 1)
    dev = torch.device("cuda:0")
    fe = OrbFrontend(levels, vstep=640, rows=2210, max_keypoints=2048, ctx=gpu_ctx)
    kp, desc, counts = fe.alloc_outputs(B + 1, dev)
    fe(torch.from_numpy(pyr).to(dev), kp, desc, counts)
    torch.cuda.synchronize()
    qd, td = desc[:B].contiguous(), desc[1:].contiguous()
    qc, tc = counts[:B].clone(), counts[1:].clone()
    tc[2] = 0                                   # a pair without train descriptors
    qc[3] = 3000                                # un-clamped count (> max_keypoints) is clamped to the stride
    idx, dist, dist2 = matchHammingBatch(qd, qc, td, tc, ctx=gpu_ctx)
    torch.cuda.synchronize()
    hq, ht = qd.cpu().numpy().view(np.uint32), td.cpu().numpy().view(np.uint32)
    hqc, htc = qc.cpu().numpy().view(np.uint32), tc.cpu().numpy().view(np.uint32)
    hi, hd, h2 = idx.cpu().numpy(), dist.cpu().numpy().view(np.uint32), dist2.cpu().numpy().view(np.uint32)
    for b in range(B):
        nq, nt = min(int(hqc[b]), 2048), min(int(htc[b]), 2048)
        ei, ed, e2 = orc.match_hamming(hq[b, :nq], ht[b, :nt].reshape(nt, 8))
        assert (hi[b, :nq] == ei).all() and (hd[b, :nq] == ed).all() and (h2[b, :nq] == e2).all()
    # size-independent property: matching a set against itself finds every descriptor at distance 0,
    # at the FIRST index holding an identical descriptor
    sidx, sdist, _ = matchHammingBatch(qd, counts[:B].contiguous(), qd, counts[:B].contiguous(), ctx=gpu_ctx)
    torch.cuda.synchronize()
    si, sd = sidx.cpu().numpy(), sdist.cpu().numpy()
    hc = counts[:B].cpu().numpy().view(np.uint32)
    for b in range(B):
        n = min(int(hc[b]), 2048)
        assert (sd[b, :n] == 0).all()
        assert (si[b, :n] <= np.arange(n)).all()
        assert (hq[b, si[b, :n]] == hq[b, :n]).all()
