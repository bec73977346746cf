"""CPU test of the oracle's Matcher::searchByBow restatement (oracle/oracle_match.cpp, reference matcher.cpp:619-721)
against an independent numpy / dict walk of the same reference lines."""
import numpy as np

import oracle


def _ref_walk(k1, d1, fv1, k2, d2, fv2, th_low, nratio, histo_len, check):
    bits = np.unpackbits
    matches, hist = [], [[] for _ in range(histo_len)]
    for node in sorted(set(fv1) & set(fv2)):
        for idx1 in fv1[node]:
            best1, best2, bidx = 256, 256, -1
            for idx2 in fv2[node]:
                dist = int(bits(d1[idx1] ^ d2[idx2]).sum())
                if dist < best1:
                    best2, best1, bidx = best1, dist, idx2
                elif dist < best2:
                    best2 = dist
            if best1 < th_low and bidx >= 0 and np.float32(best1) < np.float32(nratio) * np.float32(best2):
                matches.append((idx1, bidx, -1, float(best1)))
                if check:
                    rot = np.float32(k1["angle"][idx1]) - np.float32(k2["angle"][bidx])
                    if rot < 0:
                        rot = np.float32(rot + np.float32(360.0))
                    v = np.float32(rot * np.float32(np.float32(1.0) / np.float32(histo_len)))
                    b = int(np.floor(v + 0.5)) if v >= 0 else int(np.ceil(v - 0.5))
                    if b == histo_len:
                        b = 0
                    hist[b].append(len(matches) - 1)
    if not check:
        return matches
    i1, i2, i3 = oracle.three_maxima([len(h) for h in hist])
    return [matches[i] for b in range(histo_len) if b in (i1, i2, i3) for i in hist[b]]


def test_search_by_bow_oracle_vs_independent_walk():
    rng = np.random.default_rng(11)
    n1, n2 = 400, 380
    k1 = np.zeros(n1, oracle.KEYPOINT); k2 = np.zeros(n2, oracle.KEYPOINT)
    k1["angle"] = rng.uniform(0, 360, n1).astype(np.float32); k2["angle"] = rng.uniform(0, 360, n2).astype(np.float32)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    d2 = d1[rng.integers(0, n1, n2)].copy()
    d2[:, :3] ^= rng.integers(0, 256, (n2, 3), dtype=np.uint8)
    fv1, fv2 = {}, {}
    for i in range(n1):
        fv1.setdefault(int(d1[i, 5] % 37), []).append(i)      # node = a hash of the descriptor: matching pairs share it
    for i in rng.permutation(n2):
        fv2.setdefault(int(d2[i, 5] % 37) + (40 if i % 11 == 0 else 0), []).append(int(i))
    for th, ratio, hl, chk in ((50, 0.9, 30, True), (120, 0.8, 30, False), (256, 1.5, 45, True)):
        got = oracle.search_by_bow(k1, d1, fv1, k2, d2, fv2, th_low=th, nratio=ratio, histo_len=hl, check_orientation=chk)
        exp = _ref_walk(k1, d1, fv1, k2, d2, fv2, th, ratio, hl, chk)
        assert len(exp) > 20
        assert [tuple(m) for m in got.tolist()] == [(a, b, c, np.float32(d)) for a, b, c, d in exp]
