"""Known answers for the oracle's bag-of-words restatement (DBoW2's transform on the reference's
vocabulary text format), worked by hand on a two-level tree and cross-checked against a direct
numpy restatement on random trees."""
import numpy as np

from tests import voc_util


def _desc(bits):
    d = np.zeros(32, np.uint8)
    for b in bits:
        d[b >> 3] |= 1 << (b & 7)
    return d


def test_hand_built_tree(oracle, tmp_path):
    # root -> A (bits 0..7 set), B (no bits); A -> A0 (bits 0..7), A1 (bits 0..15); B -> B0 (no bits), B1 (bit 200)
    A, B = _desc(range(8)), _desc([])
    nodes = [(0, 0, A, 0.0), (0, 0, B, 0.0),
             (1, 1, _desc(range(8)), 2.0), (1, 1, _desc(range(16)), 3.0),
             (2, 1, _desc([]), 0.0), (2, 1, _desc([200]), 5.0)]
    arrays = (np.array([n[0] for n in nodes], np.int32), np.array([n[1] for n in nodes], np.uint8),
              np.stack([n[2] for n in nodes]), np.array([n[3] for n in nodes]))
    path = tmp_path / "voc.txt"
    voc_util.write_text(path, 2, 2, arrays)
    voc = oracle.Vocabulary(path=path)
    assert voc.info() == (2, 2, 7, 4)
    feats = np.stack([_desc(range(8)),          # -> A -> A0 (word 0, w 2)
                      _desc(range(14)),         # -> A (6 vs 14) -> A1 (2 vs 6)  (word 1, w 3)
                      _desc([200, 201]),        # -> B (2 vs 10) -> B1 (1 vs 2)  (word 3, w 5)
                      _desc([]),                # -> B -> B0: word 2 has weight 0: dropped
                      _desc(range(4)),          # tie at the root (4 vs 4): first child A; then A0 (4 vs 12)
                      _desc(range(8))])         # word 0 again
    wi, wv, ni, ns, ft = voc.transform(feats, levelsup=1)   # node ids at level 1
    assert wi.tolist() == [0, 1, 3]
    raw = np.array([2.0 + 2.0 + 2.0, 3.0, 5.0])
    assert np.array_equal(wv, raw / raw.sum())
    assert ni.tolist() == [1, 2] and ns.tolist() == [0, 4, 5] and ft.tolist() == [0, 1, 4, 5, 2]
    # levelsup >= L: every feature is filed under the root
    wi2, wv2, ni2, ns2, ft2 = voc.transform(feats, levelsup=4)
    assert ni2.tolist() == [0] and ft2.tolist() == [0, 1, 2, 4, 5] and np.array_equal(wv2, wv)
    # a file without the final newline is the same vocabulary
    voc_util.write_text(path, 2, 2, arrays, trailing_newline=False)
    assert oracle.Vocabulary(path=path).info() == (2, 2, 7, 4)


def _numpy_transform(k, L, arrays, feats, levelsup, weighting=0):
    parent, leaf, desc, weight = arrays
    n = len(parent)
    children = [[] for _ in range(n + 1)]
    word = np.zeros(n + 1, np.int64)
    w = 0
    for i in range(n):
        children[parent[i]].append(i + 1)
        if leaf[i]:
            word[i + 1] = w
            w += 1
    lut = np.array([bin(i).count("1") for i in range(256)])
    bow, fv = {}, {}
    for fi, f in enumerate(feats):
        node, level, nid = 0, 0, 0
        while children[node]:
            level += 1
            ch = children[node]
            d = [int(lut[np.bitwise_xor(f, desc[c - 1])].sum()) for c in ch]
            node = ch[int(np.argmin(d))]
            if level == L - levelsup:
                nid = node
        wt = weight[node - 1]          # the stored weight, whatever the weighting mode
        if wt > 0:
            bow[word[node]] = bow.get(word[node], 0.0) + wt if word[node] in bow else wt
            fv.setdefault(nid, []).append(fi)
    keys = sorted(bow)
    norm = 0.0
    for kk in keys:
        norm += abs(bow[kk])
    vals = [bow[kk] / norm for kk in keys] if norm > 0 else [bow[kk] for kk in keys]
    return keys, vals, {kk: fv[kk] for kk in sorted(fv)}


def test_random_trees_against_numpy(oracle):
    rng = np.random.default_rng(2)
    for k, L, ragged, levelsup, weighting in [(10, 3, False, 1, 0), (6, 4, True, 2, 0), (3, 5, True, 4, 0), (10, 2, False, 4, 1)]:
        arrays = voc_util.make_vocabulary(k, L, seed=k * 10 + L, ragged=ragged)
        voc = oracle.Vocabulary(arrays=(k, L) + arrays + (weighting,))
        # features: perturbed node descriptors (meaningful descents) and pure noise
        pick = rng.integers(0, len(arrays[0]), 300)
        feats = arrays[2][pick].copy()
        feats[:, rng.integers(0, 32)] ^= rng.integers(0, 256, 300).astype(np.uint8)
        feats = np.concatenate([feats, rng.integers(0, 256, (100, 32), dtype=np.uint8)])
        wi, wv, ni, ns, ft = voc.transform(feats, levelsup)
        keys, vals, fv = _numpy_transform(k, L, arrays, feats, levelsup, weighting)
        assert wi.tolist() == keys and np.array_equal(wv, np.array(vals))
        assert ni.tolist() == list(fv)
        for r, kk in enumerate(fv):
            assert ft[ns[r]:ns[r + 1]].tolist() == fv[kk]
        assert abs(wv.sum() - 1.0) < 1e-12
