"""Synthetic vocabularies in the reference's text format (ORBVocabulary::saveAsText /
loadFromText, core/lib/ORB/vocabulary.cc:96-178): a k-ary tree of depth L whose children are
their parent's descriptor with random bits flipped, idf-like leaf weights, a few stopped words
(weight 0), and - when ragged - some nodes with fewer children and some leaves above depth L."""
import numpy as np


def make_vocabulary(k, L, seed=0, ragged=False, stop_fraction=0.05):
    """-> (parent int32[n], is_leaf uint8[n], desc uint8[n, 32], weight float64[n]) for nodes 1..n
    in file order (breadth first, the children of a node consecutive)."""
    rng = np.random.default_rng(seed)
    parent, leaf, desc, weight = [], [], [], []
    root_desc = rng.integers(0, 256, 32, dtype=np.uint8)
    frontier = [(0, root_desc, 0)]          # (node id, descriptor, depth)
    next_id = 1
    while frontier:
        new_frontier = []
        for nid, d, depth in frontier:
            nch = k if not ragged else int(rng.integers(max(1, k - 3), k + 1))
            for _ in range(nch):
                flips = rng.integers(0, 256, int(rng.integers(8, 40)))
                cd = d.copy()
                for b in flips:
                    cd[b >> 3] ^= np.uint8(1 << (b & 7))
                is_leaf = depth + 1 == L or (ragged and depth + 1 >= 2 and rng.random() < 0.05)
                parent.append(nid)
                leaf.append(1 if is_leaf else 0)
                desc.append(cd)
                if is_leaf:
                    weight.append(0.0 if rng.random() < stop_fraction else float(rng.uniform(0.1, 9.0)))
                else:
                    weight.append(0.0)
                    new_frontier.append((next_id, cd, depth + 1))
                next_id += 1
        frontier = new_frontier
    return (np.array(parent, np.int32), np.array(leaf, np.uint8), np.array(desc, np.uint8).reshape(-1, 32),
            np.array(weight, np.float64))


def write_text(path, k, L, arrays, scoring=0, weighting=0, trailing_newline=True):
    parent, leaf, desc, weight = arrays
    with open(path, "w") as f:
        f.write("%d %d %d %d\n" % (k, L, scoring, weighting))
        lines = []
        for i in range(len(parent)):
            lines.append("%d %d %s %r" % (parent[i], leaf[i], " ".join(str(int(b)) for b in desc[i]), float(weight[i])))
        f.write("\n".join(lines))
        if trailing_newline:
            f.write("\n")
