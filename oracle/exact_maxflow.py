"""Exact-arithmetic adjudication of tie-degenerate comparisons.  TEST INFRASTRUCTURE ONLY.

The reference reports T = {nodes that can reach the sink in the residual graph of its maximum flow} (what_segment(), reference
lib/maxflow/src/graph.h:561-571).  In exact arithmetic that set is UNIQUE: it is the sink side of the minimum cut whose source side is
largest (every maximum flow has the same set).  Float64 capacities are dyadic rationals, so the whole graph scales to integers
(Python's, of any size) and a maximum flow can be computed without a single rounding: Dinic's algorithm (Dinic 1970; the published
algorithm, nothing of the reference's BK) on an adjacency-array residual graph, then a backward search from the sink.

``canonical_sink_side(n, i, j, cap, rev, tr)`` returns that set for graphs small enough for pure Python (the fixtures of
tests/golden: 99 ... 4 080 nodes); ``adjudicate`` says which of two label sets -- the compiled reference's, the HIP path's -- equals it.
For graphs beyond that size ``relation`` still decides, from the two label sets alone, what exact arithmetic can say without a solve:
which cut is smaller as a rational, and -- on an exact tie -- which source side contains the other (the canonical source side
contains both).
"""
from collections import deque
from fractions import Fraction

import numpy as np


def _scaled(values):
    """float64 array -> (Python ints, shift): v == int / 2**shift exactly, one shift for all"""
    fr = [Fraction(float(v)) for v in values]
    shift = max([f.denominator.bit_length() - 1 for f in fr] or [0])
    return [int(f * (1 << shift)) for f in fr], shift


def canonical_sink_side(n, i, j, cap, rev, tr, limit=6000):
    """bool array, True where the node can reach the sink in the residual graph of an EXACT maximum flow; None beyond ``limit`` nodes.
    Edges (i[k], j[k]) carry cap[k] (i -> j) and rev[k] (j -> i); tr[v] > 0: source -> v, < 0: v -> sink (graph.h:416-425)."""
    n = int(n)
    if n > limit:
        return None
    i, j = np.asarray(i, dtype=np.int64), np.asarray(j, dtype=np.int64)
    vals = np.concatenate([np.asarray(cap, np.float64), np.asarray(rev, np.float64), np.abs(np.asarray(tr, np.float64))])
    ints, _ = _scaled(vals)
    m = i.size
    icap, irev, itr = ints[:m], ints[m:2 * m], ints[2 * m:]
    S, T = n, n + 1
    head, nxt, to, res = [-1] * (n + 2), [], [], []

    def arc(u, v, c, r):
        for a, b, w in ((u, v, c), (v, u, r)):
            to.append(b); res.append(w); nxt.append(head[a]); head[a] = len(to) - 1

    for k in range(m):
        arc(int(i[k]), int(j[k]), icap[k], irev[k])
    trf = np.asarray(tr, np.float64)
    for v in range(n):
        if trf[v] > 0:
            arc(S, v, itr[v], 0)
        elif trf[v] < 0:
            arc(v, T, itr[v], 0)
    while True:  # Dinic: level graph by BFS, blocking flow by iterative DFS with current-arc pointers
        level = [-1] * (n + 2)
        level[S] = 0
        q = deque([S])
        while q:
            u = q.popleft()
            e = head[u]
            while e >= 0:
                if res[e] > 0 and level[to[e]] < 0:
                    level[to[e]] = level[u] + 1
                    q.append(to[e])
                e = nxt[e]
        if level[T] < 0:
            break
        cur = list(head)
        while True:
            path, u = [], S
            while u != T:
                e = cur[u]
                while e >= 0 and not (res[e] > 0 and level[to[e]] == level[u] + 1):
                    e = nxt[e]
                cur[u] = e
                if e < 0:
                    if not path:
                        break
                    level[u] = -1  # dead end: retreat
                    u = to[path.pop() ^ 1]
                    continue
                path.append(e)
                u = to[e]
            if u != T:
                break
            f = min(res[e] for e in path)
            for e in path:
                res[e] -= f
                res[e ^ 1] += f
    # backward search from the sink over arcs with residual > 0 (arc e: u -> to[e]; its reverse e ^ 1 leaves to[e])
    can = [False] * (n + 2)
    can[T] = True
    q = deque([T])
    while q:
        v = q.popleft()
        e = head[v]
        while e >= 0:  # arcs v -> w; the arc w -> v is e ^ 1
            w = to[e]
            if not can[w] and res[e ^ 1] > 0:
                can[w] = True
                q.append(w)
            e = nxt[e]
    return np.array(can[:n], dtype=bool)


def relation(source_a, source_b, value_a, value_b):
    """what two cuts (bool source sides, exact rational capacities) say about each other without a solve"""
    a, b = np.asarray(source_a, bool).ravel(), np.asarray(source_b, bool).ravel()
    out = {"capacity": "equal" if value_a == value_b else ("first_smaller" if value_a < value_b else "second_smaller")}
    if (a == b).all():
        out["source_sides"] = "identical"
    elif (a | b == a).all():
        out["source_sides"] = "first_contains_second"
    elif (a | b == b).all():
        out["source_sides"] = "second_contains_first"
    else:
        out["source_sides"] = "incomparable"
    return out


def adjudicate(labels_hip, labels_ref, n, i, j, cap, rev, tr, value_hip, value_ref):
    """{"hip_vs_reference": relation, "canonical": "hip" | "reference" | "both" | "neither" | "not solved (n nodes)"}: which label set (True =
    source side) is the sink-tree complement exact arithmetic defines"""
    out = {"hip_vs_reference": relation(labels_hip, labels_ref, value_hip, value_ref)}
    t = canonical_sink_side(n, i, j, cap, rev, tr)
    if t is None:
        out["canonical"] = "not solved (%d nodes)" % int(n)
        return out
    src = ~t
    h, r = bool((np.asarray(labels_hip, bool).ravel() == src).all()), bool((np.asarray(labels_ref, bool).ravel() == src).all())
    out["canonical"] = "both" if h and r else ("hip" if h else ("reference" if r else "neither"))
    out["hip_differs_from_canonical"] = int((np.asarray(labels_hip, bool).ravel() != src).sum())
    out["reference_differs_from_canonical"] = int((np.asarray(labels_ref, bool).ravel() != src).sum())
    return out
