"""FPRev-style probe: the summation tree NumPy's (1,n)@(n,) [sdot] or (r,n)@(n,) row i [sgemv] uses, from masked inputs."""
import numpy as np, sys
import os
M = np.float32(2.0 ** int(os.environ.get("MEXP", "40")))

def lca_sizes(n, call):
    s = np.zeros((n, n), dtype=np.int64)
    for i in range(n):
        for j in range(i + 1, n):
            a = np.ones(n, dtype=np.float32); a[i] = M; a[j] = -M
            s[i, j] = s[j, i] = n - int(round(float(call(a))))
    return s

def build(S, s):
    S = sorted(S)
    if len(S) == 1: return S[0]
    i = S[0]
    top = max(s[i, j] for j in S if j != i)
    inner = [i] + [j for j in S if j != i and s[i, j] < top]
    rest = [j for j in S if j != i and s[i, j] == top]
    # rest may hold several sibling subtrees: split by mutual LCA size < top
    groups = []
    todo = list(rest)
    while todo:
        g0 = todo[0]
        g = [g0] + [j for j in todo[1:] if s[g0, j] < top]
        groups.append(g); todo = [j for j in todo if j not in g]
    return tuple([build(inner, s)] + [build(g, s) for g in groups])

def fmt(t):
    if isinstance(t, tuple): return "(" + " ".join(fmt(c) for c in t) + ")"
    return str(t)

if __name__ == "__main__":
    r = int(sys.argv[1]); ns = [int(v) for v in sys.argv[2].split(",")]
    row = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    for n in ns:
        x = np.ones(n, dtype=np.float32)
        def call(a):
            P = np.ones((r, n), dtype=np.float32); P[row] = a
            return (P @ x)[row]
        s = lca_sizes(n, call)
        print(n, fmt(build(range(n), s)))
