"""Where query_one_kernel - ONE query's lookup / count / order / cut - spends its time: s_memrealtime ticks (10 ns) at the kernel's
stage boundaries, from a probe build of the library:
    python tools/ab_build.py tools/_ab/lib_oneprobe.so -DLSHRS_AB_ONE_PROBE
    LSHRS_HIP_LIBRARY=$PWD/tools/_ab/lib_oneprobe.so python tools/one_query_stages.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lshrs_amd import LSHRS, InMemoryStorage, _native

rng = np.random.default_rng(0)
n, dim = 1_000_000, 768
data = rng.standard_normal((n, dim)).astype(np.float32)
idx = LSHRS(dim=dim, num_perm=256, storage=InMemoryStorage(), packed_ingest=True)
idx.index(np.arange(n), data)
idx.set_corpus(torch.from_numpy(data).cuda())
q = data[rng.choice(n, 300, replace=False)] + 0.1 * rng.standard_normal((300, dim)).astype(np.float32)
lib = ctypes.CDLL(_native.LIBRARY)
buf = (ctypes.c_ulonglong * 16)()
names = ["copy of the vector", "lookup (+ scan)", "gather of the members", "sort 1", "dedupe scan, heads, run lengths, keys", "sort 2", "emit ids", "cut + publish"]
acc = np.zeros(8)
for i, v in enumerate(q):
    idx.get_top_k(v, topk=10)
    torch.cuda.synchronize()
    lib.lshrs_ab_one_probe(buf)
    t = np.array(buf[:9], dtype=np.float64)
    if i >= 50:
        acc += np.diff(t) * 0.01
for nm, us in zip(names, acc / 250):
    print(f"{nm:45s} {us:6.2f} us")
print(f"{'total':45s} {acc.sum() / 250:6.2f} us")
