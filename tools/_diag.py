import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hpsdf_loader
import oracle as O
H = hpsdf_loader.load(); ctx = H.Context(0)
one = 0x3F800000
for first in (0, 0x80000000):
    n = one // 61 + 1
    got, want = H.selftest_acosf(ctx, first, 61, n), O.acosf_batch(first, 61, n)
    bad = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
    x = (first + bad.astype(np.uint64) * 61).astype(np.uint32).view(np.float32)
    print("first %08x: %d mismatches of %d" % (first, len(bad), n))
    if len(bad):
        print("  |x| range of mismatches: %.9g .. %.9g" % (np.abs(x).min(), np.abs(x).max()))
        h = np.histogram(np.abs(x), bins=[0, 1e-38, 1e-30, 1e-20, 1e-10, 1e-5, 0.1, 0.5, 0.75, 0.9, 0.99, 1.0])
        print("  histogram", h)
        for i in bad[:8]:
            xi = np.uint32(first + int(i) * 61)
            print("   x bits %08x x %.9g got %08x want %08x" % (xi, np.array([xi], np.uint32).view(np.float32)[0], got.view(np.uint32)[i], want.view(np.uint32)[i]))
