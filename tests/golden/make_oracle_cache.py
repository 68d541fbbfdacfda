"""Compute the CPU-oracle fixtures of the full-width GPU parity tests (tests/golden/oracle_cache/*.pt).

    python tests/golden/make_oracle_cache.py [--only SUBSTRING] [--force]

Runs on a CPU-only machine: imports the GPU test modules (their oracle cases are registered at import and touch no
device), runs every registered case through `oracle/` and stores the result keyed by `oracle_cache.digest()` (sha256 of
oracle/*.py + the input builders).  The `-m gpu` tests then compare the HIP path against these files instead of
re-running the oracle on the GPU box's 16-CPU quota; a fixture whose digest no longer matches the tree fails the test
that reads it.  `oracle/` itself is pinned to the reference by tests/test_oracle_golden.py (fixtures generated from the
imported, unmodified reference by tests/golden/make_golden.py).
"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import oracle_cache as OC  # noqa: E402

MODULES = ['test_gpu_fullwidth', 'test_gpu_precision', 'test_gpu_persistent', 'test_gpu_feature_b8']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    ap.add_argument('--force', action='store_true', help='recompute fixtures whose digest is current')
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    for m in MODULES:
        importlib.import_module(m)
    dg = OC.digest()
    total = 0
    for name in sorted(OC.REGISTRY):
        if a.only and a.only not in name:
            continue
        p = OC.path(name)
        if os.path.exists(p) and not a.force:
            try:
                if torch.load(p, weights_only=False)['digest'] == dg:
                    print('%-40s current' % name)
                    total += os.path.getsize(p)
                    continue
            except Exception:      # noqa: BLE001
                pass
        t0 = time.time()
        OC.write(name, OC.REGISTRY[name]())
        total += os.path.getsize(p)
        print('%-40s %6.1f s  %8.1f KB' % (name, time.time() - t0, os.path.getsize(p) / 1024), flush=True)
    # drop fixtures of cases that no longer exist
    if not a.only:
        for f in sorted(os.listdir(OC.DIR)):
            if f.endswith('.pt') and f[:-3] not in OC.REGISTRY:
                os.remove(os.path.join(OC.DIR, f))
                print('removed stale', f)
    print('digest', dg, 'total %.1f MB' % (total / 1e6))


if __name__ == '__main__':
    main()
