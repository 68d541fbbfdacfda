"""Build libofx.so (the C-ABI HIP library) for gfx950 with hipcc, in-tree.

    python -m octfusion_amd.build [--force]

hipcc cross-compiles without a GPU.  Every translation unit is compiled to its own object
(in parallel, rebuilt only when it or a header changed) and the objects are linked into the
shared library.  The .so / .o files are git-ignored but travel with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libofx.so')
SOURCES = ['ofx_octree.hip', 'ofx_graph.hip', 'ofx_gemm.hip', 'ofx_gemm2.hip', 'ofx_norm.hip', 'ofx_dense.hip',
           'ofx_misc.hip', 'ofx_loss.hip', 'ofx_points.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(os.path.dirname(HERE), 'include', 'ofx.h'))
    return hs


def _newer(deps, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _newer([src] + hdrs, obj):
            jobs.append([hipcc] + FLAGS + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _newer(objs, LIB):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
