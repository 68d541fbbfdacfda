"""Build libofx.so (the C-ABI HIP library) for gfx950 with hipcc, in-tree.

    python -m octfusion_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the
gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libofx.so')
SOURCES = ['ofx_octree.hip', 'ofx_graph.hip', 'ofx_gemm.hip', 'ofx_norm.hip', 'ofx_dense.hip', 'ofx_misc.hip']


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), 'include', 'ofx.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
           '-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
