"""Build libofx.so (the C-ABI HIP library) for gfx950 with hipcc, in-tree.

    python -m octfusion_amd.build [--force] [--ablation]

hipcc cross-compiles without a GPU.  Every translation unit is compiled to its own object (in parallel) and the
objects are linked into the shared library.  Rebuilds are decided by CONTENT, not by mtime: an object carries the
sha256 of its source, every header and the flags it was built from (``<obj>.sha``), and the library embeds the hash
of the whole source set (``ofx_build_hash()``), which ``octfusion_amd._lib`` compares with the tree before the
first call -- a stale ``libofx.so`` (the file is git-ignored and travels with the gpurun snapshot) is refused
instead of silently run.

``--ablation`` builds a SECOND library, libofx_ablation.so, with -DOFX_ABLATION: the timing-ablation variants of
the one-tile-per-block planes kernel (wrong results by construction) and the per-block clock-stamp buffer of both
planes kernels.  The product library never contains them; tools select the profiling build with OFX_LIB=<path>.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libofx.so')
SOURCES = ['ofx_octree.hip', 'ofx_graph.hip', 'ofx_gemm.hip', 'ofx_gemm2.hip', 'ofx_gemm3.hip', 'ofx_norm.hip',
           'ofx_dense.hip', 'ofx_misc.hip', 'ofx_loss.hip', 'ofx_points.hip', 'ofx_probe.hip', 'ofx_narrow.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']


def _headers():
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h'))
    hs.append(os.path.join(os.path.dirname(HERE), 'include', 'ofx.h'))
    return hs


def _sha(paths, extra=()):
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b'\0')
        with open(p, 'rb') as f:
            h.update(f.read())
        h.update(b'\0')
    for e in extra:
        h.update(e.encode() + b'\0')
    return h.hexdigest()


def source_hash(flags=None):
    """sha256 (first 16 hex digits) of every source, header and compile flag of the library."""
    flags = FLAGS if flags is None else flags
    return _sha([os.path.join(CSRC, s) for s in SOURCES] + _headers(), flags)[:16]


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


ABLATION_LIB = os.path.join(HERE, 'libofx_ablation.so')     # profiling build: OFX_LIB=<this> python tools/...


def build(force=False, verbose=False, ablation=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # the profiling build has its own objects and file: it never replaces the product library
    obj_dir, lib = (os.path.join(CSRC, '_obj_ablation'), ABLATION_LIB) if ablation else (OBJ, LIB)
    os.makedirs(obj_dir, exist_ok=True)
    flags = FLAGS + (['-DOFX_ABLATION'] if ablation else [])
    hdrs = _headers()
    jobs, objs, stamps = [], [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.replace('.hip', '.o'))
        want = _sha([src] + hdrs, flags)
        objs.append(obj)
        if force or not os.path.exists(obj) or _read(obj + '.sha') != want:
            jobs.append([hipcc] + flags + ['-c', src, '-o', obj])
            stamps.append((obj + '.sha', want))

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        subprocess.run(cmd, check=True)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
        for path, want in stamps:
            with open(path, 'w') as f:
                f.write(want)
    # the hash of the whole source set, compiled into the library
    tree = source_hash(flags)
    info_c = os.path.join(obj_dir, 'ofx_buildinfo.cpp')
    info_o = os.path.join(obj_dir, 'ofx_buildinfo.o')
    text = ('extern "C" const char* ofx_build_hash() { return "%s"; }\n'
            'extern "C" int ofx_build_ablation() { return %d; }\n' % (tree, 1 if ablation else 0))
    relink = bool(jobs) or force or not os.path.exists(lib)
    if _read(info_c) != text.strip() or not os.path.exists(info_o):
        with open(info_c, 'w') as f:
            f.write(text)
        run([hipcc, '-O2', '-fPIC', '-c', info_c, '-o', info_o])
        relink = True
    if relink:
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + [info_o])
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True, ablation='--ablation' in sys.argv))
