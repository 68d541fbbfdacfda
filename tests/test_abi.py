"""CPU-side checks of the C-ABI boundary: the library builds/loads, exports every symbol
include/ofx.h declares, and the product refuses to run without a GPU (no fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'ofx.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ofx_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from octfusion_amd import build, _lib
    build.build()
    L = _lib.lib()
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), 'libofx.so does not export %s' % n
    # and the ctypes table binds exactly the header's functions
    assert sorted(_lib.EXPORTS) == names
    assert L.ofx_version() >= 1
    assert L.ofx_status_string(-1) == b'invalid argument'
    assert L.ofx_graphconv_packed_k(128, 5) == 7 * 128 + 64
    assert L.ofx_packed_k(931) == 960


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_product_fails_loudly_without_gpu():
    from octfusion_amd import _lib, ops
    from octfusion_amd.octree import Octree, split2octree_small
    with pytest.raises(_lib.OfxError):
        Octree(4, 2, 1)
    with pytest.raises(_lib.OfxError):
        split2octree_small(torch.zeros(1, 8, 4, 4, 4), 4, 2)
    with pytest.raises(_lib.OfxError):
        ops.gather_mean(torch.zeros(4, 4), torch.zeros(29, dtype=torch.int32), torch.zeros(1, dtype=torch.int32))
    # training side: losses, NeuralMPU gradient, point-cloud build have no CPU path either
    from octfusion_amd import vae_training as VT
    from octfusion_amd.octree import Points, build_octree_batch
    with pytest.raises(_lib.OfxError):
        VT.octree_ce(torch.zeros(4, 2), torch.zeros(4, dtype=torch.int32))
    with pytest.raises(_lib.OfxError):
        VT.sdf_reg_loss(torch.zeros(4), torch.zeros(4, 3), torch.zeros(4), torch.zeros(4, 3))
    with pytest.raises(_lib.OfxError):
        VT.kl_sample(torch.zeros(4, 6), torch.zeros(4, 3), 3)
    with pytest.raises(_lib.OfxError):
        build_octree_batch([Points(torch.zeros(8, 3), torch.zeros(8, 3))], 6, 4)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'octfusion_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f
