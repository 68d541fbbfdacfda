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


def test_planes_kernels_have_no_scratch():
    """The planes GraphConv kernels load registers with inline asm and count the waits by hand: a compiler spill of
    such a register could be stored before its data has landed, and a spill reload inside a k-step makes hipcc drain the
    DMA queue.  So the kernels must compile without scratch ('register budget')."""
    import re
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from octfusion_amd import build
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

    def audit(src):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, 'k.s')
            subprocess.run([hipcc] + build.FLAGS + ['-S', '--cuda-device-only', '-o', out, os.path.join(build.CSRC, src)],
                           check=True, capture_output=True)
            text = open(out).read()
        rows = re.findall(r'\.name:\s+(\S*gconv\d_kernel\S*)\n(.*?)\.wavefront_size', text, re.S)
        assert rows, src
        bad = []
        for name, body in rows:
            priv = int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', body).group(1))
            spill = int(re.search(r'\.vgpr_spill_count:\s+(\d+)', body).group(1))
            if priv or spill:
                bad.append((name, priv, spill))
        return len(rows), bad
    with ThreadPoolExecutor(2) as ex:
        res = list(ex.map(audit, ['ofx_gemm3.hip', 'ofx_gemm2.hip']))
    assert res[0][0] == 12 and res[1][0] == 12, res        # 3 contraction modes x 2 geometries x 2 tile widths each
    assert not res[0][1] and not res[1][1], res


def test_state_dict_boundary_at_the_real_configs():
    """The drop-in boundary: `load_ckpt(strict=True)` (octfusion_model_union.py:525-545) binds the
    state_dict keys of the reference's nets.  tests/golden/boundary_keys.json holds the (key, shape) lists, in order,
    of the reference's OWN UNet3DModel built from its three diffusion YAMLs and of its GraphVAE built from the two VAE
    YAMLs (tests/golden/make_golden.py g_boundary); the product's modules must give the same lists."""
    import json
    import torch
    from octfusion_amd import configs
    from octfusion_amd.graph_unet_union import UNet3DModel
    from octfusion_amd.graph_vae import GraphVAE
    want = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'boundary_keys.json')))
    sizes = {}
    for name in ('snet_uncond', 'snet_cond', 'obja_uncond'):
        rec = want[name]
        with torch.device('meta'):
            net = UNet3DModel(**configs.unet_params(name, rec['stage_flag']))
        have = [[k, list(v.shape)] for k, v in net.state_dict().items()]
        assert have == rec['keys'], name
        sizes[name] = len(have)
    assert sizes == {'snet_uncond': 232, 'snet_cond': 312, 'obja_uncond': 379}
    for name, cfg in (('vae_snet', 'snet_uncond'), ('vae_obja_depth864', 'obja_uncond')):
        rec = want[name]
        kw = configs.vae_params(cfg)
        m = rec['model']          # the YAML's own numbers against the restated config
        assert (kw['depth'], kw['channel_in'], kw['nout'], kw['full_depth'], kw['depth_stop'], kw['depth_out'],
                kw['resblk_type'], kw['resblk_num'], kw['code_channel'], kw['embed_dim']) == (
            m['depth'], m['channel'], m['nout'], m['full_depth'], m['depth_stop'], m['depth_out'], m['resblock_type'],
            m['resblk_num'], m['code_channel'], m['embed_dim'])
        with torch.device('meta'):
            vae = GraphVAE(**kw)
        assert [[k, list(v.shape)] for k, v in vae.state_dict().items()] == rec['keys'], name
