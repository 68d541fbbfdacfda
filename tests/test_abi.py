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
    # gconv3: 3 contraction modes x 2 geometries x 2 tile widths + the dense-GEMM instantiations (ND = 1: 2 pair modes x 2
    # geometries); gconv2: 3 x 2 x 2
    assert res[0][0] == 16 and res[1][0] == 12, res
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


def test_persistent_launch_schedule_invariants():
    """The schedule of the persistent planes GraphConv (csrc/ofx_gemm3.hip: whole-tile rounds + a stream-K region) is
    integer arithmetic shared by host and kernel (g3_bound): checked here on the CPU, through ofx_gconv3_plan, for the
    layer shapes of the three configs at several batch sizes and for random shapes.
      * the shares tile the region: bound(0) = 0, bound(G) = U, monotone;
      * every piece -- share boundary to tile edge, or boundary to boundary inside one tile -- has >= 8 k-steps (the
        kernel's loop shape), so no block ever gets an empty or too-short piece;
      * rounds * G + region tiles = all tiles, and the region holds between G and 2G tiles when there are >= G tiles;
      * shares of >= one tile (the 'early combine' regime) cut every tile at most once;
      * nearest snapping keeps every share within 4 k-steps of the even split (the first version: 7)."""
    import ctypes
    import random
    from octfusion_amd import _lib
    L = _lib.lib()
    KMIN = 8

    def plan(n_rows, cout, nkt, wm, ni, cus=256):
        buf = (ctypes.c_int32 * (5 + 2 * cus + 2))()
        G = L.ofx_gconv3_plan(n_rows, cout, nkt, wm, ni, cus, buf, len(buf))
        if G == 0:
            return None
        return dict(G=buf[0], q=buf[1], rem=buf[2], U=buf[3], rounds=buf[4], bounds=list(buf[5:5 + G + 1]))

    rnd = random.Random(7)
    shapes = []
    for B in (1, 2, 4, 8):
        for n1, d in ((27126, 6), (8450, 5), (4096, 4), (406050, 8), (96000, 7)):
            for cin, cout in ((128, 128), (256, 128), (384, 128), (256, 256), (512, 256), (768, 256), (512, 512), (64, 64)):
                shapes.append((n1 * B, cout, 7 * (cin // 32) + (7 * (d - 1) + 31) // 32))
    for _ in range(300):
        shapes.append((rnd.randint(1, 3_000_000), rnd.choice((64, 72, 128, 200, 256, 320, 512)), rnd.randint(1, 200)))
    seen_early = seen_late = seen_rounds = 0
    for n_rows, cout, nkt in shapes:
        for wm, ni in ((4, 2), (2, 2), (4, 1), (2, 1)):
            p = plan(n_rows, cout, nkt, wm, ni)
            tiles = -(-n_rows // (wm * 64)) * -(-cout // (64 * ni))
            if p is None:
                assert nkt < KMIN or tiles * nkt < 2 * KMIN * 8 or tiles * nkt >= 2 ** 31, (n_rows, cout, nkt, wm, ni)
                continue
            G, U, b = p['G'], p['U'], p['bounds']
            assert G % 8 == 0 and 8 <= G <= 256 * (1 if wm == 4 else 2)
            region_tiles = tiles - p['rounds'] * G
            assert U == region_tiles * nkt and p['q'] * G + p['rem'] == U
            if tiles >= G and p['rounds'] >= 0 and tiles >= 256 * (1 if wm == 4 else 2):
                assert G <= region_tiles < 2 * G, (tiles, G, p['rounds'])
                seen_rounds += p['rounds'] > 0
            assert b[0] == 0 and b[G] == U and all(b[i] <= b[i + 1] for i in range(G))
            cuts = {}
            for i in range(1, G):
                t, r = divmod(b[i], nkt)
                assert r == 0 or KMIN <= r <= nkt - KMIN, (n_rows, cout, nkt, wm, ni, i, b[i])
                even = i * p['q'] + min(i, p['rem'])
                assert abs(b[i] - even) <= (4 if nkt >= 2 * KMIN else KMIN), (b[i], even, nkt)
                if r:
                    cuts.setdefault(t, []).append(r)
            for t, rs in cuts.items():            # pieces between two cuts inside one tile
                rs = sorted(rs)
                assert all(y - x >= KMIN for x, y in zip(rs, rs[1:])), (n_rows, cout, nkt, rs)
            nonempty = [b[i + 1] - b[i] for i in range(G) if b[i + 1] > b[i]]
            assert min(nonempty) >= KMIN
            if p['rounds'] == 0:
                assert len(nonempty) == G or tiles * nkt < G * 2 * KMIN + nkt, 'a block without work and without rounds'
            if p['q'] >= nkt:
                seen_early += 1
                assert all(len(rs) == 1 for rs in cuts.values())
            else:
                seen_late += 1
    assert seen_early > 50 and seen_late > 50 and seen_rounds > 20
