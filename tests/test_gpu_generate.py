"""The generate driver end to end on the GPU (train.py:166-185 -> octfusion_model_union.py:354-401): lr loop -> octree
-> hr loop -> GraphVAE.decode_code -> SDF lattice, through the CLI function, on a narrow net; and the claim that a
BATCH of shapes draws, per shape, exactly the noise the reference's one-shape-per-call loop draws."""
import json
import os

import pytest
import torch

import common as C
from test_gpu_fullwidth import dev

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _tiny():
    from octfusion_amd import configs
    configs.CONFIGS['tiny_uncond'] = dict(configs.SNET_UNCOND, model_channels=[32, 32])
    configs.VAES['tiny_uncond'] = configs.VAES['snet_uncond']
    return 'tiny_uncond'


def test_generate_cli_through_sdf(tmp_path, capsys):
    from octfusion_amd import generate as G, ops
    name = _tiny()
    out_dir = str(tmp_path / 'gen')
    res = G.main(['--config', name, '--shapes', '3', '--steps', '4', '--batch', '2', '--sdf-resolution', '32',
                  '--seed', '5', '--out', out_dir])
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith('{')][-1]
    assert json.loads(line)['shapes'] == 3
    assert res['rank0_indices'] == [0, 1, 2] and res['shapes_per_call'] == 2 and res['seconds_per_shape'] > 0
    for phase in ('lr_steps', 'octree_and_graph', 'hr_steps', 'vae_decode', 'sdf'):
        assert res['rank0_phase_seconds'][phase] > 0, phase
    for i in range(3):
        small = torch.load(os.path.join(out_dir, str(i), 'split_small.pth'))
        assert tuple(small.shape) == (8, 16, 16, 16) and set(small.unique().tolist()) <= {-1.0, 1.0}
        sdf = torch.load(os.path.join(out_dir, str(i), 'sdf.pt'))
        assert tuple(sdf.shape) == (32, 32, 32) and bool(torch.isfinite(sdf).all())
    assert not ops.sync_error(dev())


def test_batched_shapes_draw_the_reference_per_shape_noise():
    """octfusion_model_union.py:372,390 + train.py:181: shape `i` is generated alone with seed_everything(seed + i)
    before the lr loop and seed_everything(seed) before the hr loop.  CascadeSampler's per-shape generators must give
    a batch exactly those draws: initial noise and the k-th step noise of every shape equal what the global RNG gives
    after manual_seed, in the reference's call order (randn(shape), then one randn_like per x0-branch step)."""
    from octfusion_amd import configs, pipeline, sampler
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.graph_unet_union import UNet3DModel
    from octfusion_amd.octree import split2octree_small
    with torch.device('meta'):
        net = UNet3DModel(**configs.unet_params('snet_uncond', 'hr'))
    cs = pipeline.CascadeSampler(net, configs.CONFIGS['snet_uncond'], device=dev())
    seed, idxs, steps = 11, [4, 9, 2], 10
    gens = [cs._gen(seed + i) for i in idxs]
    init, noise = cs._dense_noise(gens, (1, 8, 16, 16, 16), steps, 'x0', sampler.TRUNCATED_TIME)
    for b, i in enumerate(idxs):
        torch.manual_seed(seed + i)
        torch.cuda.manual_seed(seed + i)
        want0 = torch.randn((1, 8, 16, 16, 16), device=dev())
        assert torch.equal(init[b:b + 1], want0)
        for k, (_, t_next) in enumerate(sampler.sampling_times(steps)):
            wk = torch.randn_like(want0)                      # the reference draws on EVERY step
            if noise[k] is not None:
                assert bool(t_next > sampler.TRUNCATED_TIME)
                assert torch.equal(noise[k][b:b + 1], wk)
            else:
                assert not bool(t_next > sampler.TRUNCATED_TIME)
    # sparse stage: shape b's rows of the batched tensor == the tensor of a batch-of-one doctree of that shape
    from octfusion_amd import synthetic
    split = synthetic.shell6_split(3, jitter=True).to(dev())
    doc = DualOctree(split2octree_small(split, 6, 4))
    gens = [cs._gen(seed) for _ in idxs]
    init, _ = cs._node_noise(gens, doc, 6, 3, steps, 'eps')
    bid = doc.batch_id32(6).long()
    for b in range(3):
        doc1 = DualOctree(split2octree_small(split[b:b + 1], 6, 4))
        torch.manual_seed(seed)
        torch.cuda.manual_seed(seed)
        want = torch.randn((doc1.total_num, 3), device=dev())
        assert torch.equal(init[bid == b], want)


def test_write_outputs_keeps_the_depth8_cascade_readable(tmp_path):
    """generate.write_outputs on a 3-stage result: <i>/split_large.pth per shape (tools/gen_split.py:50-54), which
    checkpoint.read_splits + split2octree_large turn back into the same depth-8 octree, shape by shape."""
    from octfusion_amd import checkpoint, configs, generate as G, synthetic
    from octfusion_amd.octree import split2octree_large, split2octree_small
    cfg = configs.CONFIGS['obja_uncond']
    split = synthetic.shell6_split(2, jitter=True).to(dev())
    oc6 = split2octree_small(split, 6, 4)
    x6, y6, z6, _ = oc6.xyzb(6)
    oc8 = split2octree_large(oc6, synthetic.shell8_split_large(x6.cpu(), y6.cpu(), z6.cpu()).to(dev()), 6)
    out_dir = str(tmp_path / 'gen3')
    G.write_outputs(out_dir, [7, 3], {'octree_small': oc6, 'octree_large': oc8}, cfg)
    for b, i in enumerate([7, 3]):
        small, large = checkpoint.read_splits(os.path.join(out_dir, str(i)), dev())
        assert large is not None and large.shape[1] == 8 and set(large.unique().tolist()) <= {-1.0, 1.0}
        one6 = split2octree_small(small, 6, 4)
        assert large.shape[0] == int(one6.nnum[6])
        one8 = split2octree_large(one6, large, 6)
        ref6 = split2octree_small(split[b:b + 1], 6, 4)
        xr, yr, zr, _ = ref6.xyzb(6)
        ref8 = split2octree_large(ref6, synthetic.shell8_split_large(xr.cpu(), yr.cpu(), zr.cpu()).to(dev()), 6)
        for d in (7, 8):
            assert torch.equal(one8.keys[d], ref8.keys[d]) and torch.equal(one8.children[d], ref8.children[d])
