"""Cascade orchestration: what the reference's ``OctFusionModel.sample`` does between the
stages (models/octfusion_model_union.py:354-401 for 2 stages, octfusion_model_union_3t.py:152-214
for 3), on device, for a whole batch of independent shapes.

    lr   : DDIM x0 branch on the dense [B, 8, 16^3] split codes (truncation 0.7 + sign)
    ->     split2octree_small -> DualOctree (depth 6)
    hr   : DDIM on [N6, C] node features (eps branch for ShapeNet, x0 for Objaverse)
    ->     (3-stage) split_large = last nnum[6] rows -> split2octree_large -> DualOctree (depth 8)
    feat : (3-stage) DDIM on [N8, 3] latent codes
    ->     GraphVAE.decode_code (grows 6 -> 8 on device when 2-stage)

    sdf  : NeuralMPU sweep of the decoded field on the resolution^3 lattice in [-sdf_scale, sdf_scale]^3
           (get_sdfs, octfusion_model_union.py:425-433) -- one kernel launch per shape.
Marching cubes (skimage on the host, octfusion_model_union.py:435-468) is outside the device path.
"""
import torch

from . import mpu, sampler
from .dual_octree import DualOctree
from .octree import split2octree_large, split2octree_small


class CascadeSampler:
    def __init__(self, net, cfg, vae=None, device=None):
        """net: graph_unet_union.UNet3DModel (EMA weights); cfg: a dict from octfusion_amd.configs."""
        self.net = net
        self.cfg = cfg
        self.vae = vae
        self.device = device or next(net.parameters()).device
        self.full_depth = cfg['full_depth']
        self.stages = list(cfg['unet_type'])
        self.df_type = list(cfg['df_type'])
        self.depths = list(cfg['input_depth'])

    @torch.no_grad()
    def _seed(self, value):
        torch.manual_seed(value)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(value)

    def sample(self, batch_size, ddim_steps=200, label=None, split_small=None, noises=None, sdf_resolution=None,
               sdf_scale=0.9, use_graph=None, seed=None, save_index=0):
        """Returns a dict with the per-stage results.  `noises` (optional) = dict of explicit
        init / step noise tensors per stage for reproducible runs.  sdf_resolution (e.g. 256) adds
        out['sdfs'] [B, R, R, R] (needs the VAE).
        seed / save_index: the reference's per-shape seeding of the 2-stage model -- seed_everything(seed +
        save_index) before the lr loop, seed_everything(seed) before the hr loop (octfusion_model_union.py:372,390);
        the 3-stage model does not reseed (octfusion_model_union_3t.py:168,184,204: commented out)."""
        noises = noises or {}
        out = {}
        S = 1 << self.full_depth
        reseed = seed is not None and len(self.stages) == 2
        if split_small is None:
            if seed is not None:
                self._seed(seed + save_index if reseed else seed)
            n = noises.get('lr', {})
            split_small = sampler.sample_loop(
                self.net, (batch_size, self.cfg['input_channels'][0], S, S, S), batch_size, ddim_steps, 'lr',
                self.df_type[0], self.device, label=label, truncated_index=sampler.TRUNCATED_TIME,
                init_noise=n.get('init'), step_noise=n.get('steps'), use_graph=use_graph)
        out['split_small'] = split_small
        octree = split2octree_small(split_small, self.depths[1], self.full_depth)
        out['octree_small'] = octree
        if len(self.stages) < 2:
            return out
        doctree = DualOctree(octree)
        n = noises.get('hr', {})
        if reseed:
            self._seed(seed)
        x = sampler.sample_loop(self.net, (doctree.total_num, self.cfg['input_channels'][1]), batch_size, ddim_steps,
                                'hr', self.df_type[1], self.device, doctree=doctree, unet_lr=self.net.unet_lr,
                                label=label, init_noise=n.get('init'), step_noise=n.get('steps'), use_graph=use_graph)
        out['hr'] = x
        if len(self.stages) >= 3:
            nn6 = int(octree.nnum[self.depths[1]])
            split_large = x[x.shape[0] - nn6:].contiguous()
            octree = split2octree_large(octree, split_large, self.depths[1])
            out['octree_large'] = octree
            doctree = DualOctree(octree)
            n = noises.get('feature', {})
            x = sampler.sample_loop(self.net, (doctree.total_num, self.cfg['input_channels'][2]), batch_size,
                                    ddim_steps, 'feature', self.df_type[2], self.device, doctree=doctree,
                                    unet_lr=self.net.unet_hr, label=label, init_noise=n.get('init'),
                                    step_noise=n.get('steps'), use_graph=use_graph)
            out['feature'] = x
        out['doctree'] = doctree
        if self.vae is not None:
            out['decoded'] = self.vae.decode_code(x, doctree)
            if sdf_resolution:
                out['sdfs'] = mpu.calc_sdf(out['decoded']['neural_mpu'], batch_size, size=sdf_resolution,
                                           bbmin=-sdf_scale, bbmax=sdf_scale)
        return out
