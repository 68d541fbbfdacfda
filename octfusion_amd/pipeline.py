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


class StepNoise:
    """``step_noise`` of sampler.sample_loop for a stage whose per-step tensors are too large to draw up front (the
    feature stage at B = 8: 200 steps x 39 MB): element i is drawn when the loop asks for it.  It is a STREAM -- the
    reference draws on every x0-branch step, used or not (octfusion_model_union.py:338-343) -- so steps that are never
    asked for are still drawn, in order, and an element can be asked for once, in step order."""

    def __init__(self, draw, n):
        self._draw, self._n, self._pos = draw, n, 0

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if not self._pos <= i < self._n:
            raise IndexError('step noise is a stream: element %d asked for after %d were drawn' % (i, self._pos))
        while self._pos < i:
            self._draw()                     # a step that used no noise: its draw is consumed all the same
            self._pos += 1
        self._pos += 1
        return self._draw()

    def finish(self):
        """Consume the draws nobody asked for (the last step of a stage uses no noise, the reference draws it anyway):
        the next stage of a model that does not reseed continues the same streams."""
        while self._pos < self._n:
            self._draw()
            self._pos += 1


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

    def _seed(self, value):
        torch.manual_seed(value)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(value)

    # ---- per-shape noise streams -------------------------------------------------------------------------------
    # The reference generates ONE shape per call and seeds the global RNG per shape (train.py:166-185 ->
    # octfusion_model_union.py:372,390: seed + save_index before the lr loop, seed before the hr loop; the 3-stage
    # model never reseeds).  A batch of shapes reproduces that exactly if every shape draws from its own generator
    # in the order the reference's loop would: initial noise, then one tensor per x0-branch step that uses noise.
    def _gen(self, seed):
        g = torch.Generator(device=self.device)
        g.manual_seed(int(seed))
        return g

    def _dense_noise(self, gens, shape1, ddim_steps, df_type, truncated):
        """(init [B, ...], step list) for a dense stage from per-shape generators."""
        init = torch.cat([torch.randn(shape1, generator=g, device=self.device) for g in gens])
        steps = [None] * ddim_steps
        if df_type == 'x0':
            # the reference evaluates randn_like on EVERY x0-branch step (torch.where picks it or zeros,
            # octfusion_model_union.py:338-343): the stream advances even where the draw is dropped
            for i, (_, t_next) in enumerate(sampler.sampling_times(ddim_steps)):
                draw = torch.cat([torch.randn(shape1, generator=g, device=self.device) for g in gens])
                if bool(t_next > truncated):
                    steps[i] = draw
        return init, steps

    def _node_noise(self, gens, doctree, depth, channels, ddim_steps, df_type):
        """The same for a sparse stage: shape b's rows of the batched tensor (all depth blocks, in order) are exactly
        the rows of the tensor a batch-of-one doctree of that shape has, so its noise is drawn at that size and
        scattered to its rows."""
        bid = doctree.batch_id32(depth).long()
        order = torch.argsort(bid, stable=True)
        counts = torch.bincount(bid, minlength=doctree.batch_size).tolist()

        def draw():
            out = torch.empty(bid.shape[0], channels, dtype=torch.float32, device=self.device)
            out[order] = torch.cat([torch.randn(n, channels, generator=g, device=self.device)
                                    for g, n in zip(gens, counts)])
            return out
        init = draw()
        steps = StepNoise(draw, ddim_steps) if df_type == 'x0' else [None] * ddim_steps
        return init, steps

    @torch.no_grad()
    def sample(self, *args, **kwargs):
        """_sample_once, retried ONCE when the fp16x3 range guard trips (ops.raise_on_range_error: operands beyond the
        fp16 range; the process has been switched to bf16x3 by then).  With per-shape generators or a seed the retry
        draws the same noise; a caller that relies on the global RNG state gets a fresh draw."""
        from . import ops
        before = ops.get_precision()
        timings = kwargs.get('timings')
        try:
            return self._sample_once(*args, **kwargs)
        except ops.OfxRangeError:
            # retry only if raise_on_range_error really moved the process to another precision (it does for 'fp16x3'
            # with AUTO_RANGE_FALLBACK; in the reduced 'fp16' mode, or with the fallback off, a second pass in the same
            # arithmetic would fail the same way)
            if ops.get_precision() == before:
                raise
            import warnings
            warnings.warn('octfusion_amd: fp16x3 range guard tripped -- sampling again in %s' % ops.get_precision())
            if timings is not None:
                timings.clear()                      # the failed pass's phases are not part of the result
            return self._sample_once(*args, **kwargs)

    def _sample_once(self, batch_size, ddim_steps=200, label=None, split_small=None, noises=None, sdf_resolution=None,
                     sdf_scale=0.9, use_graph=None, seed=None, save_index=0, shape_indices=None, timings=None):
        """Returns a dict with the per-stage results.  `noises` (optional) = dict of explicit
        init / step noise tensors per stage for reproducible runs.  sdf_resolution (e.g. 256) adds
        out['sdfs'] [B, R, R, R] (needs the VAE).
        seed / save_index: the reference's per-shape seeding of the 2-stage model -- seed_everything(seed +
        save_index) before the lr loop, seed_everything(seed) before the hr loop (octfusion_model_union.py:372,390);
        the 3-stage model does not reseed (octfusion_model_union_3t.py:168,184,204: commented out).
        shape_indices (list of batch_size result indices, needs seed): every shape of the batch gets the noise the
        reference's one-shape-per-call loop would draw for that result index -- a batch of shapes generates what
        batch_size calls with save_index = shape_indices[b] would.
        timings (optional dict): filled with seconds per phase (host-synchronised: adds a few syncs)."""
        import time as _time
        noises = dict(noises or {})
        out = {}
        S = 1 << self.full_depth
        two_stage = len(self.stages) == 2
        reseed = seed is not None and two_stage
        gens = None
        if shape_indices is not None:
            assert seed is not None and len(shape_indices) == batch_size
            gens = [self._gen(seed + int(i)) for i in shape_indices]

        def lap(name, t0):
            if timings is not None:
                if self.device.type == 'cuda':
                    torch.cuda.synchronize()
                timings[name] = timings.get(name, 0.0) + _time.perf_counter() - t0
            return _time.perf_counter()
        t0 = lap('_start', _time.perf_counter())
        if split_small is None:
            if gens is not None and 'lr' not in noises:
                init, steps = self._dense_noise(gens, (1, self.cfg['input_channels'][0], S, S, S), ddim_steps,
                                                self.df_type[0], sampler.TRUNCATED_TIME)
                noises['lr'] = {'init': init, 'steps': steps}
            elif seed is not None:
                self._seed(seed + save_index if reseed else seed)
            n = noises.get('lr', {})
            split_small = sampler.sample_loop(
                self.net, (batch_size, self.cfg['input_channels'][0], S, S, S), batch_size, ddim_steps, 'lr',
                self.df_type[0], self.device, label=label, truncated_index=sampler.TRUNCATED_TIME,
                init_noise=n.get('init'), step_noise=n.get('steps'), use_graph=use_graph)
            t0 = lap('lr_steps', t0)
        out['split_small'] = split_small
        octree = split2octree_small(split_small, self.depths[1], self.full_depth)
        out['octree_small'] = octree
        if len(self.stages) < 2:
            return out
        doctree = DualOctree(octree)
        t0 = lap('octree_and_graph', t0)
        if gens is not None and 'hr' not in noises:
            if two_stage:
                gens = [self._gen(seed) for _ in shape_indices]          # :390 -- the SAME seed for every shape
            init, steps = self._node_noise(gens, doctree, self.depths[1], self.cfg['input_channels'][1], ddim_steps,
                                           self.df_type[1])
            noises['hr'] = {'init': init, 'steps': steps}
        elif reseed:
            self._seed(seed)
        n = noises.get('hr', {})
        x = sampler.sample_loop(self.net, (doctree.total_num, self.cfg['input_channels'][1]), batch_size, ddim_steps,
                                'hr', self.df_type[1], self.device, doctree=doctree, unet_lr=self.net.unet_lr,
                                label=label, init_noise=n.get('init'), step_noise=n.get('steps'), use_graph=use_graph)
        if isinstance(n.get('steps'), StepNoise):
            n['steps'].finish()
        out['hr'] = x
        t0 = lap('hr_steps', t0)
        if len(self.stages) >= 3:
            nn6 = int(octree.nnum[self.depths[1]])
            split_large = x[x.shape[0] - nn6:].contiguous()
            octree = split2octree_large(octree, split_large, self.depths[1])
            out['octree_large'] = octree
            doctree = DualOctree(octree)
            t0 = lap('octree_and_graph', t0)
            if gens is not None and 'feature' not in noises:
                init, steps = self._node_noise(gens, doctree, self.depths[2], self.cfg['input_channels'][2],
                                               ddim_steps, self.df_type[2])
                noises['feature'] = {'init': init, 'steps': steps}
            n = noises.get('feature', {})
            x = sampler.sample_loop(self.net, (doctree.total_num, self.cfg['input_channels'][2]), batch_size,
                                    ddim_steps, 'feature', self.df_type[2], self.device, doctree=doctree,
                                    unet_lr=self.net.unet_hr, label=label, init_noise=n.get('init'),
                                    step_noise=n.get('steps'), use_graph=use_graph)
            if isinstance(n.get('steps'), StepNoise):
                n['steps'].finish()
            out['feature'] = x
            t0 = lap('feature_steps', t0)
        out['doctree'] = doctree
        if self.vae is not None:
            out['decoded'] = self.vae.decode_code(x, doctree)
            t0 = lap('vae_decode', t0)
            if sdf_resolution:
                out['sdfs'] = mpu.calc_sdf(out['decoded']['neural_mpu'], batch_size, size=sdf_resolution,
                                           bbmin=-sdf_scale, bbmax=sdf_scale)
                t0 = lap('sdf', t0)
        if timings is not None:
            timings.pop('_start', None)
        return out
