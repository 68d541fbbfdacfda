"""Per-stage DDIM step time, eager launches vs hipGraph replay (snet_uncond, batch 8, shell-6)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import configs, graph_unet_union as U, sampler, synthetic
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
cfg = configs.SNET_UNCOND
net = U.UNet3DModel(**{k: v for k, v in dict(cfg, stage_flag='hr').items() if k != 'df_type'})
net.load_state_dict(synthetic.random_state_dict(net))
net = net.to(dev).eval()
B, steps = 8, 24
doc = DualOctree(split2octree_small(synthetic.shell6_split(B).to(dev), 6, 4))
for name, shape, kw in [('lr', (B, 8, 16, 16, 16), dict(unet_type='lr', df_type='x0', truncated_index=0.7)),
                        ('hr', (doc.total_num, 3), dict(unet_type='hr', df_type='eps', doctree=doc, unet_lr=net.unet_lr))]:
    for g in (False, True):
        sampler.sample_loop(net, shape, B, 4, device=dev, use_graph=g, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sampler.sample_loop(net, shape, B, steps, device=dev, use_graph=g, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('%s stage, batch %d: %.2f ms/step (%s, incl. capture)' % (name, B, dt / steps * 1e3, 'hipGraph' if g else 'eager'))
