"""Wall time of one training step (forward + backward + AdamW + EMA) at the bench size: snet_uncond union net,
shell-6 octrees, batch 8.  Stage 2 (hr, eps objective) and stage 1 (lr, x0 objective)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import configs, graph_unet_union as U, synthetic, training as TR
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
cfg = configs.SNET_UNCOND
net = U.UNet3DModel(**{k: v for k, v in dict(cfg, stage_flag='hr').items() if k != 'df_type'})
net.load_state_dict(synthetic.random_state_dict(net))
net = net.to(dev).eval()
ema = copy.deepcopy(net)
B = 8
split = synthetic.shell6_split(B).to(dev)
doc = DualOctree(split2octree_small(split, 6, 4))
opt = TR.AdamW(dict(net.named_parameters()), lr=1e-4)
codes = torch.randn(doc.total_num, 3, device=dev)


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    out = [fn() for _ in range(n)]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


t_hr, l_hr = timeit(lambda: TR.hr_stage_step(net, opt, codes, doc, 6, ema=ema))
print('stage 2 (hr + nested lr), batch %d, N6 = %d: %.1f ms per training step (losses %s)' %
      (B, doc.total_num, t_hr * 1e3, ['%.4f' % v for v in l_hr]))
opt_lr = TR.AdamW(dict(net.unet_lr.named_parameters()), lr=1e-4)
t_lr, l_lr = timeit(lambda: TR.lr_stage_step(net.unet_lr, opt_lr, split, ema=None))
print('stage 1 (lr), batch %d: %.1f ms per training step (losses %s)' % (B, t_lr * 1e3, ['%.4f' % v for v in l_lr]))
print('peak memory %.1f GB' % (torch.cuda.max_memory_allocated() / 2 ** 30))
