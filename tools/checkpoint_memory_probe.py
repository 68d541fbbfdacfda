"""Peak device memory and wall time of one training step of the Objaverse FEATURE stage (feature net with the hr net
nested as its middle, x0 objective; octfusion_obja_uncond.yaml: use_checkpoint True) on the shell-8 batch, with and
without activation checkpointing (backward.CHECKPOINT):

    python tools/checkpoint_memory_probe.py [--batch 8] --out gpurun_out/checkpoint_memory.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from octfusion_amd import backward as BW, configs, synthetic, training as TR
from octfusion_amd.graph_unet_union import UNet3DModel

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--out', default=None)
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
net = UNet3DModel(**configs.unet_params('obja_uncond', 'feature'))
net.load_state_dict(synthetic.random_state_dict(net))
net = net.to(dev).eval()
_, doc, _ = bench.build_tree('shell8', a.batch, dev)
codes = torch.randn(doc.total_num, 3, device=dev)
opt = TR.AdamW(TR.trainable_parameters(net, 'feature'), lr=1e-4)
res = {'config': 'obja_uncond feature stage, shell-8 x%d, N8 = %d, x0 objective' % (a.batch, doc.total_num),
       'module_flag_use_checkpoint': bool(net.unet_feature.use_checkpoint)}
base = torch.cuda.memory_allocated()
for name, flag in (('checkpointed', True), ('all_activations_kept', False)):
    BW.CHECKPOINT = flag
    try:
        TR.hr_stage_step(net, opt, codes, doc, 8, stage='feature', df_type='x0')      # warm-up (packs, tables)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        loss = TR.hr_stage_step(net, opt, codes, doc, 8, stage='feature', df_type='x0')
        torch.cuda.synchronize()
        res[name] = {'peak_GB': torch.cuda.max_memory_allocated() / 2 ** 30, 'step_ms': 1e3 * (time.perf_counter() - t0),
                     'loss': loss, 'resident_before_step_GB': base / 2 ** 30}
    except torch.cuda.OutOfMemoryError as e:
        res[name] = {'error': 'out of memory: %s' % str(e)[:200]}
    torch.cuda.empty_cache()
BW.CHECKPOINT = None
print(json.dumps(res))
if a.out:
    json.dump(res, open(a.out, 'w'), indent=1)
