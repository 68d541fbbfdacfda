"""GraphVAE.decode_code at shell-8 size (depth 6 -> 8 growth on device) + the 256^3 SDF sweep: wall time per shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import synthetic, mpu as M
from octfusion_amd.graph_vae import GraphVAE
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small, split2octree_large

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
vae = GraphVAE(depth=8, channel_in=4, nout=4, full_depth=4, depth_stop=6, depth_out=8, resblk_type='basic',
               resblk_num=2, code_channel=16, embed_dim=3)
vae.load_state_dict(synthetic.random_state_dict(vae))
vae = vae.to(dev).eval()
oc6 = split2octree_small(synthetic.shell6_split(1, jitter=False).to(dev), 6, 4)
x, y, z, b = oc6.xyzb(6)
oc8 = split2octree_large(oc6, synthetic.shell8_split_large(x.cpu(), y.cpu(), z.cpu()).to(dev), 6)
doc8 = DualOctree(oc8)
code = torch.randn(doc8.csr(6)[2] if False else DualOctree(oc6).total_num, 3, device=dev)


def run(update):
    doc = DualOctree(oc6) if update else doc8
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = vae.decode_code(code if update else torch.randn(doc8.graph[6]['node_type'].numel() if False else code.shape[0], 3, device=dev), DualOctree(oc6) if update else doc, update_octree=update)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    sdf = M.calc_sdf(out['neural_mpu'], 1, 256, bbmin=-0.9, bbmax=0.9)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return t1 - t0, t2 - t1, out


for it in range(3):
    td, ts, out = run(True)
    print('decode_code (grow 6->8 on device): %.1f ms   sdf 256^3: %.2f ms   nnum[6..8] = %s' %
          (td * 1e3, ts * 1e3, [int(v) for v in out['octree_out'].nnum[6:9]]))
