"""RCCL on the hardware the driver's single-GPU tiers have: a ONE-rank `nccl` (= RCCL on ROCm) process group on cuda:0
and every collective helper of octfusion_amd.dist forced through it -- the weight broadcast that replaces the
reference's per-rank checkpoint read (train.py:166-185, octfusion_model_union.py:525-545), the timing reductions of
bench.py / generate.py, the gradient averaging of the training path.  (World sizes > 1 are covered on CPU with gloo in
tests/test_dist.py; the 1 -> 8 GPU curve is the driver's.)"""
import pytest
import torch

from test_gpu_fullwidth import dev

pytestmark = pytest.mark.gpu


def test_rccl_world1_runs_every_collective_on_the_device():
    import torch.distributed as td
    from octfusion_amd import configs, dist, synthetic, training
    from octfusion_amd.graph_unet_union import UNet3DModel
    assert not td.is_initialized()
    rank, local_rank, world = dist.init(force=True)
    try:
        assert td.is_initialized() and td.get_backend() == 'nccl' and td.get_world_size() == 1 and (rank, world) == (0, 1)
        net = UNet3DModel(**configs.unet_params('snet_uncond', 'lr'))
        net.load_state_dict(synthetic.random_state_dict(net))
        net = net.to(dev())
        before = {k: v.clone() for k, v in net.state_dict().items()}
        versions = [p._version for p in net.parameters()]
        nbytes = dist.broadcast_module_(net, src=0)
        n_float = sum(v.numel() for v in list(net.parameters()) + list(net.buffers()) if v.is_floating_point())
        assert nbytes == 4 * n_float > 0
        for k, v in net.state_dict().items():
            assert torch.equal(v, before[k]), k                       # rank 0's own weights come back bit for bit
        assert all(p._version > v for p, v in zip(net.parameters(), versions))     # packed-weight caches see the write
        assert dist.max_over_ranks(3.25, dev()) == 3.25
        assert dist.gather_floats(1.5, dev()) == [1.5]
        g = {'a.weight': torch.randn(257, 33, device=dev()), 'b.bias': torch.randn(5, device=dev())}
        want = {k: v.clone() for k, v in g.items()}
        assert dist.all_reduce_mean_(g, bucket_bytes=1 << 12) == 4 * (257 * 33 + 5)       # two buckets
        for k in g:
            assert torch.equal(g[k], want[k])
        assert dist.all_reduce_mean_(g) == 4 * (257 * 33 + 5)                            # the set check runs again (every call, every rank)
        opt = training.AdamW({'w': torch.nn.Parameter(torch.randn(64, 8, device=dev()))})
        assert opt.sync_() == 4 * 64 * 8 * 3                                             # parameters + two moments
        dist.barrier()
        torch.cuda.synchronize()
    finally:
        dist.shutdown()
    assert not td.is_initialized()
