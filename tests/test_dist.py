"""World-size-2 gloo tests (CPU) of the multi-GPU path: shard rule, the single flattened weight
broadcast, max-over-ranks timing."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from octfusion_amd import dist as D
    r, lr, w = D.init(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                        # ranks start with DIFFERENT weights
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3))
    nbytes = D.broadcast_module_(net, src=0)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()] +
                     [b.reshape(-1).float() for b in net.buffers() if b.is_floating_point()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    tmax = D.max_over_ranks(1.0 + rank, torch.device('cpu'))
    D.barrier()
    if rank == 0:
        torch.save({'same': same, 'nbytes': nbytes, 'tmax': tmax, 'numel': flat.numel()}, out)
    dist.destroy_process_group()


def test_broadcast_and_timing_world2(tmp_path):
    out = str(tmp_path / 'r.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r['same']
    assert r['nbytes'] == 4 * r['numel']
    assert r['tmax'] == 2.0


def test_shard_rule():
    sys.path.insert(0, ROOT)
    from octfusion_amd.dist import shard_indices
    items = [shard_indices(10, r, 4) for r in range(4)]
    assert items[1] == [1, 5, 9]
    assert sorted(sum(items, [])) == list(range(10))


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher re-executes itself under torch.distributed.run with one rank per
    GPU; --bootstrap-only stops after rendezvous + one broadcast + the max-reduction (no HIP call), on gloo here."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--bootstrap-only'],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    rec = json.loads(line)
    assert rec == {'bootstrap': 'ok', 'world': 2, 'max_rank_seen': 1.0, 'broadcast_value': 0.0, 'shard_of_10': [0, 2, 4, 6, 8],
                   'ranks': [[0, 0, 'cpu'], [1, 1, 'cpu']], 'shapes_of_64_per_rank': [32.0, 32.0],
                   'shard_of_64_last_rank': list(range(1, 64, 2))}


def test_bench_bootstrap_at_world_8():
    """the launch the driver uses for the 8-GPU line (`--gpus 8`, one rank per GPU) rendezvouses,
    runs every collective kind of the timed path, shards 64 shapes by the reference's rule (train.py:166-185: rank r takes
    i = r, r + 8, ...) and prints EXACTLY ONE JSON line on stdout -- over gloo on this CPU-only machine."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env['OMP_NUM_THREADS'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--bootstrap-only'],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['world'] == 8 and rec['max_rank_seen'] == 7.0 and rec['broadcast_value'] == 0.0
    assert rec['ranks'] == [[i, i, 'cpu'] for i in range(8)]
    assert rec['shapes_of_64_per_rank'] == [8.0] * 8
    assert rec['shard_of_64_last_rank'] == [7, 15, 23, 31, 39, 47, 55, 63]


def test_bench_rank_command_is_the_documented_one():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.rank_bootstrap_cmd(['--gpus', '8', '--steps', '5'], 8, port=29511)
    assert cmd[1:] == ['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr',
                       '127.0.0.1', '--master-port', '29511', os.path.join(ROOT, 'bench.py'), '--gpus', '8',
                       '--steps', '5']


def test_generate_shards_like_the_reference():
    """train.py:166-185: result_index = iter_i * world + rank, stop at the total."""
    sys.path.insert(0, ROOT)
    from octfusion_amd.dist import shard_indices
    world, total = 8, 21
    for rank in range(world):
        ref = []
        it = 0
        while it * world + rank < total:
            ref.append(it * world + rank)
            it += 1
        assert shard_indices(total, rank, world) == ref


def _grad_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from octfusion_amd import dist as D
    D.init(backend='gloo')
    g = torch.Generator().manual_seed(7 + rank)
    grads = {'b.weight': torch.randn(33, 5, generator=g), 'a.bias': torch.randn(5, generator=g),
             'c.weights': torch.randn(1000, generator=g)}
    mine = {k: v.clone() for k, v in grads.items()}
    nbytes = D.all_reduce_mean_(grads, bucket_bytes=1024)            # small buckets: several collectives
    gathered = {}
    for k, v in mine.items():
        parts = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(parts, v)
        gathered[k] = sum(parts) / world
    ok = all(torch.allclose(grads[k], gathered[k], rtol=0, atol=1e-6) for k in grads)
    if rank == 0:
        torch.save({'ok': ok, 'nbytes': nbytes, 'numel': sum(v.numel() for v in grads.values())}, out)
    dist.destroy_process_group()


def _diverging_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from octfusion_amd import dist as D
    D.init(backend='gloo')
    grads = {'a.weight': torch.ones(9, 4) * (rank + 1), 'b.bias': torch.ones(4)}
    D.all_reduce_mean_(grads)                                        # step 1: the same set everywhere
    ok1 = bool(torch.allclose(grads['a.weight'], torch.full((9, 4), 1.5)))
    if rank == 1:
        del grads['b.bias']                                          # step 2: rank 1 lost a gradient
    raised = False
    try:
        D.all_reduce_mean_(grads)
    except RuntimeError as e:
        raised = 'different gradient sets' in str(e)
    flags = [None] * world
    dist.all_gather_object(flags, (ok1, raised))
    if rank == 0:
        torch.save(flags, out)
    dist.destroy_process_group()


def test_gradient_set_divergence_after_step_one_raises_on_every_rank(tmp_path):
    """the gradient-set check is a collective EVERY rank issues on EVERY call.  A set that diverges
    after a verified first step (a parameter without gradient on one rank) must raise on both ranks -- not hang, not
    reduce mismatched buckets -- which a rank-local 'already verified' skip could not guarantee."""
    out = str(tmp_path / 'd.pt')
    mp.spawn(_diverging_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert torch.load(out) == [(True, True), (True, True)]


def test_gradient_all_reduce_world2(tmp_path):
    """data-parallel training: the bucketed gradient average equals the mean of the per-rank gradients."""
    out = str(tmp_path / 'g.pt')
    mp.spawn(_grad_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r['ok'] and r['nbytes'] == 4 * r['numel']


def test_committed_bench_lines_follow_the_contract():
    """the four bench lines committed under profiles/r02, r05 and r06 carry every field the bench contract names (they
    are the output of bench.py on the GPU box; this guards the schema against drift).  From round 6 on: no `frac` above 1
    anywhere in a line (VERDICT r05 weak #6a: a fraction above 1 means the byte count does not describe what the kernel
    does), the feature line is judged against HBM (SURVEY 8d's max() rule) and carries the fp16 single-pass peer value."""
    import json

    def fracs(o, path=''):
        if isinstance(o, dict):
            for k, v in o.items():
                if k in ('frac', 'frac_max_rule', 'hbm_frac', 'mfma_frac') and isinstance(v, (int, float)):
                    yield path + '/' + k, v
                else:
                    yield from fracs(v, path + '/' + str(k))
        elif isinstance(o, list):
            for i, v in enumerate(o):
                yield from fracs(v, path + '/%d' % i)
    for wl in ('hr', 'lr', 'hr_cond', 'feature'):
        line = json.load(open(os.path.join(ROOT, 'profiles', 'r06', 'bench_r06_%s.json' % wl)))
        bad = [(p_, v) for p_, v in fracs(line) if v > 1.0]
        assert not bad, (wl, bad)
        assert 'bound_rule' in line['roofline'] or wl == 'lr'
        assert 'unattributed_ms_per_step' in line['roofline_tail'] and line['roofline_tail']['unattributed_ms_per_step'] < 2.0
    feat = json.load(open(os.path.join(ROOT, 'profiles', 'r06', 'bench_r06_feature.json')))
    assert feat['roofline']['bound'] == 'hbm' and feat['roofline']['unit'] == 'GB/s' and feat['value_fp16_single_pass'] > feat['value']
    hc = json.load(open(os.path.join(ROOT, 'profiles', 'r06', 'bench_r06_hr_cond.json')))
    assert 'N = 108504' in hc['roofline']['traffic_layer']          # its own shapes, not the B = 8 layers
    for rnd, wl in [(r_, w_) for r_ in ('r02', 'r05', 'r06') for w_ in ('hr', 'lr', 'hr_cond', 'feature')]:
        line = json.load(open(os.path.join(ROOT, 'profiles', rnd, 'bench_%s_%s.json' % (rnd, wl))))
        for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                  'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
            assert k in line, (wl, k)
        assert line['n_gpus'] == 1 and line['higher_is_better'] is True and line['scaling'] == 'weak'
        assert line['vs_baseline'] is None and line['data'] == 'synthetic' and 'workload' in line['config']
        assert abs(line['value'] - 1e3 / line['ms_per_step']) <= 1e-6 * line['value']
        r = line['roofline']
        for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
            assert k in r, (wl, k)
        assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
        c = line['cpu_baseline']
        for k in ('value', 'unit', 'cores', 'kind', 'sample'):
            assert k in c, (wl, k)
        assert c['kind'] in ('port', 'reference')


def _gen_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from octfusion_amd import dist as D, generate as G
    D.init(backend='gloo')
    # every rank but 0 starts from its own constructor weights; prepare() must leave rank 0's U-Net AND VAE everywhere
    torch.manual_seed(7 + rank)
    net, vae, nbytes = G.prepare('snet_uncond', rank, torch.device('cpu'))

    def digest(m):
        return torch.stack([p.detach().double().sum() for p in m.parameters()] +
                           [p.detach().double().abs().sum() for p in m.parameters()])
    dig = torch.cat([digest(net), digest(vae)])
    gathered = [torch.zeros_like(dig) for _ in range(world)]
    dist.all_gather(gathered, dig)
    groups = G.plan(11, rank, world, 3)
    all_groups = [None] * world
    dist.all_gather_object(all_groups, groups)
    D.barrier()
    if rank == 0:
        n_par = sum(p.numel() for p in net.parameters()) + sum(p.numel() for p in vae.parameters())
        n_buf = sum(b.numel() for m in (net, vae) for b in m.buffers() if b.is_floating_point())
        torch.save({'same': all(torch.equal(gathered[0], g) for g in gathered), 'nbytes': nbytes,
                    'numel': n_par + n_buf, 'groups': all_groups, 'nonzero': float(dig.abs().sum()) > 0}, out)
    dist.destroy_process_group()


def test_generate_prepare_world2(tmp_path):
    """The generate driver on two ranks (gloo): U-Net and VAE weights of rank 0 reach every rank in the one broadcast
    per model, and the ranks' index groups are the reference's rank-strided result indices (train.py:168), batched."""
    out = str(tmp_path / 'g.pt')
    mp.spawn(_gen_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r['same'] and r['nonzero']
    assert r['nbytes'] == 4 * r['numel']
    assert r['groups'] == [[[0, 2, 4], [6, 8, 10]], [[1, 3, 5], [7, 9]]]


def test_generate_samples_with_ema_weights(tmp_path):
    """A checkpoint whose df_* and ema_df_* sets differ: the sampling net must hold the EMA set (train.py:181,
    octfusion_model_union.py:319,391)."""
    sys.path.insert(0, ROOT)
    from octfusion_amd import checkpoint, configs, generate as G
    from octfusion_amd.graph_unet_union import UNet3DModel
    torch.manual_seed(1)
    df = UNet3DModel(**configs.unet_params('snet_uncond', 'hr'))
    ema = UNet3DModel(**configs.unet_params('snet_uncond', 'hr'))
    with torch.no_grad():
        for p in ema.parameters():
            p.add_(0.25)
    path = str(tmp_path / 'df_steps-latest.pth')
    checkpoint.save_ckpt(path, df, ema, global_step=3, stage_flag='hr')
    net, _ = G.build_models('snet_uncond', ckpt=path, with_vae=False)
    want, have, other = ema.state_dict(), net.state_dict(), df.state_dict()
    assert list(want) == list(have)
    assert all(torch.equal(want[k], have[k]) for k in want)
    assert not all(torch.equal(other[k], have[k]) for k in want)


def test_checkpoint_files_are_read_with_the_restricted_unpickler(tmp_path):
    """Checkpoints and sample files are data from elsewhere: the reference's formats hold tensors, numbers and plain
    containers (octfusion_model_union.py:501-523, model_utils.py:18-28, gen_split.py:50-54), so they load under
    torch's weights-only unpickler -- including the optimiser dict a reference-trained file carries; a file that smuggles
    another Python object is refused unless the caller says it trusts it (allow_pickle=True)."""
    import pickle
    import pytest
    sys.path.insert(0, ROOT)
    from octfusion_amd import checkpoint
    lin = torch.nn.Linear(4, 3)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.unet_lr = torch.nn.Linear(4, 3)
    df, ema = Net(), Net()
    adam = torch.optim.Adam(df.parameters())
    with torch.enable_grad():                       # (GPU test modules switch autograd off process-wide at import)
        df.unet_lr(torch.ones(2, 4)).sum().backward()
    adam.step()
    good = {'df_unet_lr': df.unet_lr.state_dict(), 'ema_df_unet_lr': ema.unet_lr.state_dict(), 'opt': adam.state_dict(),
            'global_step': 7}
    p_good = str(tmp_path / 'df_good.pth')
    torch.save(good, p_good)
    fresh = Net()
    assert checkpoint.load_ckpt(p_good, fresh, fresh) == 7
    assert torch.equal(fresh.unet_lr.weight, ema.unet_lr.weight)

    class Payload:
        def __reduce__(self):
            return (print, ('executed while unpickling',))
    p_bad = str(tmp_path / 'df_bad.pth')
    torch.save(dict(good, extra=Payload()), p_bad)
    with pytest.raises(pickle.UnpicklingError):
        checkpoint.load_ckpt(p_bad, Net(), None)
    assert checkpoint.load_ckpt(p_bad, Net(), None, allow_pickle=True) == 7
    # VAE layouts and sample files
    for name, obj in (('vae.pth', lin.state_dict()), ('vae_wrapped.pth', {'autoencoder': lin.state_dict()}),
                      ('vae.solver.tar', {'model_dict': lin.state_dict(), 'epoch': 3})):
        path = str(tmp_path / name)
        torch.save(obj, path)
        sd = checkpoint.vae_state_dict(path)
        assert torch.equal(sd['weight'], lin.weight)
    d = tmp_path / 'sample'
    d.mkdir()
    torch.save(torch.ones(8, 16, 16, 16), str(d / 'split_small.pth'))
    small, large = checkpoint.read_splits(str(d), 'cpu')
    assert small.shape == (1, 8, 16, 16, 16) and large is None


def test_step_noise_stream_semantics():
    """pipeline.StepNoise: the lazily drawn per-step noise of a sparse x0 stage is the same sequence an up-front draw
    gives -- including the draws of steps that never ask for noise -- and refuses out-of-order or repeated access
    instead of silently handing out a different sample."""
    import pytest
    sys.path.insert(0, ROOT)
    from octfusion_amd.pipeline import StepNoise
    g = torch.Generator().manual_seed(5)
    eager = [torch.randn(7, 3, generator=g) for _ in range(9)]
    g2 = torch.Generator().manual_seed(5)
    lazy = StepNoise(lambda: torch.randn(7, 3, generator=g2), 9)
    assert len(lazy) == 9
    for i in (0, 1, 4, 5, 8):                      # steps 2, 3, 6, 7 use no noise: their draws are consumed anyway
        assert torch.equal(lazy[i], eager[i])
    with pytest.raises(IndexError):
        lazy[8]                                    # asked for twice
    # finish(): the draws nobody asked for are consumed, so that the stream continues where an up-front draw leaves it
    g3 = torch.Generator().manual_seed(5)
    lazy3 = StepNoise(lambda: torch.randn(7, 3, generator=g3), 9)
    lazy3[2]
    lazy3.finish()
    assert torch.equal(torch.randn(7, 3, generator=g3), torch.randn(7, 3, generator=g))   # both streams: 9 draws behind them
    lazy2 = StepNoise(lambda: torch.zeros(1), 3)
    lazy2[1]
    with pytest.raises(IndexError):
        lazy2[0]                                   # out of order
    with pytest.raises(IndexError):
        lazy2[3]
