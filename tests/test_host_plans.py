"""Host-side planning logic that needs no GPU: the aux-row plan of the GroupNorm launch (dual_octree.DualOctree.aux_plan:
which 64-row block writes which multi-neighbour mean row, include/ofx.h ofx_gn_apply_planes) on synthetic CSR graphs,
and the quota-aware CPU baseline sizing of bench.py."""
import random
import types

import torch


def _stub(seg_ptr, col, multi_seg, N):
    from octfusion_amd.dual_octree import DualOctree
    s = types.SimpleNamespace()
    s._ext = {}
    s.csr = lambda d: (seg_ptr, col, N, int(col.numel()))
    s.ext = lambda d: (None, multi_seg, int(multi_seg.numel()))
    s.aux_plan = lambda d, rows_per_block=64: DualOctree.aux_plan(s, d, rows_per_block)
    s.oct_plan = lambda d, shift=0: DualOctree.oct_plan(s, d, shift)
    s.batch_id32 = lambda d: (torch.arange(N, dtype=torch.int32) * 3) // max(N, 1)
    return s


def _random_graph(N, seed):
    rnd = random.Random(seed)
    seg_ptr, col, multi = [0], [], []
    for s in range(N * 7):
        k = rnd.choice([0, 1, 1, 1, 2, 4, 4, 5, 9])
        if k > 1 and rnd.random() < 0.8:                       # siblings: consecutive rows of one aligned group of eight
            base = rnd.randrange(0, max(1, N // 8)) * 8
            rows = [min(N - 1, base + j) for j in rnd.sample(range(8), min(k, 8))]
        else:
            rows = [rnd.randrange(N) for _ in range(k)]
        col += sorted(rows)
        seg_ptr.append(len(col))
        if k > 1:
            multi.append(s)
    return (torch.tensor(seg_ptr, dtype=torch.int32), torch.tensor(col, dtype=torch.int32),
            torch.tensor(multi, dtype=torch.int32))


def test_aux_plan_partitions_the_aux_rows_by_owner_block():
    for N, seed in ((1, 0), (70, 1), (700, 2), (1000, 3)):
        seg_ptr, col, multi_seg = _random_graph(N, seed)
        V = int(multi_seg.numel())
        if V == 0:
            continue
        plan, n_left = _stub(seg_ptr, col, multi_seg, N).aux_plan(5)
        plan = plan.tolist()
        mb = (N + 63) // 64
        ptr = plan[:mb + 1]
        n_own = ptr[mb]
        owned = plan[mb + 1:mb + 1 + n_own]
        assert plan[mb + 1 + n_own] == n_left
        left = plan[mb + 2 + n_own:]
        assert len(left) == n_left and left[0] == 0                       # the zero row is always a leftover
        assert ptr[0] == 0 and all(a <= b for a, b in zip(ptr, ptr[1:]))
        assert sorted(owned + left) == list(range(V + 1))                 # every aux row exactly once
        for blk in range(mb):
            for v in owned[ptr[blk]:ptr[blk + 1]]:
                s = int(multi_seg[v - 1])
                srcs = col[int(seg_ptr[s]):int(seg_ptr[s + 1])].tolist()
                assert all(r // 64 == blk for r in srcs), (blk, v, srcs)  # all sources inside the owner's 64 rows
        for v in left[1:]:
            s = int(multi_seg[v - 1])
            srcs = col[int(seg_ptr[s]):int(seg_ptr[s + 1])].tolist()
            assert len({r // 64 for r in srcs}) > 1                       # leftovers really span blocks
    # no multi-neighbour segment at all: only the zero row, as a leftover
    seg_ptr = torch.arange(0, 7 * 10 + 1, dtype=torch.int32)
    col = torch.zeros(70, dtype=torch.int32)
    plan, n_left = _stub(seg_ptr, col, torch.zeros(0, dtype=torch.int32), 10).aux_plan(4)
    assert n_left == 1 and plan.tolist() == [0, 0, 1, 0]


def test_oct_plan_gives_every_aux_row_to_one_octet_or_to_the_leftovers():
    """dual_octree.DualOctree.oct_plan (include/ofx.h ofx_gn_apply_planes_oct): an entry (v, mask) of octet o means that
    the sources of aux row v are exactly the rows 8 o - shift + j of the set bits j; every other aux row is a leftover
    whose head points at its CSR segment, flattened."""
    for N, seed, shift in ((1, 0, 0), (70, 1, 0), (70, 1, 3), (700, 2, 5), (1000, 3, 7), (1003, 4, 2)):
        seg_ptr, col, multi_seg = _random_graph(N, seed)
        V = int(multi_seg.numel())
        if V == 0:
            continue
        st = _stub(seg_ptr, col, multi_seg, N)
        plan, sh, n_own, n_left, (o_ptr, o_ent, o_head, o_src) = st.oct_plan(5, shift)
        assert sh == shift and o_ent % 2 == 0 and o_head % 4 == 0 and n_own + n_left == V + 1
        plan = plan.tolist()
        n_oct = (N + shift + 7) // 8
        ptr = plan[o_ptr:o_ptr + n_oct + 1]
        assert ptr[0] == 0 and ptr[-1] == n_own and all(a <= b for a, b in zip(ptr, ptr[1:]))
        ent = plan[o_ent:o_ent + 2 * n_own]
        head = [plan[o_head + 4 * i:o_head + 4 * i + 4] for i in range(n_left)]
        assert o_src == o_head + 4 * n_left and head[0] == [0, 0, 0, 0]
        assert sorted(ent[0::2] + [h[0] for h in head]) == list(range(V + 1))     # every aux row exactly once
        for o in range(n_oct):
            for i in range(ptr[o], ptr[o + 1]):
                v, mask = ent[2 * i], ent[2 * i + 1]
                s = int(multi_seg[v - 1])
                srcs = col[int(seg_ptr[s]):int(seg_ptr[s + 1])].tolist()
                assert 0 < mask < 256
                assert sorted(srcs) == [8 * o - shift + j for j in range(8) if mask >> j & 1], (o, v, mask, srcs)
        bid = st.batch_id32(5).tolist()
        nxt = 0
        for v, start, cnt, b in head[1:]:
            s = int(multi_seg[v - 1])
            srcs = col[int(seg_ptr[s]):int(seg_ptr[s + 1])].tolist()
            assert start == nxt and cnt == len(srcs) and plan[o_src + start:o_src + start + cnt] == srcs
            assert b == bid[s // 7]
            assert len({(r + shift) // 8 for r in srcs}) > 1 or len(set(srcs)) < len(srcs)
            nxt += cnt
        assert len(plan) == o_src + nxt + 1
    # no multi-neighbour segment: only the zero row, as a leftover
    seg_ptr = torch.arange(0, 7 * 10 + 1, dtype=torch.int32)
    col = torch.zeros(70, dtype=torch.int32)
    plan, sh, n_own, n_left, (o_ptr, o_ent, o_head, o_src) = _stub(seg_ptr, col, torch.zeros(0, dtype=torch.int32), 10).oct_plan(4, 0)
    assert n_own == 0 and n_left == 1 and not plan.any() and len(plan) > o_src


def test_cpu_baseline_sizes_itself_inside_the_cgroup_quota(tmp_path, monkeypatch):
    import bench
    q = bench.cgroup_cpu_quota()
    assert q is None or q > 0
    # the parsing of both cgroup layouts
    import builtins
    real_open = builtins.open
    files = {'/sys/fs/cgroup/cpu.max': '1600000 100000\n'}

    def fake_open(path, *a, **k):
        if path in files:
            p = tmp_path / 'f'
            p.write_text(files[path])
            return real_open(p, *a, **k)
        if str(path).startswith('/sys/fs/cgroup/'):
            raise OSError
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, 'open', fake_open)
    assert bench.cgroup_cpu_quota() == 16.0
    files['/sys/fs/cgroup/cpu.max'] = 'max 100000\n'
    assert bench.cgroup_cpu_quota() is None
    del files['/sys/fs/cgroup/cpu.max']
    files['/sys/fs/cgroup/cpu/cpu.cfs_quota_us'] = '400000\n'
    files['/sys/fs/cgroup/cpu/cpu.cfs_period_us'] = '100000\n'
    assert bench.cgroup_cpu_quota() == 4.0


def test_whole_box_cpu_workers_orchestration(monkeypatch):
    """bench.cpu_baseline's multi-process figure (pinned workers, file rendezvous, summed rates) on this host, with
    4-thread workers so that two of them fit: the dense lr stage, 1 warm-up + 2 timed steps each."""
    import os
    import bench
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    q = bench.cgroup_cpu_quota()
    if q is not None:
        usable = min(usable, int(q + 0.5))
    if usable < 8:
        import pytest
        pytest.skip('needs 8 usable CPUs for two 4-thread workers')
    monkeypatch.setenv('OFX_CPU_WORKER_THREADS', '4')
    monkeypatch.setitem(bench.WORKLOADS['lr'], 'cpu', (2, 3))
    r = bench.cpu_baseline('lr', 4)
    assert r['usable_cpus'] == usable and r['threads_used'] == min(32, usable) and r['single_process']['value'] > 0
    wb = r['whole_box']
    assert wb['processes'] == min(8, usable // 4) and wb['threads_per_process'] == 4 and wb['value'] > 0
    assert len(wb['timed_s_per_worker']) == wb['processes'] and wb['common_window_s'] > 0
    assert r['value'] >= max(r['single_process']['value'], 0) * 0.999 or r['value'] == wb['value']


def _cpu_octree(depth, fd, seed):
    """A one-element octree on CPU tensors with random splits below the full layer (only the arrays batch_slices /
    merge_octrees touch: keys, children, nnum, nnum_nempty)."""
    from octfusion_amd.octree import Octree
    g = torch.Generator().manual_seed(seed)
    oc = Octree.__new__(Octree)
    oc.depth, oc.full_depth, oc.batch_size, oc.device = depth, fd, 1, torch.device('cpu')
    oc.keys, oc.children = [], []
    oc.nnum, oc.nnum_nempty = torch.zeros(depth + 1, dtype=torch.int64), torch.zeros(depth + 1, dtype=torch.int64)
    keys = torch.zeros(1, dtype=torch.int64)
    for d in range(depth + 1):
        if d > 0:
            parents = oc.keys[d - 1][oc.children[d - 1] >= 0]
            keys = (parents[:, None] * 8 + torch.arange(8)[None, :]).reshape(-1)
        split = torch.ones(keys.numel(), dtype=torch.bool) if d < fd else torch.rand(keys.numel(), generator=g) < 0.4
        if d == depth:
            split[:] = False
        child = torch.full((keys.numel(),), -1, dtype=torch.int32)
        child[split] = torch.arange(int(split.sum()), dtype=torch.int32)
        oc.keys.append(keys)
        oc.children.append(child)
        oc.nnum[d], oc.nnum_nempty[d] = keys.numel(), int(split.sum())
    return oc


def test_batch_slices_inverts_merge_octrees(monkeypatch):
    """Octree.batch_slices (the lanes of sampler.sample_loop) on CPU tensors: slicing the merge of four one-element octrees
    gives back the elements / the merges of runs of them, keys, child pointers and counts."""
    from octfusion_amd import _lib
    from octfusion_amd.octree import merge_octrees
    monkeypatch.setattr(_lib, 'require_device', lambda: None)
    elems = [_cpu_octree(4, 1, s) for s in (3, 4, 5, 6)]
    elems[2] = _cpu_octree(4, 1, 99)
    elems[2].children[1][:] = -1                     # an element with nothing below the full layer
    elems[2].nnum_nempty[1] = 0
    for d in (2, 3, 4):
        elems[2].keys[d], elems[2].children[d] = elems[2].keys[d][:0], elems[2].children[d][:0]
        elems[2].nnum[d] = elems[2].nnum_nempty[d] = 0
    whole = merge_octrees(elems)
    for bounds in ([0, 1, 2, 3, 4], [0, 2, 4], [0, 3, 4], [1, 3]):
        got = whole.batch_slices(bounds)
        assert len(got) == len(bounds) - 1
        for oc, b0, b1 in zip(got, bounds, bounds[1:]):
            want = merge_octrees(elems[b0:b1])
            assert oc.batch_size == b1 - b0
            assert torch.equal(oc.nnum, want.nnum) and torch.equal(oc.nnum_nempty, want.nnum_nempty)
            for d in range(5):
                assert torch.equal(oc.keys[d], want.keys[d]) and torch.equal(oc.children[d], want.children[d]), (bounds, b0, d)
    import pytest
    with pytest.raises(ValueError):
        whole.batch_slices([0, 5])
    with pytest.raises(ValueError):
        whole.batch_slices([2, 2])


def test_lane_count_rules(monkeypatch):
    from octfusion_amd import sampler
    doc = types.SimpleNamespace(split_batch=lambda n: None)
    monkeypatch.setattr(sampler, 'LANES', 1)
    assert sampler.lane_count(8, doc, True) == 1                      # the default: one stream
    monkeypatch.setattr(sampler, 'LANES', 2)
    assert sampler.lane_count(8, doc, True) == 2
    assert sampler.lane_count(8, doc, False) == 1                     # eager launches are host-bound: nothing to overlap
    assert sampler.lane_count(8, None, True) == 1                     # the dense lr stage
    assert sampler.lane_count(1, doc, True) == 1                      # one shape cannot be split
    assert sampler.lane_count(8, object(), True) == 1                 # a doctree that cannot split itself
    monkeypatch.setattr(sampler, 'LANES', 4)
    assert sampler.lane_count(3, doc, True) == 3
