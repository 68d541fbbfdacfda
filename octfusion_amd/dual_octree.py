"""DualOctree: the neighbour graph the graph convolutions run on.

Same constructor and attribute surface as the reference's
models/networks/dualoctree_networks/dual_octree.py::DualOctree:
``graph[d]['edge_idx'|'edge_dir'|'node_type'|'keyd'|'node_mask']``,
``batch_id(depth)``, ``node_child(d)``, ``nnum / lnum / ncum / nenum``,
``batch_size``, ``total_num``, ``octree``, ``device``, ``depth``, ``full_depth``,
``post_processing_for_docnn()``.

The build is a different algorithm from the reference's (see
csrc/ofx_graph.hip): two HIP passes (count, fill) per depth emit CSR segments
keyed by (row, dir) directly in the reference's sorted order.  The native CSR
(`csr(d)`), the node-type-fraction slab (`type_frac(d, nt)`), the int32 batch
ids and the pool / unpool row maps are cached here because the doctree is
shared read-only by all 200 denoising steps.  The reference's COO tensors
(int64 ``edge_idx`` / ``edge_dir``) are materialised lazily, only if someone
reads them.
"""
import ctypes

import torch

from . import _lib
from ._lib import call, ptr, stream


class _Graph(dict):
    """graph[d] with the reference's keys; COO views are built on first access."""

    def __init__(self, owner, d):
        super().__init__()
        self._owner = owner
        self._d = d

    def __missing__(self, key):
        if key in ('edge_idx', 'edge_dir'):
            self._owner._expand(self._d)
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return key in ('edge_idx', 'edge_dir') or dict.__contains__(self, key)


class DualOctree:
    def __init__(self, octree, prev=None):
        """prev: the DualOctree of the SAME octree before it was grown (VAE decoder, graph_vae.py:203-210, where the
        reference rebuilds everything).  Graph depth d only depends on the keys of depths <= d and the child pointers
        of depths < d, so every depth whose inputs are still the very same arrays is adopted from `prev` and only
        the new depths are built."""
        _lib.require_device()
        self.octree = octree
        self.device = octree.device
        self.depth = octree.depth
        self.full_depth = octree.full_depth
        self.batch_size = octree.batch_size
        dev = self.device
        depth, fd = self.depth, self.full_depth

        # node numbers (host, like the reference's octree.nnum tensors)
        self.nnum = octree.nnum[:depth + 1].clone()
        self.nenum = octree.nnum_nempty[:depth + 1].clone()
        self.ncum = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(self.nnum, 0)])
        self.lnum = self.nnum - self.nenum

        # depth-concatenated tree arrays (dual_octree.py:42-44)
        self.child = torch.cat([octree.children[d] for d in range(depth + 1)])
        self.key = torch.cat([octree.keys[d] for d in range(depth + 1)])
        total = int(self.ncum[depth + 1])
        L = _lib.lib()
        self._leafrank = torch.empty(total, dtype=torch.int32, device=dev)
        mx = int(self.nnum.max())
        ws = torch.empty(L.ofx_tree_leafrank_ws_bytes(mx), dtype=torch.uint8, device=dev)
        self._nnum_c = (ctypes.c_int64 * (depth + 1))(*[int(v) for v in self.nnum])
        self._nne_c = (ctypes.c_int64 * (depth + 1))(*[int(v) for v in self.nenum])
        call('ofx_tree_leafrank', ptr(self.child), ctypes.addressof(self._nnum_c), depth,
             ptr(self._leafrank), ptr(ws), stream())
        self._tree = _lib.OfxTree(depth, fd, self.batch_size, ptr(self.child), ptr(self.key),
                                  ptr(self._leafrank), ctypes.addressof(self._nnum_c),
                                  ctypes.addressof(self._nne_c))

        self.graph = [dict() for _ in range(depth + 1)]
        self._csr = {}
        self._nbr = {}
        self._ext = {}
        self._rev = {}
        self._bid32 = {}
        self._count = {}
        self._tf = {}
        self._maps = {}
        self._ntype8 = {}
        self.batch_id_dict = {}
        leaf_base = 0
        self._leaf_base = {}
        self._src = [(octree.keys[d], octree.children[d], octree.children[d]._version) for d in range(depth + 1)]
        reuse_to = self._reusable_depth(prev) if prev is not None else fd - 1
        self.adopted_depths = list(range(fd, reuse_to + 1))
        for d in range(fd, depth + 1):
            self._leaf_base[d] = leaf_base
            if d <= reuse_to:
                self._adopt(prev, d, unpool_ok=d < reuse_to)
            else:
                self._build_depth(d, leaf_base + int(self.nnum[d]))
            leaf_base += int(self.lnum[d])
        self.total_num = self.batch_id_dict[depth].shape[0]

    # ------------------------------------------------------------------
    def _reusable_depth(self, prev):
        """deepest graph depth of `prev` whose inputs are unchanged (identity of the per-depth arrays)."""
        if prev.full_depth != self.full_depth or prev.batch_size != self.batch_size or prev.device != self.device:
            return self.full_depth - 1
        top = -1
        for d in range(min(prev.depth, self.depth) + 1):
            k, c, v = self._src[d]
            pk, pc, pv = prev._src[d]
            if k is not pk or int(prev.nnum[d]) != int(self.nnum[d]):
                break
            top = d                                     # graph d needs keys[d] but not children[d]
            if c is not pc or v != pv:
                break
        return top if top >= self.full_depth else self.full_depth - 1

    def _adopt(self, prev, d, unpool_ok):
        for name in ('_csr', '_nbr', '_ext', '_bid32', '_ntype8', 'batch_id_dict', '_count'):
            getattr(self, name)[d] = getattr(prev, name)[d]
        for key, v in prev._ext.items():                # aux plans of the adopted depth (keyed ('aux_plan', d, rows) / ('oct_plan', d, shift))
            if isinstance(key, tuple) and key[0] in ('aux_plan', 'oct_plan') and key[1] == d:
                self._ext[key] = v
        if d in prev._rev:
            self._rev[d] = prev._rev[d]
        g = _Graph(self, d)
        for k, v in dict.items(prev.graph[d]):
            dict.__setitem__(g, k, v)
        self.graph[d] = g
        for key, v in prev._tf.items():
            if (key[1] if key[0] == 'tfp' else key[0]) == d:
                self._tf[key] = v
        for key, v in prev._maps.items():
            if key[1] == d and (key[0] in ('pad', 'pool') or (key[0] == 'unpool' and unpool_ok)):
                self._maps[key] = v

    def _build_depth(self, d, N):
        dev = self.device
        tree = ctypes.byref(self._tree)
        seg_cnt = torch.empty(N * 7, dtype=torch.int32, device=dev)
        call('ofx_graph_count', tree, d, ptr(seg_cnt), stream())
        seg_ptr = torch.empty(N * 7 + 1, dtype=torch.int32, device=dev)
        ws = torch.empty(_lib.lib().ofx_scan_ws_bytes(N * 7), dtype=torch.uint8, device=dev)
        call('ofx_scan_i32', ptr(seg_cnt), ptr(seg_ptr), N * 7, ptr(ws), stream())
        # the multi-neighbour ranks only need the segment sizes, so both totals (edges, multi-neighbour segments)
        # come back in ONE host read per depth -- they size the column array and the pre-averaged-row scratch
        flag = seg_cnt
        call('ofx_graph_multi_flag', ptr(seg_ptr), N, ptr(flag), stream())
        rank = torch.empty(N * 7 + 1, dtype=torch.int32, device=dev)
        call('ofx_scan_i32', ptr(flag), ptr(rank), N * 7, ptr(ws), stream())
        E, V = torch.stack([seg_ptr[-1], rank[-1]]).tolist()
        col = torch.empty(E, dtype=torch.int32, device=dev)
        call('ofx_graph_fill', tree, d, ptr(seg_ptr), ptr(col), stream())
        bid = torch.empty(N, dtype=torch.int32, device=dev)
        ntype = torch.empty(N, dtype=torch.uint8, device=dev)
        keyd = torch.empty(N, dtype=torch.int64, device=dev)
        nm_len = int(self.ncum[d] + self.nnum[d] - self.ncum[self.full_depth])
        nmask = torch.empty(nm_len, dtype=torch.uint8, device=dev)
        call('ofx_graph_nodes', tree, d, ptr(bid), ptr(ntype), ptr(keyd), ptr(nmask), stream())
        self._csr[d] = (seg_ptr, col, N, E)
        nbr = torch.empty(N * 7, dtype=torch.int32, device=dev)
        call('ofx_graph_primary', ptr(seg_ptr), ptr(col), N, ptr(nbr), stream())
        self._nbr[d] = nbr
        # + 4 entries: the persistent GraphConv fetches the table in 16-B chunks and may read 12 B past the end (ofx.h)
        nbr_ext = torch.zeros(N * 7 + 4, dtype=torch.int32, device=dev)[:N * 7]
        multi_seg = torch.empty(max(V, 1), dtype=torch.int32, device=dev)
        call('ofx_graph_primary_ext', ptr(seg_ptr), ptr(col), N, ptr(rank), ptr(nbr_ext), ptr(multi_seg), stream())
        self._ext[d] = (nbr_ext, multi_seg, V)
        self._bid32[d] = bid
        self._ntype8[d] = ntype
        self.batch_id_dict[d] = bid.to(torch.int64)
        g = _Graph(self, d)
        g['node_type'] = ntype.to(torch.int64)
        g['keyd'] = keyd
        g['node_mask'] = nmask.bool()
        self.graph[d] = g
        self._count[d] = torch.bincount(bid, minlength=self.batch_size).to(torch.float32)

    def _expand(self, d):
        seg_ptr, col, N, E = self._csr[d]
        row = torch.empty(E, dtype=torch.int64, device=self.device)
        c64 = torch.empty(E, dtype=torch.int64, device=self.device)
        edir = torch.empty(E, dtype=torch.int64, device=self.device)
        call('ofx_graph_expand', ptr(seg_ptr), N, ptr(col), ptr(row), ptr(c64), ptr(edir), stream())
        dict.__setitem__(self.graph[d], 'edge_idx', torch.stack([row, c64]))
        dict.__setitem__(self.graph[d], 'edge_dir', edir)

    # ---- reference API --------------------------------------------------
    def post_processing_for_docnn(self):
        """No-op: self loops, compact numbering, node attributes and edge order are
        produced by the build itself (dual_octree.py:400-409)."""
        return self

    def batch_id(self, depth, nempty=False):
        return self.batch_id_dict[depth]

    def split_batch(self, parts):
        """The batch as `parts` independent dual octrees of consecutive elements: [(DualOctree, rows, (b0, b1))] with
        `rows` the int64 indices of the part's rows among this doctree's rows at the finest graph depth (a row tensor
        x of the whole batch is x[rows] for the part, in the part's own row order: every depth segment of the graph is
        batch-sorted, so a part keeps the relative order).  Shapes of a batch share nothing in the network (GroupNorm
        statistics are per element, graphs never cross elements), so a step over the whole batch equals the steps over
        the parts; sampler.sample_loop runs them as lanes on separate HIP streams."""
        B = self.batch_size
        parts = max(1, min(int(parts), B))
        bounds = [B * p // parts for p in range(parts + 1)]
        bid = self.batch_id(self.depth)
        out = []
        for oc, b0, b1 in zip(self.octree.batch_slices(bounds), bounds, bounds[1:]):
            rows = torch.nonzero((bid >= b0) & (bid < b1)).squeeze(1)
            sub = DualOctree(oc)
            if sub.total_num != rows.numel():
                raise _lib.OfxError('split_batch: part %d..%d has %d graph rows, the batch holds %d of them'
                                    % (b0, b1, sub.total_num, rows.numel()))
            out.append((sub, rows, (b0, b1)))
        return out

    def node_child(self, depth):
        s = int(self.ncum[depth])
        return self.child[s: s + int(self.nnum[depth])]

    # ---- native handles used by octfusion_amd.modules ---------------------
    def csr(self, d):
        """(seg_ptr int32 [N*7+1], col int32 [E], N, E)."""
        return self._csr[d]

    def nbr(self, d):
        """int32 [N*7]: the single neighbour of segment (row, dir), -1 none, -2 several (see csr)."""
        return self._nbr[d]

    def node_type8(self, d):
        """uint8 [N_d]: node type of every row of graph depth d (the one-hot class of modules.py:199-203)."""
        return self._ntype8[d]

    def ext(self, d):
        """(nbr_ext int32 [N*7], multi_seg int32 [V], V): the branch-free gather table (ofx.h)."""
        return self._ext[d]

    def aux_plan(self, d, rows_per_block=None):
        """(plan int32, leftover count) for ofx_gn_apply_planes: which 64-row block of the GroupNorm launch writes which
        aux row of graph depth d (include/ofx.h).  An aux row = the mean over a multi-neighbour segment; its sources are
        the finer neighbours across one face of a coarse leaf -- siblings, i.e. rows of one aligned group of eight -- so
        almost every aux row has all its sources inside one block, which then writes it from cache right after its own
        rows.  Built once per doctree depth (a few torch index ops, no host sync beyond the sizes)."""
        if rows_per_block is None:
            rows_per_block = _lib.lib().ofx_gn_apply_rows()          # the kernel's own block size (never assumed)
        key = ('aux_plan', d, rows_per_block)
        if key in self._ext:
            return self._ext[key]
        seg_ptr, col, N, E = self.csr(d)
        _, multi_seg, V = self.ext(d)
        dev = seg_ptr.device
        mb = (N + rows_per_block - 1) // rows_per_block
        if V == 0:
            plan = torch.cat([torch.zeros(mb + 1, dtype=torch.int32, device=dev),
                              torch.tensor([1, 0], dtype=torch.int32, device=dev)])
            self._ext[key] = (plan, 1)
            return self._ext[key]
        ms = multi_seg[:V].long()
        start, end = seg_ptr[ms].long(), seg_ptr[ms + 1].long()
        lens = end - start
        seg_of_edge = torch.repeat_interleave(torch.arange(V, device=dev), lens)
        edge = torch.repeat_interleave(start - torch.cumsum(lens, 0) + lens, lens) + torch.arange(int(lens.sum()), device=dev)
        src_blk = col[edge].long() // rows_per_block
        lo = torch.full((V,), mb, dtype=torch.long, device=dev).scatter_reduce_(0, seg_of_edge, src_blk, 'amin')
        hi = torch.full((V,), -1, dtype=torch.long, device=dev).scatter_reduce_(0, seg_of_edge, src_blk, 'amax')
        owned = lo == hi
        ids = torch.arange(1, V + 1, device=dev)
        own_ids, own_blk = ids[owned], lo[owned]
        order = torch.argsort(own_blk, stable=True)
        counts = torch.bincount(own_blk, minlength=mb)
        ptr_ = torch.zeros(mb + 1, dtype=torch.long, device=dev)
        ptr_[1:] = torch.cumsum(counts, 0)
        left = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), ids[~owned]])
        plan = torch.cat([ptr_, own_ids[order], torch.tensor([left.numel()], device=dev), left]).to(torch.int32)
        self._ext[key] = (plan, int(left.numel()))
        return self._ext[key]

    def oct_plan(self, d, shift=None):
        """Aux-row plan of ofx_gn_apply_planes_oct (include/ofx.h) for graph depth d: (plan int32, shift, n_own, n_left,
        offsets (ptr, ent, left_head, left_src) into plan).

        The rows of a graph depth are [coarse leaves | all nodes of depth d]; the depth-d nodes come in sibling groups of
        eight.  `shift` pads the leaf prefix to a multiple of eight, so that "octet" o = rows 8 o - shift .. 8 o - shift + 7
        of the depth-d part IS a sibling group.  An aux row (multi-neighbour segment) whose sources all lie in one octet
        -- the four finer neighbours across a face of a coarse leaf are siblings: 83 % of the aux rows of the shell trees
        at depth 6-8, all of them at depth 5 -- is owned by that octet: entry (aux row id, 8-bit mask of the octet's rows).
        Every other aux row is a leftover: head (aux row id, first slot in the flat source list, sources, batch element).
        Layout: ptr [n_oct + 1] | pad | ent [n_own][2] | pad | left_head [n_left][4] (the zero row first) | left_src."""
        if shift is None:
            shift = (8 - self._leaf_base[d] % 8) % 8
        key = ('oct_plan', d, shift)
        if key in self._ext:
            return self._ext[key]
        seg_ptr, col, N, E = self.csr(d)
        _, multi_seg, V = self.ext(d)
        dev = seg_ptr.device
        n_oct = (N + shift + 7) // 8
        ent_off = (n_oct + 1 + 1) & ~1
        if V == 0:
            head_off = (ent_off + 3) & ~3
            plan = torch.zeros(head_off + 4 + 1, dtype=torch.int32, device=dev)
            self._ext[key] = (plan, shift, 0, 1, (0, ent_off, head_off, head_off + 4))
            return self._ext[key]
        ms = multi_seg[:V].long()
        start, end = seg_ptr[ms].long(), seg_ptr[ms + 1].long()
        lens = end - start
        seg_of_edge = torch.repeat_interleave(torch.arange(V, device=dev), lens)
        edge = torch.repeat_interleave(start - torch.cumsum(lens, 0) + lens, lens) + torch.arange(int(lens.sum()), device=dev)
        src = col[edge].long() + shift
        oc = src >> 3
        lo = torch.full((V,), n_oct, dtype=torch.long, device=dev).scatter_reduce_(0, seg_of_edge, oc, 'amin')
        hi = torch.full((V,), -1, dtype=torch.long, device=dev).scatter_reduce_(0, seg_of_edge, oc, 'amax')
        bits = torch.zeros(V, dtype=torch.long, device=dev).scatter_add_(0, seg_of_edge, torch.ones_like(src) << (src & 7))
        # (the sources of a segment are distinct rows: inside one octet their bits add up to the mask; a repeated source
        # would carry into the next bit and is sent to the leftovers)
        pop = torch.zeros_like(bits)
        for j in range(8):
            pop += (bits >> j) & 1
        owned = (lo == hi) & (bits < 256) & (pop == lens)
        ids = torch.arange(1, V + 1, device=dev)
        own_ids, own_oct, own_bits = ids[owned], lo[owned], bits[owned]
        order = torch.argsort(own_oct, stable=True)
        ptr_ = torch.zeros(n_oct + 1, dtype=torch.long, device=dev)
        ptr_[1:] = torch.cumsum(torch.bincount(own_oct, minlength=n_oct), 0)
        ent = torch.stack([own_ids[order], own_bits[order]], dim=1).reshape(-1)
        n_own = int(own_ids.numel())
        # leftovers: the CSR segments flattened, so that the kernel's chain is head -> source -> row
        lv = ~owned
        l_len = lens[lv]
        l_start = torch.cumsum(l_len, 0) - l_len
        l_bid = self.batch_id32(d)[ms[lv] // 7].long()        # (every source of a segment lies in the batch element of its row)
        head = torch.cat([torch.zeros(1, 4, dtype=torch.long, device=dev),
                          torch.stack([ids[lv], l_start, l_len, l_bid], dim=1)]).reshape(-1)
        l_src = col[edge[lv[seg_of_edge]]].long()
        n_left = int(l_len.numel()) + 1
        head_off = (ent_off + 2 * n_own + 3) & ~3
        pad0 = torch.zeros(ent_off - (n_oct + 1), dtype=torch.long, device=dev)
        pad1 = torch.zeros(head_off - (ent_off + 2 * n_own), dtype=torch.long, device=dev)
        tail = torch.zeros(1, dtype=torch.long, device=dev)              # (left_src is never empty: slot 0 of the zero row)
        plan = torch.cat([ptr_, pad0, ent, pad1, head, l_src, tail]).to(torch.int32)
        self._ext[key] = (plan, shift, n_own, n_left, (0, ent_off, head_off, head_off + 4 * n_left))
        return self._ext[key]

    def rev(self, d):
        """Reverse graph of depth d for GraphConv's backward pass (ofx.h): dict with rev_ptr [N*7+1], rev_row [E],
        rev_w [E] (1 / size of the forward segment the edge came from), nbr (primary table), nbr_ext, multi_seg, V.
        Built on first use."""
        if d in self._rev:
            return self._rev[d]
        seg_ptr, col, N, E = self._csr[d]
        dev = self.device
        cnt = torch.empty(N * 7, dtype=torch.int32, device=dev)
        call('ofx_graph_reverse_count', ptr(seg_ptr), ptr(col), N, ptr(cnt), stream())
        rev_ptr = torch.empty(N * 7 + 1, dtype=torch.int32, device=dev)
        ws = torch.empty(_lib.lib().ofx_scan_ws_bytes(N * 7), dtype=torch.uint8, device=dev)
        call('ofx_scan_i32', ptr(cnt), ptr(rev_ptr), N * 7, ptr(ws), stream())
        rev_row = torch.empty(E, dtype=torch.int32, device=dev)
        rev_w = torch.empty(E, dtype=torch.float32, device=dev)
        call('ofx_graph_reverse_fill', ptr(seg_ptr), ptr(col), N, ptr(rev_ptr), ptr(cnt), ptr(rev_row), ptr(rev_w),
             stream())
        nbr = torch.empty(N * 7, dtype=torch.int32, device=dev)
        call('ofx_graph_primary_w', ptr(rev_ptr), ptr(rev_row), ptr(rev_w), N, ptr(nbr), stream())
        flag = cnt
        call('ofx_graph_multi_flag_w', ptr(rev_ptr), ptr(rev_w), N, ptr(flag), stream())
        rank = torch.empty(N * 7 + 1, dtype=torch.int32, device=dev)
        call('ofx_scan_i32', ptr(flag), ptr(rank), N * 7, ptr(ws), stream())
        V = int(rank[-1].item())
        nbr_ext = torch.empty(N * 7, dtype=torch.int32, device=dev)
        multi_seg = torch.empty(max(V, 1), dtype=torch.int32, device=dev)
        call('ofx_graph_primary_ext_w', ptr(rev_ptr), ptr(rev_row), ptr(rev_w), N, ptr(rank), ptr(nbr_ext),
             ptr(multi_seg), stream())
        self._rev[d] = dict(rev_ptr=rev_ptr, rev_row=rev_row, rev_w=rev_w, nbr=nbr, nbr_ext=nbr_ext,
                            multi_seg=multi_seg, V=V, N=N, E=E)
        return self._rev[d]

    def batch_id32(self, d):
        return self._bid32[d]

    def count(self, d):
        """nodes per batch element at graph depth d (fp32 [B])."""
        return self._count[d]

    def type_frac(self, d, nt):
        """[N_d, pad32(7*nt)] fp32: per (row, dir) fraction of neighbours of each node type."""
        key = (d, nt)
        if key not in self._tf:
            seg_ptr, col, N, E = self._csr[d]
            ld = (7 * nt + 31) // 32 * 32
            tf = torch.empty(N, ld, dtype=torch.float32, device=self.device)
            call('ofx_graph_type_frac', ptr(seg_ptr), ptr(col), ptr(self._ntype8[d]), N, nt, ptr(tf), ld,
                 stream())
            self._tf[key] = tf
        return self._tf[key]

    def type_frac_planes(self, d, nt, mode):
        """type_frac(d, nt) as operand planes of the LDS-DMA GraphConv (zero-padded to a whole chunk)."""
        key = ('tfp', d, nt, mode)
        if key not in self._tf:
            from . import ops
            self._tf[key] = ops.planes_split(self.type_frac(d, nt), mode)
        return self._tf[key]

    def get_input_feature(self, all_leaf_nodes=True):
        """dual_octree.py:343-360: the 'ND' feature of the finest octree layer (zero rows for its empty nodes),
        preceded by zero rows for the leaves of the shallower layers -> [N_depth, 4], the VAE encoder's input."""
        data = self.octree.get_input_feature('ND', nempty=False)
        if all_leaf_nodes:
            n_graph = self.csr(self.depth)[2]
            out = torch.zeros(n_graph, data.shape[1], dtype=torch.float32, device=data.device)
            out[n_graph - data.shape[0]:] = data
            data = out
        return data

    def pad_rows(self, d):
        """int32 [N_d]: position of graph row r inside the node_mask-long padded array
        (graph_vae.py:214-221: `pad[node_mask] = reg`)."""
        key = ('pad', d)
        if key not in self._maps:
            self._maps[key] = torch.nonzero(self.graph[d]['node_mask']).reshape(-1).to(torch.int32)
        return self._maps[key]

    def pool_maps(self, d):
        """Row maps for GraphDownsample d -> d-1 (modules.py:409-423).

        copy_src[r]: source row in x for output row r (-1: produced by the GEMM);
        gemm_rows[c]: output row of the c-th pooled (non-leaf) node of depth d-1.
        """
        key = ('pool', d)
        if key not in self._maps:
            numd, lnum1, nnum1 = int(self.nnum[d]), int(self.lnum[d - 1]), int(self.nnum[d - 1])
            Nd = self._csr[d][2]
            L0 = Nd - numd - lnum1
            leaf = self.node_child(d - 1) < 0
            s = int(self.ncum[d - 1])
            lrank = self._leafrank[s: s + nnum1].to(torch.int64)
            node_src = torch.where(leaf, L0 + lrank, torch.full_like(lrank, -1))
            copy_src = torch.cat([torch.arange(L0, device=self.device, dtype=torch.int64), node_src])
            gemm_rows = L0 + torch.nonzero(~leaf).reshape(-1)
            self._maps[key] = (copy_src.to(torch.int32), gemm_rows.to(torch.int32), L0 + nnum1)
        return self._maps[key]

    def unpool_maps(self, d):
        """Row maps for GraphUpsample d -> d+1 (modules.py:458-467).

        copy_src[r] for the first L0 + lnum[d] output rows; a_rows = rows of x that get
        unpooled (non-leaf nodes of depth d); the GEMM output starts at row n_copy.
        """
        key = ('unpool', d)
        if key not in self._maps:
            numd = int(self.nnum[d])
            Nd = self._csr[d][2]
            L0 = Nd - numd
            leaf = self.node_child(d) < 0
            leaf_idx = torch.nonzero(leaf).reshape(-1)
            copy_src = torch.cat([torch.arange(L0, device=self.device, dtype=torch.int64), L0 + leaf_idx])
            a_rows = L0 + torch.nonzero(~leaf).reshape(-1)
            self._maps[key] = (copy_src.to(torch.int32), a_rows.to(torch.int32), int(copy_src.numel()))
        return self._maps[key]
