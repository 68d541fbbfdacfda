"""Oracle: dual-octree neighbour graph (torch CPU int64).  TEST INFRASTRUCTURE.

Restates reference models/networks/dualoctree_networks/dual_octree.py level by
level (the product builds the same graph with a different, per-(node,face)
algorithm in HIP; both are checked against golden vectors captured from the
reference file itself).

Edge direction codes (dual_octree.py:85-97,247): 0:+z 1:-z 2:+y 3:-y 4:+x 5:-x
6:self.
"""
import torch

from .octree import cumsum, key2xyz, xyz2key

# dual_octree.py:85-112 (lookup tables)
_NGH = torch.tensor([[0, 0, 1], [0, 0, -1], [0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0]],
                    dtype=torch.int64)
_DIR_TABLE = torch.tensor([[1, 3, 5, 7], [0, 2, 4, 6], [2, 3, 6, 7], [0, 1, 4, 5],
                           [4, 5, 6, 7], [0, 1, 2, 3]], dtype=torch.int64)
_REMAP = torch.tensor([1, 0, 3, 2, 5, 4], dtype=torch.int64)
_INTER_ROW = torch.tensor([0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3,
                           4, 4, 4, 5, 5, 5, 6, 6, 6, 7, 7, 7], dtype=torch.int64)
_INTER_COL = torch.tensor([1, 2, 4, 0, 3, 5, 0, 3, 6, 1, 2, 7,
                           0, 5, 6, 1, 4, 7, 2, 4, 7, 3, 5, 6], dtype=torch.int64)
_INTER_DIR = torch.tensor([0, 2, 4, 1, 2, 4, 3, 0, 4, 3, 1, 4,
                           5, 0, 2, 5, 1, 2, 5, 3, 0, 5, 3, 1], dtype=torch.int64)


class OracleDualOctree:
    """Same attribute surface the reference modules read."""

    def __init__(self, octree):
        # dual_octree.py:19-63
        self.octree = octree
        self.device = 'cpu'
        self.depth = octree.depth
        self.full_depth = octree.full_depth
        self.batch_size = octree.batch_size
        self.nnum = octree.nnum.clone()
        self.nenum = octree.nnum_nempty.clone()
        self.ncum = cumsum(self.nnum, dim=0, exclusive=True)
        self.lnum = self.nnum - self.nenum
        self.node_depth = torch.cat([
            torch.full((int(self.nnum[d]),), d, dtype=torch.int64) for d in range(self.depth + 1)])
        self.child = torch.cat([c for c in octree.children if c is not None]).to(torch.int64)
        self.key = torch.cat([k for k in octree.keys if k is not None])
        self.keyd = self.key | (self.node_depth << 58)
        x, y, z, b = key2xyz(self.key)
        self.xyz = torch.stack([x, y, z], dim=1)
        self.batch = b

        self._graph = [None] * (self.depth + 1)
        self.graph = [None] * (self.depth + 1)
        self._graph[self.full_depth] = self._dense_graph(self.full_depth)
        for d in range(self.full_depth + 1, self.depth + 1):
            self._graph[d] = self._sparse_graph(d, self._graph[d - 1])

        self.batch_id_dict = {}
        self._calc_batch_id()
        self.total_num = len(self.batch_id_dict[self.depth])

    # dual_octree.py:65-82
    def _calc_batch_id(self):
        leaf = torch.zeros(0, dtype=torch.int64)
        for d in range(self.full_depth, self.depth + 1):
            cur = self.octree.batch_id(d, nempty=False)
            if d > self.full_depth:
                empty = self.octree.children[d - 1] < 0
                leaf = torch.cat([leaf, self.octree.keys[d - 1][empty] >> 48])
            self.batch_id_dict[d] = torch.cat([leaf, cur])

    def batch_id(self, depth, nempty=False):
        return self.batch_id_dict[depth]

    def node_child(self, depth):
        # dual_octree.py:189-193
        s = int(self.ncum[depth])
        return self.child[s: s + int(self.nnum[depth])]

    # dual_octree.py:124-155
    def _dense_graph(self, depth):
        bnd = 1 << depth
        num = bnd ** 3
        ki = torch.arange(num, dtype=torch.int64)
        x, y, z, _ = key2xyz(ki, depth)
        xi = torch.stack([x, y, z], dim=1)
        xj = xi.unsqueeze(1) + _NGH                      # [num, 6, 3]
        valid = ((xj > -1) & (xj < bnd)).all(dim=-1).reshape(-1)
        flat = xj.reshape(-1, 3)[valid]
        col = xyz2key(flat[:, 0], flat[:, 1], flat[:, 2], None, depth)
        row = ki.unsqueeze(1).repeat(1, 6).reshape(-1)[valid]
        edir = torch.arange(6, dtype=torch.int64).repeat(num)[valid]
        dis = torch.arange(self.batch_size, dtype=torch.int64).unsqueeze(1) * num + self.ncum[depth]
        row = (row.unsqueeze(0) + dis).reshape(-1)
        col = (col.unsqueeze(0) + dis).reshape(-1)
        edir = edir.unsqueeze(0).repeat(self.batch_size, 1).reshape(-1)
        return {'edge_idx': torch.stack([row, col]), 'edge_dir': edir}

    # dual_octree.py:166-187
    def _relative_dir(self, vi, vj, depth, rescale=True):
        xi = self.xyz[vi]
        xj = self.xyz[vj]
        xn = xi.unsqueeze(1) + _NGH
        scale = torch.ones_like(vj)
        if rescale:
            dj = self.node_depth[vj]
            scale = torch.pow(2.0, depth - dj)           # one fp32 pow, as in the reference
            xj = xj * scale.unsqueeze(-1)
        xj = xj.unsqueeze(1)
        s = scale.view(-1, 1, 1)
        inbox = ((xn >= xj) & (xn < xj + s)).all(dim=-1)
        return torch.argmax(inbox.byte(), dim=-1)

    # dual_octree.py:195-239
    def _sparse_graph(self, depth, graph):
        ncum_d = int(self.ncum[depth])
        nn = int(self.nnum[depth])
        base = (torch.arange(nn // 8, dtype=torch.int64) * 8 + ncum_d).unsqueeze(1)
        row_i = (_INTER_ROW.unsqueeze(0) + base).reshape(-1)
        col_i = (_INTER_COL.unsqueeze(0) + base).reshape(-1)
        dir_i = _INTER_DIR.unsqueeze(0).repeat(nn // 8, 1).reshape(-1)

        row, col = graph['edge_idx'][0], graph['edge_idx'][1]
        edir = graph['edge_dir']
        leaf_r = self.child[row] < 0
        leaf_c = self.child[col] < 0
        keep = leaf_r & leaf_c
        only_row = (~leaf_r) & leaf_c
        both = (~leaf_r) & (~leaf_c)

        vi, vj = row[only_row], col[only_row]
        rd = self._relative_dir(vi, vj, depth - 1)
        row_o1 = (self.child[vi].unsqueeze(1) * 8 + _DIR_TABLE[rd]).reshape(-1) + ncum_d
        col_o1 = vj.unsqueeze(1).repeat(1, 4).reshape(-1)
        dir_o1 = rd.unsqueeze(1).repeat(1, 4).reshape(-1)

        row_o2 = col_o2 = dir_o2 = torch.zeros(0, dtype=torch.int64)
        if both.any():
            vi, vj = row[both], col[both]
            rd = self._relative_dir(vi, vj, depth - 1, rescale=False)
            row_o2 = (self.child[vi].unsqueeze(1) * 8 + _DIR_TABLE[rd]).reshape(-1) + ncum_d
            dir_o2 = rd.unsqueeze(1).repeat(1, 4).reshape(-1)
            col_o2 = (self.child[vj].unsqueeze(1) * 8 + _DIR_TABLE[_REMAP[rd]]).reshape(-1) + ncum_d

        return {
            'edge_idx': torch.stack([torch.cat([row[keep], row_i, row_o1, col_o1, row_o2]),
                                     torch.cat([col[keep], col_i, col_o1, row_o1, col_o2])]),
            'edge_dir': torch.cat([edir[keep], dir_i, dir_o1, _REMAP[dir_o1], dir_o2])}

    # dual_octree.py:400-409
    def post_processing_for_docnn(self):
        fd, depth = self.full_depth, self.depth
        leaf_nodes = self.child < 0
        leaf_masks, lnts, keyd_leaf = [], [], []
        for i, d in enumerate(range(fd, depth + 1)):
            # add_self_loops :241-249
            row, col = self._graph[d]['edge_idx']
            edir = self._graph[d]['edge_dir']
            uniq = torch.unique(row, sorted=True)
            row = torch.cat([row, uniq])
            col = torch.cat([col, uniq])
            edir = torch.cat([edir, torch.full_like(uniq, 6)])
            # remap_node_idx :265-271
            nd = int(self.nnum[d])
            mask = torch.cat([leaf_nodes[:int(self.ncum[d])], torch.ones(nd, dtype=torch.bool)])
            remap = torch.cumsum(mask.long(), 0) - 1
            row, col = remap[row], remap[col]
            # add_node_type :381-389
            ntype = d - fd
            lnts_cat = lnts[:i]
            node_type = torch.cat(lnts_cat + [torch.full((nd,), ntype, dtype=torch.int64)])
            lnts.append(torch.full((int(self.lnum[d]),), ntype, dtype=torch.int64))
            # add_node_keyd :362-369 / add_node_mask :391-398
            s = int(self.ncum[d])
            keyd_d = self.keyd[s: s + nd]
            leaf_d = self.child[s: s + nd] < 0
            keyd = torch.cat(keyd_leaf[:i] + [keyd_d])
            node_mask = torch.cat(leaf_masks[:i] + [torch.ones(nd, dtype=torch.bool)])
            keyd_leaf.append(keyd_d[leaf_d])
            leaf_masks.append(leaf_d)
            # sort_edges :332-341 (stable here; the reference's argsort is unstable, so
            # in-segment order is unspecified there -- compare canonicalised)
            order = torch.argsort(row * 7 + edir, stable=True)
            self.graph[d] = {'edge_idx': torch.stack([row[order], col[order]]),
                             'edge_dir': edir[order], 'node_type': node_type,
                             'keyd': keyd, 'node_mask': node_mask}


def canonical_edges(edge_idx, edge_dir):
    """Sort edges by (row, dir, col): the order-free form used for bit-exact compares."""
    row, col = edge_idx[0].to(torch.int64), edge_idx[1].to(torch.int64)
    n = int(max(int(row.max()), int(col.max())) + 1) if row.numel() else 1
    key = (row * 7 + edge_dir.to(torch.int64)) * n + col
    order = torch.argsort(key, stable=True)
    return torch.stack([row[order], col[order]]), edge_dir.to(torch.int64)[order]
