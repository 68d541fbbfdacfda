"""nn.Module mirrors of the reference's dual-octree building blocks.

Constructor arguments, ``forward`` signatures, parameter names and shapes follow
reference models/networks/modules.py (cited per class) so checkpoints load with
``strict=True`` and callers (graph_unet_hr / graph_vae) need no change.  The
bodies do not: every forward is a short sequence of libofx launches
(octfusion_amd.ops) -- fused gather+contraction GraphConv, two-kernel GroupNorm
with fused SiLU, row-map pool / unpool -- with no boolean-mask indexing (hence no
host sync inside a denoising step) and no materialised ``col_data``.

Inference only (the sampling path runs under no_grad): these
modules do not build an autograd graph.
"""
import math

import torch
import torch.nn as nn

from . import ops


def _gn_groups(channels, group=32):
    # reference modules.py:271-280
    if channels <= 32:
        group = channels // 4
    elif channels % group != 0:
        group = 30
    assert channels % group == 0
    return group


class GraphConv(nn.Module):
    """reference modules.py:163-220."""

    def __init__(self, in_channels, out_channels, n_edge_type=7, avg_degree=7, n_node_type=0,
                 use_bias=False):
        super().__init__()
        assert n_edge_type == 7, 'the dual octree graph has 7 edge types'
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.use_bias = use_bias
        self.n_edge_type = n_edge_type
        self.avg_degree = avg_degree
        self.n_node_type = n_node_type
        node_channel = n_node_type if n_node_type > 1 else 0
        self.weights = nn.Parameter(torch.empty(n_edge_type * (in_channels + node_channel), out_channels))
        if use_bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()
        self._pw = ops.PackedWeight()
        self._pw2 = ops.PackedPlanes()
        self._pno = ops.PackedNarrowOut()
        self.emit_stats = True

    def planes_mode(self, doctree, d, cin=None):
        """Operand-plane format this layer wants its input in (0: plain fp32 rows): the LDS-DMA kernel takes the
        layers with enough 256 x 128 tiles and whole 32 (64)-channel chunks."""
        mode = ops.planes_mode()
        if not mode:
            return 0
        cin = self.in_channels if cin is None else (cin if ops.planes_pairs(mode) else 64)
        N = doctree.csr(d)[2]
        tiles = ((N + 255) // 256) * ((self.out_channels + 127) // 128)
        if (cin % (32 if ops.planes_pairs(mode) else 64) or self.out_channels % 4 or self.out_channels < 64 or
                tiles < ops.PLANES_MIN_TILES):
            return 0
        return mode

    def reset_parameters(self):
        fan_in = self.avg_degree * self.in_channels
        fan_out = self.avg_degree * self.out_channels
        a = math.sqrt(3.0) * math.sqrt(2.0 / float(fan_in + fan_out))
        nn.init.uniform_(self.weights, -a, a)
        if self.use_bias:
            nn.init.zeros_(self.bias)

    @torch.no_grad()
    def forward(self, x, doctree, d, emb=None, res=None, out=None, split_input=False):
        """``emb`` [B, Cout] (added per batch element) and ``res`` [N, Cout] are optional
        fused epilogue terms (the reference adds them with separate ops).  ``x`` may be operand planes
        (written by the preceding GroupNorm); ``split_input``: convert an fp32 ``x`` when this layer
        qualifies for the planes kernel."""
        nt = self.n_node_type if self.n_node_type > 1 else 0
        seg_ptr, col, N, E = doctree.csr(d)
        assert x.shape[0] == N, 'x has %d rows, graph depth %d has %d nodes' % (x.shape[0], d, N)
        # the output feeds a DualOctreeGroupNorm next: let the epilogue accumulate its statistics (from 16 k elements up:
        # below that a stand-alone statistics launch is as cheap; until round 4 the bar was 1 M elements, which left the
        # one-shape step -- the reference's own sampling batch -- with four 23 us statistics launches)
        stats = None
        if self.emit_stats and N * self.out_channels >= (1 << 14) and self.out_channels % 4 == 0:
            stats = ops.stats_zeros(doctree.batch_size * self.out_channels * 2, x.device)
        mode = ops.planes_of(x)
        cin_k = self.in_channels
        if not mode and emb is None and res is None:
            # the two gather-shaped layers of a U-Net (csrc/ofx_narrow.hip): its input and its output convolution
            if ops.narrow_in_ok(self.in_channels, self.out_channels, nt):
                y = ops.graphconv_narrow_in(x, seg_ptr, col, self.weights, self.in_channels, nt,
                                            doctree.node_type8(d) if nt else None, self.bias if self.use_bias else None,
                                            doctree.batch_id32(d) if stats is not None else None, out, stats,
                                            ext=doctree.ext(d))
                if stats is not None:
                    setattr(y, ops.STATS_ATTR, stats)
                return y
            if ops.NARROW_OUT and self.out_channels <= 8 and self.in_channels % 32 == 0 and N >= 4096:
                pno = self._pno.get(self.weights, self.in_channels, nt)
                bias = self.bias if self.use_bias else None
                # node-type term + bias of every row: constant per (doctree depth, weights), cached on the doctree
                tkey = ('narrow_tt', d, nt, self.weights.data_ptr(), self.weights._version,
                        None if bias is None else (bias.data_ptr(), bias._version))
                tt = doctree._tf.get(tkey)
                if tt is None and (nt or bias is not None):
                    tt = doctree._tf[tkey] = pno.type_term(doctree.type_frac(d, nt) if nt else None, nt, N, self.in_channels, bias)
                return ops.graphconv_narrow_out(x, seg_ptr, col, pno, self.in_channels, tt, out)
        if not mode and split_input:
            mode = self.planes_mode(doctree, d)
            if mode:
                x = ops.planes_split(x, mode)
        if not mode and self.in_channels < 32:
            # the network's input convolution (3 or 8 channels): zero-pad the channels to one 32 (64)-wide chunk and
            # run the planes kernel on it -- 9 k tiles instead of materialising the reference's col rows for a
            # dense GEMM (graph_unet_hr.py:116, modules.py:199-213)
            mode = self.planes_mode(doctree, d, cin=32)
            if mode:
                cin_k = 32 if ops.planes_pairs(mode) else 64
                x = ops.planes_split(x, mode, Cpad=cin_k)
        if mode:
            pw2 = self._pw2.get(self.weights, self.in_channels, nt, mode, cin_pad=cin_k)
            y = ops.graphconv_planes(x, mode, seg_ptr, col, doctree.ext(d), pw2, cin_k, nt,
                                     doctree.type_frac_planes(d, nt, mode) if nt else None,
                                     self.bias if self.use_bias else None, emb,
                                     doctree.batch_id32(d) if (emb is not None or stats is not None) else None,
                                     res, out, stats=stats)
            if stats is not None:
                setattr(y, ops.STATS_ATTR, stats)
            return y
        pw = self._pw.get(self.weights, 'graphconv', self.in_channels, nt)
        tf = doctree.type_frac(d, nt) if nt else None
        y = ops.graphconv(x, doctree.nbr(d), seg_ptr, col, pw, self.in_channels, tf,
                          self.bias if self.use_bias else None, emb,
                          doctree.batch_id32(d) if (emb is not None or stats is not None) else None, res, out,
                          ext=doctree.ext(d), stats=stats)
        if stats is not None:
            setattr(y, ops.STATS_ATTR, stats)
        return y

    def extra_repr(self):
        return 'channel_in={}, channel_out={}, n_edge_type={}, avg_degree={}, n_node_type={}'.format(
            self.in_channels, self.out_channels, self.n_edge_type, self.avg_degree, self.n_node_type)


class DualOctreeGroupNorm(nn.Module):
    """reference modules.py:262-330."""

    def __init__(self, in_channels, group=32, nempty=False):
        super().__init__()
        self.eps = 1e-5
        self.nempty = nempty
        self.in_channels = in_channels
        self.group = _gn_groups(in_channels, group)
        self.channels_per_group = in_channels // self.group
        self.weights = nn.Parameter(torch.ones(1, in_channels))
        self.bias = nn.Parameter(torch.zeros(1, in_channels))

    @torch.no_grad()
    def forward(self, data, doctree, depth, act=None, out=None, planes=0):
        """``planes``: write the result as operand planes of the LDS-DMA GraphConv (ops.planes_mode()) together with
        the aux rows (zero row + multi-neighbour means) that convolution gathers besides the tensor itself."""
        assert doctree.batch_id32(depth).shape[0] == data.shape[0]
        stats = ops.get_stats(data)
        if ops.planes_pairs(planes) and out is not None and not ops.planes_ok(out, planes):
            out = None
        aux_graph = None
        if planes:
            seg_ptr, col, _, _ = doctree.csr(depth)
            _, multi_seg, n_multi = doctree.ext(depth)
            plan = None
            if ops.AUX_PLAN == 'oct':
                plan = doctree.oct_plan(depth)
            elif ops.AUX_PLAN:
                plan = doctree.aux_plan(depth)
            aux_graph = (seg_ptr, col, multi_seg, n_multi, plan)
        y = ops.group_norm(data, doctree.batch_id32(depth), doctree.count(depth), doctree.batch_size,
                           self.weights, self.bias, self.group, self.eps, act, out, stats=stats, planes=planes,
                           aux_graph=aux_graph)
        if out is data and stats is not None:
            delattr(data, ops.STATS_ATTR)          # overwritten in place: the sums no longer describe it
        return y

    def extra_repr(self):
        return 'in_channels={}, group={}, nempty={}'.format(self.in_channels, self.group, self.nempty)


class _Linear(nn.Module):
    """nn.Linear-compatible parameters (``weight`` [out, in], ``bias``) on the MFMA GEMM."""

    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.in_features, self.out_features = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(cin)
            nn.init.uniform_(self.bias, -bound, bound)
        self._pw = ops.PackedWeight()

    @torch.no_grad()
    def forward(self, x, res=None, out=None, act_in=None, act_out=None):
        """act_in / act_out: activation of the input / the result fused into the launch (few-row path) or run as
        separate launches around the GEMM."""
        if ops.LINEAR_SMALL and x.dim() == 2 and x.shape[0] <= 16:
            return ops.linear_small(x, self.weight, self.bias, res, act_in, act_out, out)
        if act_in is not None:
            x = ops.act(x, act_in)
        y = ops.gemm(x, self._pw.get(self.weight, 'nk'), self.bias, res, out)
        return ops.act(y, act_out, out=y) if act_out is not None else y


class Conv1x1(nn.Module):
    """reference modules.py:332-339 (``linear.weight`` / ``linear.bias``)."""

    def __init__(self, channel_in, channel_out, use_bias=False):
        super().__init__()
        self.linear = _Linear(channel_in, channel_out, use_bias)

    def forward(self, x, res=None):
        return self.linear(x, res)


class Conv1x1Gn(nn.Module):
    """reference modules.py:341-351."""

    def __init__(self, channel_in, channel_out):
        super().__init__()
        self.conv = Conv1x1(channel_in, channel_out, use_bias=False)
        self.gn = DualOctreeGroupNorm(channel_out)

    def forward(self, x, doctree, depth):
        return self.gn(self.conv(x), doctree, depth)


class Conv1x1GnGelu(nn.Module):
    """reference modules.py:353-365."""

    def __init__(self, channel_in, channel_out):
        super().__init__()
        self.conv = Conv1x1(channel_in, channel_out, use_bias=False)
        self.gn = DualOctreeGroupNorm(channel_out)
        self.gelu = nn.GELU()

    def forward(self, x, doctree, depth):
        return self.gn(self.conv(x), doctree, depth, act='gelu')


class Conv1x1GnGeluSequential(Conv1x1GnGelu):
    """reference modules.py:367-380 (takes a (x, doctree, depth) tuple)."""

    def forward(self, data):
        x, doctree, depth = data
        return super().forward(x, doctree, depth)


class Downsample(nn.Module):
    """reference modules.py:382-398: [8n, C] -> [n, C] with ``weights`` [C, C, 8]."""

    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.weights = nn.Parameter(torch.empty(channels, channels, 8))
        nn.init.xavier_uniform_(self.weights)
        self._pw = ops.PackedWeight()

    def packed(self):
        # x.view(-1, 8C) @ W.flatten(1).t()  ==  nn.Linear layout [N=C, K=8C]
        return self._pw.get(self.weights.view(self.channels, self.channels * 8), 'nk')

    @torch.no_grad()
    def forward(self, x, out=None, out_rows=None, out_planes=0):
        C = self.channels
        # (ofx_gather_gemm_f32 wants a 16-B aligned base and a row pitch of whole float4s; any other slice takes the
        # reshape below, which copies)
        if (x.stride(0) != C and x.stride(1) == 1 and C % 32 == 0 and x.shape[0] % 8 == 0 and x.shape[0]
                and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and ops.zero_row(x.device).numel() >= C):
            # x is a column slice of a wider buffer (the zero-copy skip concatenation): x.view(-1, 8 C) would copy it
            # (two strided ATen copies per hr step, 47 us, until round 5) -- gather the eight children instead
            n = x.shape[0] // 8
            tab = _child_table(n, x.device)
            return ops.gather_gemm(x, tab, 8, self.packed(), n, out=out, out_rows=out_rows, out_planes=out_planes)
        return ops.gemm(x.reshape(-1, C * 8), self.packed(), out=out, out_rows=out_rows, out_planes=out_planes)

    def extra_repr(self):
        return 'channels={}'.format(self.channels)


_CHILD_TABLES = {}


def _child_table(n, device):
    """int32 [n, 8]: row 8 r + j (the j-th child of node r) -- the gather table of Downsample; cached per size."""
    key = (n, device.type, device.index)
    t = _CHILD_TABLES.get(key)
    if t is None:
        if len(_CHILD_TABLES) > 64:
            _CHILD_TABLES.clear()
        t = _CHILD_TABLES[key] = torch.arange(8 * n + 8, dtype=torch.int32, device=device)     # (+ 32 B of slack behind the last entry)
    return t


class Upsample(nn.Module):
    """reference modules.py:430-446: [n, C] -> [8n, C]."""

    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.weights = nn.Parameter(torch.empty(channels, channels, 8))
        nn.init.xavier_uniform_(self.weights)
        self._pw = ops.PackedWeight()
        self._pgp = ops.PackedGemmPlanes()

    def packed(self):
        # x @ W.flatten(1): plain [K=C, N=8C]
        return self._pw.get(self.weights.view(self.channels, self.channels * 8), 'kn')

    @torch.no_grad()
    def forward(self, x, a_rows=None, out=None, out_planes=0):
        C = self.channels
        mode = ops.planes_mode()
        if (ops.GEMM_PLANES and out is not None and a_rows is not None and ops.planes_pairs(mode) and out_planes == mode
                and C % 32 == 0 and C >= 256 and a_rows.numel() >= 256):
            # the unpool GEMM on the data path of the planes GraphConv (ofx_gemm_planes): the rows to unpool are gathered and
            # split into operand planes (M x C: a few MB), the [M, C] x [C, 8 C] product runs on the persistent LDS-DMA
            # kernel and its epilogue writes the children's rows as the planes the following GraphConv gathers
            M = a_rows.numel()
            xa = torch.empty(M, C, dtype=torch.float32, device=x.device)
            ops.rows_copy(x, xa, M, smap=a_rows, planes=mode)
            setattr(xa, ops.PLANES_ATTR, mode)
            if ops.gemm_planes(xa, self._pgp.get(self.weights.view(C, C * 8), mode), out, out_planes):
                return out
        y = ops.gemm(x, self.packed(), out=out, a_rows=a_rows, out_planes=out_planes)
        return y.view(-1, self.channels) if out is None else out

    def extra_repr(self):
        return 'channels={}'.format(self.channels)


def _planes_dst(n, C, planes, device):
    """Destination of a pool / unpool when the following GraphConv wants operand planes: the copy and the GEMM epilogue
    write hi / lo pairs straight into it (no fp32 tensor, no ofx_planes_split pass).  Returns (tensor, mode used)."""
    out = torch.empty(n, C, dtype=torch.float32, device=device)
    ok = ops.planes_pairs(planes) and C % 32 == 0 and out.data_ptr() % 128 == 0
    return out, (planes if ok else 0)


def pool_nodes(x, doctree, d, downsample, planes=0):
    """Rows of graph depth d -> rows of depth d-1 (reference modules.py:409-423, without masks).  planes: pair-plane
    mode the consumer takes (0: fp32 rows)."""
    copy_src, gemm_rows, n_out = doctree.pool_maps(d)
    C = x.shape[1]
    numd = int(doctree.nnum[d])
    out, planes = _planes_dst(n_out, C, planes, x.device)
    ops.rows_copy(x, out, n_out, smap=copy_src, planes=planes)
    downsample(x[x.shape[0] - numd:], out=out, out_rows=gemm_rows, out_planes=planes)
    if planes:
        setattr(out, ops.PLANES_ATTR, planes)
    return out


def unpool_nodes(x, doctree, d, upsample, planes=0):
    """Rows of graph depth d -> rows of depth d+1 (reference modules.py:458-467)."""
    copy_src, a_rows, n_copy = doctree.unpool_maps(d)
    C = x.shape[1]
    n_ne = a_rows.numel()
    out, planes = _planes_dst(n_copy + 8 * n_ne, C, planes, x.device)
    if planes and n_ne and out[n_copy:].data_ptr() % 128:
        planes = 0                                  # (C % 32 == 0 makes every row start a 128-B line: cannot happen)
    ops.rows_copy(x, out, n_copy, smap=copy_src, planes=planes)
    if n_ne:
        upsample(x, a_rows=a_rows, out=out[n_copy:].view(n_ne, 8 * C), out_planes=planes)
    if planes:
        setattr(out, ops.PLANES_ATTR, planes)
    return out


class GraphDownsample(nn.Module):
    """reference modules.py:400-428 (U-Net flavour; ``d`` = depth of the INPUT)."""

    def __init__(self, channels_in, channels_out, n_edge_type, avg_degree, n_node_type):
        super().__init__()
        self.channels_in = channels_in
        self.channels_out = channels_out
        self.downsample = Downsample(channels_in)
        self.conv = GraphConv(channels_in, channels_out, n_edge_type, avg_degree, n_node_type)

    def forward(self, x, doctree, d, out=None):
        return self.conv(pool_nodes(x, doctree, d, self.downsample, planes=self.conv.planes_mode(doctree, d - 1)),
                         doctree, d - 1, out=out, split_input=True)


class GraphUpsample(nn.Module):
    """reference modules.py:449-472 (U-Net flavour; ``d`` = depth of the INPUT)."""

    def __init__(self, channels_in, channels_out, n_edge_type, avg_degree, n_node_type):
        super().__init__()
        self.channels_in = channels_in
        self.channels_out = channels_out
        self.upsample = Upsample(channels_in)
        self.conv = GraphConv(channels_in, channels_out, n_edge_type, avg_degree, n_node_type)

    def forward(self, x, doctree, d, out=None):
        return self.conv(unpool_nodes(x, doctree, d, self.upsample, planes=self.conv.planes_mode(doctree, d + 1)),
                         doctree, d + 1, out=out, split_input=True)


def graphnormalization(channels):
    return DualOctreeGroupNorm(channels, min(32, channels))


class TimestepBlock(nn.Module):
    pass


class GraphResBlockEmbed(TimestepBlock):
    """reference modules.py:661-763.  use_scale_shift_norm must stay False (the reference
    branch reads a non-existent attribute, :747-751)."""

    def __init__(self, channels, emb_channels, dropout, out_channels, n_edge_type, avg_degree,
                 n_node_type, use_conv=False, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, up=False, down=False):
        super().__init__()
        if use_scale_shift_norm or use_conv:
            raise NotImplementedError('use_scale_shift_norm / use_conv are dead in the reference')
        self.channels = channels
        self.emb_channels = emb_channels
        self.out_channels = channels if out_channels is None else out_channels
        self.use_checkpoint = use_checkpoint
        self.block1_norm = graphnormalization(self.channels)
        self.silu = nn.SiLU()
        self.conv1 = GraphConv(self.channels, self.out_channels, n_edge_type, avg_degree, n_node_type)
        self.emb_layers = nn.Sequential(nn.SiLU(), _Linear(emb_channels, self.out_channels))
        self.block2_norm = graphnormalization(self.out_channels)
        self.dropout = nn.Dropout(p=dropout)
        self.conv2 = GraphConv(self.out_channels, self.out_channels, n_edge_type, avg_degree, n_node_type)
        for p in self.conv2.parameters():        # zero_module (:719)
            p.detach().zero_()
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = Conv1x1(self.channels, self.out_channels)

    @torch.no_grad()
    def forward(self, x, emb, doctree, depth, emb_act=None, out=None, emb_out=None):
        """``emb_act``: optional precomputed SiLU(emb) shared by all blocks of a step; ``emb_out``: optional
        precomputed emb_layers(emb) [B, Cout] (the U-Net evaluates the emb_layers of ALL its blocks in one GEMM);
        ``out``: optional destination (may be a column slice of a wider buffer: zero-copy skip concatenation)."""
        skip, side = x, None
        if not isinstance(self.skip_connection, nn.Identity):
            if ops.SIDE_STREAM:
                # the 1x1 skip convolution only depends on x: run it on the side stream, where it fills the CUs the
                # conv1 chain leaves idle (tile-count tails, prologues); joined right before conv2 adds it
                main, side = torch.cuda.current_stream(x.device), ops.side_stream(x.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    skip = self.skip_connection(x)
            else:
                skip = self.skip_connection(x)
        h = self.block1_norm(x, doctree, depth, act='silu', planes=self.conv1.planes_mode(doctree, depth))
        if emb_out is None:
            if emb_act is None:
                emb_act = ops.act(emb, 'silu')
            emb_out = self.emb_layers[1](emb_act)                   # [B, Cout]
        assert doctree.batch_size == emb_out.shape[0]
        h = self.conv1(h, doctree, depth, emb=emb_out)              # + emb_out[batch_id] fused
        h = self.block2_norm(h, doctree, depth, act='silu', out=h, planes=self.conv2.planes_mode(doctree, depth))
        if side is not None:
            main.wait_stream(side)
        return self.conv2(h, doctree, depth, res=skip, out=out)     # skip + h fused


class GraphResBlock(nn.Module):
    """reference modules.py:593-641 (VAE flavour)."""

    def __init__(self, channel_in, channel_out, dropout, n_edge_type=7, avg_degree=7, n_node_type=0,
                 use_checkpoint=False):
        super().__init__()
        self.channel_in = channel_in
        self.channel_out = channel_out
        self.use_checkpoint = use_checkpoint
        self.norm1 = DualOctreeGroupNorm(channel_in)
        self.conv1 = GraphConv(channel_in, channel_out, n_edge_type, avg_degree, n_node_type)
        self.norm2 = DualOctreeGroupNorm(channel_out)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = GraphConv(channel_out, channel_out, n_edge_type, avg_degree, n_node_type)
        if channel_in != channel_out:
            self.conv1x1c = Conv1x1Gn(channel_in, channel_out)

    @torch.no_grad()
    def forward(self, x, doctree, depth):
        h = self.norm1(x, doctree, depth, act='silu', planes=self.conv1.planes_mode(doctree, depth))
        h = self.conv1(h, doctree, depth)
        h = self.norm2(h, doctree, depth, act='silu', out=h, planes=self.conv2.planes_mode(doctree, depth))
        if self.channel_in != self.channel_out:
            x = self.conv1x1c(x, doctree, depth)
        return self.conv2(h, doctree, depth, res=x)


class GraphResBlocks(nn.Module):
    """reference modules.py:643-659."""

    def __init__(self, channel_in, channel_out, dropout, resblk_num, n_edge_type=7, avg_degree=7,
                 n_node_type=0, use_checkpoint=False):
        super().__init__()
        self.resblk_num = resblk_num
        channels = [channel_in] + [channel_out] * resblk_num
        self.resblks = nn.ModuleList([
            GraphResBlock(channels[i], channels[i + 1], dropout, n_edge_type, avg_degree, n_node_type,
                          use_checkpoint) for i in range(resblk_num)])

    def forward(self, data, doctree, depth):
        for blk in self.resblks:
            data = blk(data, doctree, depth)
        return data
