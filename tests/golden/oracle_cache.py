"""CPU-oracle outputs of the full-width GPU parity tests as committed fixtures.

The GPU box has a 16-CPU quota and the full-width oracle runs (hr B = 8, the shell-8 feature step, the fp64 per-layer
sweep, 50 DDIM steps) were ~60 % of the GPU suite's wall time.  They depend on nothing the GPU produces, so they are
computed ONCE, here, by `python tests/golden/make_oracle_cache.py` (which imports the test modules on a CPU-only
machine and runs every registered case), and the `-m gpu` tests only run the HIP side against the stored result.

  * a *case* = a named CPU-only function returning {key: tensor}; registered by the test module that uses it
    (`@case('name')` / `register(name, fn)`), looked up with `get(name)`;
  * a fixture is keyed by `case_digest(name)`: the sha256 of oracle/*.py, tests/golden/common.py,
    tests/golden/oracle_cache.py, octfusion_amd/synthetic.py, octfusion_amd/configs.py AND the test module that
    registered the case (its oracle invocation: arguments, steps, timesteps -- ADVICE r05) -- everything a case's inputs
    and outputs are derived from.  A fixture whose digest differs from the tree FAILS LOUDLY (stale oracle output must
    never pass a test; tests/test_oracle_golden.py::test_oracle_cache_fixtures_are_current checks every committed file
    on the CPU); a missing fixture falls back to computing the case on the spot (the slow path of round 4);
  * tensors of more than FULL_MAX elements are stored as a `Sketch` (unless the case wraps them in `Full`):
      - a seeded sample of whole rows (first / last rows always included; a case may name more rows, e.g. the rows
        next to the tile / share / XCD-group boundaries of a launch plan) for the element-wise figures;
      - BLOCK PROJECTIONS over ALL rows (round 6): for every block of PBLOCK = 64 consecutive rows
        NPROJ = 8 sums  sum_i r_p(i) * (t[i, :] . q_p)  with seeded signs r_p, q_p in {-1, +1}, float64.  A difference
        of `e` (relative to the tensor's max) in EVERY element of ONE row moves its block's projections by
        ~ e * sqrt(C) against a tolerance of tol * sqrt(64 * C), i.e. a single wrong row is seen from e >= 8 * tol and
        a wrong 64-row tile from e >= tol -- in ANY of the rows, where the row sample of round 5 saw 1-2.5 % of them
        (tests/test_oracle_golden.py::test_sketch_detects_a_localised_error plants both);
      - NBUCKET = 256 signed bucket sums (rows i = b mod 256), kept as a second, differently-shaped view.
    `errors(y, ref)` takes a tensor, a `Sketch` or a `Full` reference and folds all views into `rel_to_max`.

Test infrastructure only: nothing under octfusion_amd/ imports this file (tests/test_abi.py checks).
"""
import glob
import hashlib
import os
import zlib

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DIR = os.path.join(ROOT, 'tests', 'golden', 'oracle_cache')
FULL_MAX = 1 << 16          # elements stored verbatim (256 KB of fp32)
SAMPLE_ELEMS = 1 << 15      # elements of the row sample of a sketch
NBUCKET = 256
NPROJ = 8                   # block projections per block of PBLOCK rows
PBLOCK = 64

REGISTRY = {}
_DIGEST = None
_LOADED = {}


def digest():
    global _DIGEST
    if _DIGEST is None:
        h = hashlib.sha256()
        files = sorted(glob.glob(os.path.join(ROOT, 'oracle', '*.py')))
        files += [os.path.join(ROOT, 'tests', 'golden', 'common.py'), os.path.abspath(__file__),
                  os.path.join(ROOT, 'octfusion_amd', 'synthetic.py'), os.path.join(ROOT, 'octfusion_amd', 'configs.py')]
        for f in files:
            h.update(os.path.relpath(f, ROOT).encode())
            h.update(open(f, 'rb').read())
        _DIGEST = h.hexdigest()[:16]
    return _DIGEST


def register(name, fn):
    REGISTRY[name] = fn
    return fn


_FILE_SHA = {}


def case_digest(name):
    """digest() + the bytes of the test module that registered case `name` (falls back to digest() for a case nobody
    registered in this process: `get` then cannot recompute it either)."""
    fn = REGISTRY.get(name)
    if fn is None:
        return digest()
    f = fn
    while hasattr(f, 'func'):                     # functools.partial
        f = f.func
    path_ = getattr(getattr(f, '__code__', None), 'co_filename', None)
    if not path_ or not os.path.exists(path_):
        return digest()
    if path_ not in _FILE_SHA:
        _FILE_SHA[path_] = hashlib.sha256(open(path_, 'rb').read()).hexdigest()
    return hashlib.sha256((digest() + _FILE_SHA[path_]).encode()).hexdigest()[:16]


class Full:
    """Marks a tensor a case wants stored VERBATIM whatever its size (one full-tensor comparison at the bench's size)."""

    def __init__(self, t):
        self.t = t


class Rows:
    """A tensor to be sketched with extra sampled rows `rows` (int64 indices) besides the seeded ones."""

    def __init__(self, t, rows):
        self.t, self.rows = t, rows


def case(name):
    return lambda fn: register(name, fn)


class Sketch:
    """Row sample + signed bucket sums of a [N, C] (or any-shape, flattened to rows of the last dim) tensor."""

    def __init__(self, t, name, extra_rows=None):
        shape = tuple(t.shape)
        t2 = t.detach().reshape(-1, shape[-1]) if t.dim() > 1 else t.detach().reshape(-1, 1)
        N, C = t2.shape
        self.shape, self.N, self.C = shape, N, C
        n_s = min(N, max(128, SAMPLE_ELEMS // C))
        g = torch.Generator().manual_seed(zlib.crc32(('sketch:' + name).encode()) & 0x7FFFFFFF)
        idx = torch.randperm(N, generator=g)[:n_s]
        edge = torch.cat([torch.arange(min(32, N)), torch.arange(max(0, N - 32), N)])
        parts = [idx, edge]
        if extra_rows is not None:
            parts.append(extra_rows.reshape(-1).long().clamp(0, N - 1))
        self.idx = torch.unique(torch.cat(parts)).to(torch.int32)
        self.rows = t2[self.idx.long()].float().clone()
        self.absmax = float(t2.abs().max())
        self.bsum = bucket_sums(t2.double()).float().cpu()      # (fp32 storage of the fp64 sums: 6e-8 relative)
        self.proj = block_projections(t2).float().cpu()

    def state(self):
        return dict(sketch=2, shape=self.shape, N=self.N, C=self.C, idx=self.idx, rows=self.rows, absmax=self.absmax,
                    bsum=self.bsum, proj=self.proj)

    @classmethod
    def from_state(cls, s):
        o = cls.__new__(cls)
        o.shape, o.N, o.C, o.idx, o.rows, o.absmax, o.bsum, o.proj = (s['shape'], s['N'], s['C'], s['idx'], s['rows'],
                                                                      s['absmax'], s['bsum'], s['proj'])
        return o


def _signs(i, salt):
    """[len(i), NPROJ] of +-1 (float64) from a multiplicative hash of (index i, projection, salt): device-independent."""
    p = torch.arange(NPROJ, device=i.device, dtype=torch.int64)[None, :]
    h = (i[:, None] * 2654435761 + p * 40503 + salt) & 0xFFFFFFFF
    h = ((h ^ (h >> 15)) * 2246822519) & 0xFFFFFFFF
    h = ((h ^ (h >> 13)) * 3266489917) & 0xFFFFFFFF
    return (1 - 2 * ((h >> 16) & 1)).to(torch.float64)


def block_projections(t2):
    """fp64 [ceil(N / PBLOCK), NPROJ]: per block of PBLOCK consecutive rows, sum_i r_p(i) * (t2[i, :] . q_p)."""
    N, C = t2.shape
    dev = t2.device
    q = _signs(torch.arange(C, device=dev, dtype=torch.int64), 17)          # [C, NPROJ]
    out = torch.zeros((N + PBLOCK - 1) // PBLOCK, NPROJ, dtype=torch.float64, device=dev)
    step = 1 << 18                                           # rows per chunk (bounds the fp64 temporaries)
    for a in range(0, N, step):
        i = torch.arange(a, min(N, a + step), device=dev, dtype=torch.int64)
        out.index_add_(0, i // PBLOCK, (t2[a:a + step].double() @ q) * _signs(i, 91))
    return out


def bucket_sums(t2):
    """fp64 [NBUCKET, C]: sum over rows i of sign(i) * t2[i] into bucket i % NBUCKET; sign from a multiplicative hash."""
    N = t2.shape[0]
    i = torch.arange(N, device=t2.device, dtype=torch.int64)
    sign = (1 - 2 * (((i * 2654435761) >> 13) & 1)).to(torch.float64)
    out = torch.zeros(NBUCKET, t2.shape[1], dtype=torch.float64, device=t2.device)
    out.index_add_(0, i % NBUCKET, t2.double() * sign[:, None])
    return out


def pack(name, value):
    """tensor -> fp32 tensor or Sketch state (by size); dicts recursively; scalars as they are."""
    if isinstance(value, dict):
        return {k: pack(name + '/' + str(k), v) for k, v in value.items()}
    if isinstance(value, Full):
        return dict(full=1, t=value.t.detach().float().clone())
    if isinstance(value, Rows):
        if value.t.numel() > FULL_MAX:
            return Sketch(value.t, name, value.rows).state()
        value = value.t
    if torch.is_tensor(value):
        if value.is_floating_point() and value.numel() > FULL_MAX:
            return Sketch(value, name).state()
        return value.detach().float().clone() if value.is_floating_point() else value.detach().clone()
    return value


def unpack(value):
    if isinstance(value, dict):
        if value.get('sketch') == 2:
            return Sketch.from_state(value)
        if value.get('full') == 1:
            return value['t']
        return {k: unpack(v) for k, v in value.items()}
    return value


def path(name):
    return os.path.join(DIR, name + '.pt')


def write(name, value):
    os.makedirs(DIR, exist_ok=True)
    torch.save({'digest': case_digest(name), 'name': name, 'value': pack(name, value)}, path(name))


def get(name):
    """The stored result of case `name` (stale digest -> AssertionError), else the case computed now (slow path)."""
    if name in _LOADED:
        return _LOADED[name]
    p = path(name)
    if os.path.exists(p) and os.environ.get('OFX_ORACLE_CACHE', '1') != '0':
        rec = torch.load(p, weights_only=False)
        assert rec['digest'] == case_digest(name), (
            'tests/golden/oracle_cache/%s.pt was computed from other oracle / input / case sources (%s, tree is %s): run '
            '`python tests/golden/make_oracle_cache.py`' % (name, rec['digest'], case_digest(name)))
        out = unpack(rec['value'])
    else:
        assert name in REGISTRY, 'no oracle case %r registered' % name
        out = unpack(pack(name, REGISTRY[name]()))
    _LOADED[name] = out
    return out


def errors(a, b):
    """rel-to-max and element-wise figures of `a` (HIP result, any device) against `b` (tensor or Sketch)."""
    if isinstance(b, Sketch):
        a2 = a.detach().reshape(-1, b.C)
        assert tuple(a.shape) == tuple(b.shape), (tuple(a.shape), b.shape)
        ys = a2[b.idx.long().to(a2.device)].double().cpu()
        ref = b.rows.double()
        scale = b.absmax
        d = (ys - ref).abs()
        ew = d / ref.abs().clamp(min=1e-2 * scale)
        bs = bucket_sums(a2).cpu()
        bucket_rel = float((bs - b.bsum.double()).abs().max()) / (scale * max(1.0, (b.N / NBUCKET)) ** 0.5)
        # block projections: every row of the tensor takes part; normalised like an element-wise error of independent
        # elements (a block sums PBLOCK * C signed terms)
        pj = block_projections(a2).cpu()
        proj_rel = float((pj - b.proj.double()).abs().max()) / (scale * (PBLOCK * b.C) ** 0.5)
        flat = ew.flatten()
        if flat.numel() > 4_000_000:
            flat = flat[:4_000_000]
        return dict(rel_to_max=max(float(d.max()) / scale, bucket_rel, proj_rel), elementwise_p999=float(torch.quantile(flat, 0.999)),
                    elementwise_max=float(ew.max()), scale=scale, sampled_rows=int(b.idx.numel()), bucket_rel=bucket_rel,
                    proj_rel=proj_rel, rows_covered=b.N)
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = float(b.abs().max())
    d = (a - b).abs()
    ew = d / b.abs().clamp(min=1e-2 * scale)
    return dict(rel_to_max=float(d.max()) / scale, elementwise_p999=float(torch.quantile(ew.flatten()[:4_000_000], 0.999)),
                elementwise_max=float(ew.max()), scale=scale)
