"""Import harness for the read-only reference (build container only).

Puts the `ocnn` stand-in and /root/reference on sys.path and registers
import-only stubs for packages the reference imports at module top level but
never uses on the hot path (skimage, trimesh, plyfile, omegaconf, termcolor).
Nothing here is reference code.  `/root/reference` does not exist on the GPU
box, so this module is only ever used by make_golden.py / pin tests that are
skipped when the reference is absent.
"""
import os
import sys
import types

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def available():
    return os.path.isdir(os.path.join(REF, 'models'))


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return _wrap(v)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, _AttrDict):
        return _AttrDict(v)
    return v


def load_yaml(path):
    import yaml
    with open(path) as f:
        return _AttrDict(yaml.safe_load(f))


def setup():
    if not available():
        raise RuntimeError('reference tree not present')
    sys.dont_write_bytecode = True
    for p in (REF, os.path.join(HERE, '_standin'), ROOT):
        if p in sys.path:
            sys.path.remove(p)
    sys.path[:0] = [os.path.join(HERE, '_standin'), ROOT, REF]

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    sk = stub('skimage')
    sk.measure = stub('skimage.measure', marching_cubes=None)
    stub('trimesh')
    stub('plyfile', PlyData=object, PlyElement=object)
    stub('termcolor', colored=lambda s, *a, **k: s, cprint=lambda *a, **k: None)
    om = stub('omegaconf')

    class OmegaConf:
        load = staticmethod(load_yaml)
    om.OmegaConf = OmegaConf
