"""Stand-in for the third-party `ocnn` package (absent from the container, not
vendored by the reference).  Used ONLY by tests/golden/make_golden.py in the
build container so the reference's own python files can be imported
unmodified.  Semantics live in oracle/octree.py (see its header: unpinned at
this boundary).  Never imported by the product or on the GPU box.
"""
from . import utils, octree, nn, modules, dataset  # noqa: F401
