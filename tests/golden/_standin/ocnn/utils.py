from oracle.octree import scatter_add, cumsum  # noqa: F401
