from oracle.octree import octree2voxel, octree_pad  # noqa: F401
