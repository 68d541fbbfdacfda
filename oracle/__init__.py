"""CPU oracle for the OctFusion denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product path (``octfusion_amd``) never imports this package and fails
loudly when its HIP library is missing.

What it is: a plain torch-CPU / numpy restatement of the reference's algorithm
for SURVEY.md section 8(a) rows a1-a16, written function-by-function with the
reference file:line each function follows.  Floating point work uses the SAME
op sequence as the reference (index -> zeros -> scatter_add_ x2 -> div -> mm;
three-scatter group norm; per-batch-element embedding add) so that timing it is
a fair stand-in for the reference's CPU path.

Parity pin: the reference has no tests / golden vectors of its own (SURVEY.md
section 4).  The oracle is pinned against outputs of the reference's OWN python
files (``models/networks/modules.py``, ``dual_octree.py``, ``graph_unet_*.py``,
``util_dualoctree.py``, ``octfusion_model_union.py::sample_loop``,
``graph_vae.py``) imported unmodified in the build container by
``tests/golden/make_golden.py``; the resulting vectors are committed under
``tests/golden/``.  The third-party ``ocnn`` package the reference imports is
absent from the container and un-vendored (requirements.txt:1, unpinned), so
at THAT boundary (octree container + Morton key codec + octree2voxel/pad) the
semantics are restated from ocnn-pytorch's public API and the invariants the
reference relies on -- parity at the ocnn boundary is
UNPINNED; everything above it is pinned by the golden vectors.

Training side: ``oracle/loss.py`` + ``vae.forward_train`` (the VAE objective and its autograd
gradients) are PINNED by ``tests/golden/g_vae_train.pt`` -- the reference's own GraphVAE.forward +
loss.geometry_loss + backward, every named loss and every parameter gradient.  ``oracle/points.py`` (point cloud ->
octree) restates ocnn's published build_octree / merge_octrees: that is the ocnn boundary again, so it is UNPINNED;
it is cross-checked against a brute-force occupancy pyramid and through the reference's octree <-> split-code round
trip (which IS pinned).
"""
