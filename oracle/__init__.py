"""CPU oracle for the OpenScene hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the reported CPU baseline.  The product package
``openscene_amd`` never imports this package and fails loudly when its HIP
library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * voxelize.py -- PINNED: checked bit-for-bit against outputs of the
    reference's own ``dataset/voxelizer.py`` / ``dataset/voxelization_utils.py``
    imported in the authoring container (``tests/golden/make_golden.py`` ->
    ``tests/golden/voxelize_*.npz``).
  * coords.py, sparse_ops.py -- PARITY UNPINNED against MinkowskiEngine: the
    arithmetic lives in the un-vendored third-party dependency
    NVIDIA/MinkowskiEngine (installed un-pinned from git HEAD,
    ``installation.md:37-39``), whose source and wheels are absent here.  They
    restate ME v0.5.4's published semantics and are
    anchored on the reference's call sites (``models/mink_unet.py:44-174``,
    ``models/resnet_base.py:73-118``) plus hand-derivable known-answer cases
    (``tests/test_oracle_kat.py``).  What IS pinned short of the engine itself:
    the reference's own ``models/mink_unet.py``, unmodified, executed on a
    stand-in engine made of torch's dense conv3d / conv_transpose3d
    (``tests/golden/dense_me.py`` + ``make_golden_unet.py`` ->
    ``tests/golden/unet_dense_ref.npz`` for MinkUNet18A, ``unet_dense_ref_34c.npz``
    for MinkUNet34C); ``unet_forward`` reproduces its
    outputs, every gradient and the running statistics to 1e-11 in float64
    (``tests/test_golden_unet.py``).  Layer plan, skip order, BN / residual
    placement and operator arithmetic are therefore pinned; the engine's
    CONVENTIONS (offset enumeration, even-kernel span, [Cin, Cout] 1x1
    kernels) remain restated from its documentation.
  * query.py -- restates ``run/evaluate.py:283-324`` with plain torch ops (the
    reference expression is itself plain torch, so this is the reference
    arithmetic evaluated on CPU).
"""
