"""MI355X-native PointNet++ set-abstraction / feature-propagation hot path.

Drop-in counterpart of the reference's ``Modules/PointNet2`` (pointnet2_utils / blocks / PointNet2) whose
device work goes through the C-ABI HIP library ``libpn2hip.so`` (see include/pn2_hip.h).  The directory name
is not a Python identifier; import it under the alias ``pn2_amd`` (tests/helpers.py:load_pkg, or put
``PointNet2/`` in place of the reference's ``Modules/PointNet2`` -- see INTEGRATION.md).

Importing this package never touches the GPU and never loads the HIP library; the first op call does, and
fails loudly if the library is missing (there is no CPU fallback).
"""
__version__ = "0.1.0"
