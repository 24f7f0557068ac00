"""CPU-only checks of the boundary: the C-ABI library loads, exports every symbol that
include/dodt_hip.h declares (and nothing is bound that the header lacks), host-side
tables match the oracle, and argument errors surface without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from dodt_amd import _lib, config
from dodt_amd.core import anchor_filter
from dodt_amd.core.anchor_generators import grid_anchor_3d_generator as gen
from oracle import anchors as oanchors
from oracle import points as opoints

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = config.PYRAMID_DODT


def _declared():
    text = open(os.path.join(ROOT, 'include', 'dodt_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dodt_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.SIGNATURES) == names        # the binding covers the header exactly
    assert _lib.load().dodt_version() >= 1


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from dodt_amd import device
    with pytest.raises(_lib.DodtError):
        device.Context(0)


def test_host_anchor_tables_match_oracle():
    boxes = gen.tile_anchors_3d(C['area_extents'], C['anchor_sizes'], C['anchor_stride'],
                                C['ground_plane'])
    assert np.array_equal(boxes, oanchors.tile_anchors_3d(
        C['area_extents'], C['anchor_sizes'], C['anchor_stride'], C['ground_plane']))
    anchors = gen.box_3d_to_anchor(boxes)
    assert np.array_equal(anchors, oanchors.box_3d_to_anchor(boxes))
    cells, nx, nz = gen.anchor_grid_cells(anchors, C['area_extents'], C['voxel_size'])
    assert (nx, nz) == (800, 700) and cells.dtype == np.int32
    # same indices as the reference's map_to_index on float32 corners
    vox = opoints.voxelize_2d(np.array([[0.05, 0.0, 0.05], [1.0, 0.0, 1.0]]), C['voxel_size'],
                              extents=C['area_extents'])
    lo = np.stack([anchors[:, 0] - anchors[:, 3] / 2., anchors[:, 2] - anchors[:, 5] / 2.],
                  1).astype(np.float32)
    assert np.array_equal(cells[:, :2], opoints.map_to_index(vox, lo))


def test_occupancy_bit_packing_layout():
    occ = np.zeros((800, 700), bool)
    occ[0, 0] = occ[33, 5] = occ[799, 699] = True
    bits = anchor_filter.pack_occupancy(occ)
    assert bits.shape == (700, 25) and bits.dtype == np.uint32
    assert bits[0, 0] == 1 and bits[5, 1] == 2 and bits[699, 24] == (1 << 31)
    assert bits.sum(dtype=np.uint64) == 1 + 2 + (1 << 31)


def test_config_scalars_are_float32_rounded():
    assert C['voxel_size'] == 0.10000000149011612
    assert C['height_lo'] == -0.20000000298023224
    assert C['height_hi'] == 2.299999952316284
