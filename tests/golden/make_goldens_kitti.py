#!/usr/bin/env python3
"""Golden vectors for the detection post-processing: the reference's own
box_3d_projector.project_to_image_space (avod/core/box_3d_projector.py:86-159), run in the
build container with the same inert tensorflow / cv2 stand-ins as make_goldens.py.

Run:  python tests/golden/make_goldens_kitti.py     (needs /root/reference; writes kitti_format.npz)
Stored: seeded random box_3d rows, the calibration P2 of the bundled tracking calib file,
the image size, and for every box the reference's [x1,y1,x2,y2] or NaNs where it returned None.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402


def main():
    mg._import_reference()
    from avod.core import box_3d_projector
    rng = np.random.default_rng(20260404)
    n = 400
    boxes = np.stack([rng.uniform(-45, 45, n), rng.uniform(0.8, 2.2, n), rng.uniform(-5, 75, n),
                      rng.uniform(1.0, 6.0, n), rng.uniform(0.5, 2.5, n), rng.uniform(1.0, 2.5, n),
                      rng.uniform(-np.pi, np.pi, n)], 1)
    boxes[:8, 2] = rng.uniform(0.2, 3.0, 8)           # very close: huge projections
    p2 = np.array([[721.5377, 0.0, 609.5593, 44.85728],
                   [0.0, 721.5377, 172.854, 0.2163791],
                   [0.0, 0.0, 1.0, 0.002745884]])
    image_size = (1242, 375)
    out = np.full((n, 4), np.nan)
    out_after = np.full((n, 4), np.nan)
    for i in range(n):
        r = box_3d_projector.project_to_image_space(boxes[i].copy(), p2, truncate=True,
                                                    image_size=image_size)
        if r is not None:
            out[i] = r
        r = box_3d_projector.project_to_image_space(boxes[i].copy(), p2, truncate=True,
                                                    image_size=image_size,
                                                    discard_before_truncation=False)
        if r is not None:
            out_after[i] = r
    raw = np.stack([box_3d_projector.project_to_image_space(boxes[i].copy(), p2)
                    for i in range(n)])
    np.savez_compressed(os.path.join(mg.HERE, 'kitti_format.npz'), boxes_3d=boxes, p2=p2,
                        image_size=np.asarray(image_size), img_boxes=out,
                        img_boxes_discard_after=out_after, img_boxes_raw=raw)
    print('kept %d / %d (discard before), %d (after)' % (np.isfinite(out[:, 0]).sum(), n,
                                                        np.isfinite(out_after[:, 0]).sum()))


if __name__ == '__main__':
    main()
