"""Oracle for anchor projection and the box encoders, SURVEY.md 8(a) rows a5, a6, a12, a14.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference has a numpy branch (float64, host side, used for the RPN anchors)
and a TensorFlow branch (float32, inside the graph, used for proposals) of most
of these.  Functions here take ``dtype``: float64 restates the numpy branch,
float32 restates the TF branch op by op in float32.

Reference behaviour restated (paths relative to /root/reference/avod/core):
  anchor_projector.py:13-69     project_to_bev
  anchor_projector.py:72-156    project_to_image_space
  anchor_projector.py:159-251   tf_project_to_image_space
  anchor_projector.py:254-273   reorder_projected_boxes
  anchor_encoder.py:99-150      offset_to_anchor
  box_3d_encoder.py:188-227     tf_box_3d_to_anchor
  box_3d_encoder.py:230-322     anchors_to_box_3d
  box_4c_encoder.py:85-165      tf_box_3d_to_box_4c   (np twin :18-82)
  box_4c_encoder.py:305-458     tf_box_4c_to_box_3d   (np twin :168-302)
  box_4c_encoder.py:474-484     tf_offsets_to_box_4c
  orientation_encoder.py:20-34  tf_angle_vector_to_orientation
"""
import numpy as np


def project_to_bev(anchors, bev_extents, dtype=np.float64):
    """-> (corners [x1,z1,x2,z2] in metres from the top-left, same normalised)."""
    a = np.asarray(anchors, dtype=dtype)
    x, z = a[:, 0], a[:, 2]
    hx = a[:, 3] / dtype(2.0)
    hz = a[:, 5] / dtype(2.0)
    x_min, x_max = dtype(bev_extents[0][0]), dtype(bev_extents[0][1])
    z_min, z_max = dtype(bev_extents[1][0]), dtype(bev_extents[1][1])
    x1 = x - hx
    x2 = x + hx
    z1 = z_max - (z + hz)
    z2 = z_max - (z - hz)
    c = np.stack([x1, z1, x2, z2], axis=1)
    c = c - np.array([x_min, z_min, x_min, z_min], dtype=dtype)
    rng = np.array([x_max - x_min, z_max - z_min,
                    x_max - x_min, z_max - z_min], dtype=dtype)
    return c, c / rng


def _corners(a, dtype):
    x, y, z = a[:, 0], a[:, 1], a[:, 2]
    dx2 = a[:, 3] / dtype(2.)
    dy = a[:, 4]
    dz2 = a[:, 5] / dtype(2.)
    xc = np.stack([x + dx2, x + dx2, x - dx2, x - dx2,
                   x + dx2, x + dx2, x - dx2, x - dx2], axis=1)
    yc = np.stack([y, y, y, y, y - dy, y - dy, y - dy, y - dy], axis=1)
    zc = np.stack([z + dz2, z - dz2, z - dz2, z + dz2,
                   z + dz2, z - dz2, z - dz2, z + dz2], axis=1)
    return xc, yc, zc


def project_to_image_space(anchors, p2, image_shape_hw, dtype=np.float64):
    """8 corners . P2 -> min/max box [u1,v1,u2,v2] px and normalised by (W,H).

    dtype=float64: numpy branch; results are returned as float32 like the
    reference does (anchor_projector.py:155-156).  dtype=float32: TF branch
    (tf.matmul in float32: sum over the 4 terms in index order)."""
    a = np.asarray(anchors, dtype=dtype)
    if a.shape[1] != 6:
        raise ValueError("Invalid shape for anchors {}, should be "
                         "(N, 6)".format(a.shape[1]))
    p = np.asarray(p2, dtype=dtype)
    xc, yc, zc = _corners(a, dtype)
    if dtype == np.float64:
        hom = np.stack([xc.reshape(-1), yc.reshape(-1), zc.reshape(-1),
                        np.ones(xc.size)], axis=0)
        uvw = np.dot(p, hom)
    else:
        xf, yf, zf = xc.reshape(-1), yc.reshape(-1), zc.reshape(-1)
        uvw = np.stack([(p[r, 0] * xf + p[r, 1] * yf) + p[r, 2] * zf + p[r, 3]
                        for r in range(3)], axis=0).astype(dtype)
    with np.errstate(divide='ignore', invalid='ignore'):
        u = (uvw[0] / uvw[2]).reshape(-1, 8)
        v = (uvw[1] / uvw[2]).reshape(-1, 8)
    box = np.stack([u.min(axis=1), v.min(axis=1),
                    u.max(axis=1), v.max(axis=1)], axis=1)
    h, w = dtype(image_shape_hw[0]), dtype(image_shape_hw[1])
    norm = box / np.array([w, h, w, h], dtype=dtype)
    return box.astype(np.float32), norm.astype(np.float32)


def reorder_projected_boxes(box):
    """[x1,y1,x2,y2] -> [y1,x1,y2,x2] (anchor_projector.py:254-273)."""
    b = np.asarray(box)
    return b[:, [1, 0, 3, 2]]


def offset_to_anchor(anchors, offsets, dtype=np.float64):
    """anchor_encoder.py:99-150: centre += t * dim ; dim = exp(log(dim) + t)."""
    a = np.asarray(anchors, dtype=dtype)
    t = np.asarray(offsets, dtype=dtype)
    out = np.empty_like(a)
    out[:, 0] = (t[:, 0] * a[:, 3]) + a[:, 0]
    out[:, 1] = (t[:, 1] * a[:, 4]) + a[:, 1]
    out[:, 2] = (t[:, 2] * a[:, 5]) + a[:, 2]
    out[:, 3] = np.exp(np.log(a[:, 3]) + t[:, 3])
    out[:, 4] = np.exp(np.log(a[:, 4]) + t[:, 4])
    out[:, 5] = np.exp(np.log(a[:, 5]) + t[:, 5])
    return out


def anchors_to_box_3d(anchors, fix_lw=False, dtype=np.float64):
    """box_3d_encoder.py:230-322.  l <- dim_x, w <- dim_z, h <- dim_y, ry = 0;
    with fix_lw, rows with w > l get l/w swapped and ry = -pi/2."""
    a = np.asarray(anchors, dtype=dtype)
    out = np.zeros((len(a), 7), dtype=dtype)
    out[:, 0:3] = a[:, 0:3]
    out[:, 3] = a[:, 3]
    out[:, 4] = a[:, 5]
    out[:, 5] = a[:, 4]
    if fix_lw:
        swap = out[:, 4] > out[:, 3]
        l = out[:, 3].copy()
        out[swap, 3] = out[swap, 4]
        out[swap, 4] = l[swap]
        out[swap, 6] = dtype(-np.pi / 2)
    return out


def box_3d_to_anchor_ortho(boxes_3d, dtype=np.float32):
    """tf_box_3d_to_anchor (box_3d_encoder.py:188-227): snap ry to the nearest
    multiple of pi/2 (round half to even, like tf.round and np.round)."""
    b = np.asarray(boxes_3d, dtype=dtype).reshape(-1, 7)
    half_pi = dtype(np.pi / 2)
    ry = np.round(b[:, 6] / half_pi) * half_pi
    c = np.abs(np.cos(ry))
    s = np.abs(np.sin(ry))
    out = np.empty((len(b), 6), dtype=dtype)
    out[:, 0:3] = b[:, 0:3]
    out[:, 3] = b[:, 3] * c + b[:, 4] * s
    out[:, 4] = b[:, 5]
    out[:, 5] = b[:, 4] * c + b[:, 3] * s
    return out


def box_3d_to_box_4c(boxes_3d, ground_plane, dtype=np.float32):
    """tf_box_3d_to_box_4c (box_4c_encoder.py:85-165), vectorised.
    -> (N,10) [x1..x4, z1..z4, h1, h2]."""
    b = np.asarray(boxes_3d, dtype=dtype).reshape(-1, 7)
    gp = np.asarray(ground_plane, dtype=dtype)
    anc = box_3d_to_anchor_ortho(b, dtype)
    cx, cy, cz = anc[:, 0], anc[:, 1], anc[:, 2]
    hx = anc[:, 3] / dtype(2)
    hz = anc[:, 5] / dtype(2)
    xs = np.stack([hx, hx, -hx, -hx], axis=1)
    zs = np.stack([hz, -hz, -hz, hz], axis=1)
    half_pi = dtype(np.pi / 2)
    ry = b[:, 6]
    d = ry - np.round(ry / half_pi) * half_pi
    co, si = np.cos(d)[:, None], np.sin(d)[:, None]
    # rows of tr_mat^T . [x; z; 1]  (matmul with transpose_a, :139-148)
    px = (co * xs + si * zs) + cx[:, None]
    pz = (-si * xs + co * zs) + cz[:, None]
    ground_y = -(gp[0] * cx + gp[2] * cz + gp[3]) / gp[1]
    h1 = ground_y - cy
    h2 = h1 + anc[:, 4]
    return np.concatenate([px, pz, h1[:, None], h2[:, None]],
                          axis=1).astype(dtype)


def offsets_to_box_4c(boxes_4c, offsets, dtype=np.float32):
    """box_4c_encoder.py:474-484."""
    return np.asarray(boxes_4c, dtype=dtype) + np.asarray(offsets, dtype=dtype)


def _box_info(vec, mag, p, mid, dtype):
    """calculate_box_3d_info (box_4c_encoder.py:305-366) for all rows."""
    nrm = vec / mag[:, None]
    rel = [q - mid for q in p]
    ls = np.stack([(r * nrm).sum(axis=1) for r in rel], axis=1)
    min_l = ls.min(axis=1, keepdims=True)
    max_l = ls.max(axis=1, keepdims=True)
    ortho = np.stack([-nrm[:, 1], nrm[:, 0]], axis=1)
    ws = np.stack([(r * ortho).sum(axis=1) for r in rel], axis=1)
    min_w = ws.min(axis=1)
    max_w = ws.max(axis=1)
    w_diff = (max_w + min_w)[:, None]
    ry = -np.arctan2(vec[:, 1], vec[:, 0])
    cen = mid + nrm * (min_l + max_l) / dtype(2.0) + ortho * w_diff
    return cen, (max_l - min_l)[:, 0], max_w - min_w, ry


def box_4c_to_box_3d(boxes_4c, ground_plane, dtype=np.float32):
    """tf_box_4c_to_box_3d (box_4c_encoder.py:369-458): both midline
    candidates are evaluated and blended with 0/1 float masks; the 34->12
    candidate wins only when strictly longer."""
    b = np.asarray(boxes_4c, dtype=dtype).reshape(-1, 10)
    gp = np.asarray(ground_plane, dtype=dtype)
    cor = b[:, 0:8].reshape(-1, 2, 4)
    p = [cor[:, :, k] for k in range(4)]
    m12 = (p[0] + p[1]) / dtype(2.0)
    m23 = (p[1] + p[2]) / dtype(2.0)
    m34 = (p[2] + p[3]) / dtype(2.0)
    m14 = (p[0] + p[3]) / dtype(2.0)
    va = m12 - m34
    vb = m14 - m23
    ma = np.sqrt((va * va).sum(axis=1))
    mb = np.sqrt((vb * vb).sum(axis=1))
    with np.errstate(divide='ignore', invalid='ignore'):
        ca, la, wa, ra = _box_info(va, ma, p, m34, dtype)
        cb, lb, wb, rb = _box_info(vb, mb, p, m23, dtype)
    fa = (ma > mb).astype(dtype)
    fb = dtype(1) - fa
    cen = ca * fa[:, None] + cb * fb[:, None]
    length = la * fa + lb * fb
    width = wa * fa + wb * fb
    ry = ra * fa + rb * fb
    ground_y = -(gp[0] * cen[:, 0] + gp[2] * cen[:, 1] + gp[3]) / gp[1]
    cy = ground_y - b[:, 8]
    return np.stack([cen[:, 0], cy, cen[:, 1], length, width,
                     b[:, 9] - b[:, 8], ry], axis=1).astype(dtype)


def angle_vector_to_orientation(angle_vectors, dtype=np.float32):
    """orientation_encoder.py:20-34."""
    v = np.asarray(angle_vectors, dtype=dtype)
    return np.arctan2(v[:, 1], v[:, 0])
