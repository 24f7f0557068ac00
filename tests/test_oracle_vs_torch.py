"""The TF-op half of the oracle has no reference fixture (TensorFlow cannot be installed
here), so its conv arithmetic is cross-checked against an independent implementation that IS
available: torch's CPU convolutions, configured to TF-1.3 'SAME' semantics
(SURVEY appendix A.5).  Float32 with different summation orders: 1e-5 of the output scale."""
import numpy as np
import pytest

from oracle import tfops

torch = pytest.importorskip('torch')
F = torch.nn.functional


def _close(a, b, rel=1e-5):
    scale = np.abs(b).max() + 1e-12
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= rel * scale, (np.abs(a - b).max(), scale)


@pytest.mark.parametrize('h,w,cin,cout', [(9, 12, 5, 7), (16, 8, 32, 16), (5, 5, 1, 1)])
def test_conv2d_same_matches_torch(h, w, cin, cout):
    rng = np.random.default_rng(h * w + cin)
    x = rng.normal(size=(h, w, cin)).astype(np.float32)
    k = rng.normal(size=(3, 3, cin, cout)).astype(np.float32)          # HWIO
    want = F.conv2d(torch.tensor(x).permute(2, 0, 1)[None],
                    torch.tensor(k).permute(3, 2, 0, 1), padding=1)[0].permute(1, 2, 0).numpy()
    _close(tfops.conv2d_same(x, k), want)


@pytest.mark.parametrize('h,w,cin,cout', [(6, 7, 8, 4), (11, 5, 3, 6)])
def test_conv2d_transpose_stride2_same_matches_torch(h, w, cin, cout):
    """TF 'SAME' 3x3 stride-2 transposed conv = gradient of the forward conv whose padding is
    (0 before, 1 after): out[2i + ky, 2j + kx] += x[i, j] w[ky, kx], cropped to 2H x 2W --
    torch's conv_transpose2d with padding 0, cropped the same way."""
    rng = np.random.default_rng(h + 10 * w)
    x = rng.normal(size=(h, w, cin)).astype(np.float32)
    k = rng.normal(size=(3, 3, cout, cin)).astype(np.float32)          # TF: (kh, kw, out, in)
    full = F.conv_transpose2d(torch.tensor(x).permute(2, 0, 1)[None],
                              torch.tensor(k).permute(3, 2, 0, 1), stride=2)[0]
    want = full[:, :2 * h, :2 * w].permute(1, 2, 0).numpy()
    _close(tfops.conv2d_transpose_s2_same(x, k), want)


def test_batch_norm_relu_and_pool_match_torch():
    rng = np.random.default_rng(3)
    x = rng.normal(size=(10, 14, 6)).astype(np.float32)
    beta, mean = rng.normal(size=6).astype(np.float32), rng.normal(size=6).astype(np.float32)
    var = rng.uniform(0.5, 2, 6).astype(np.float32)
    t = torch.tensor(x).permute(2, 0, 1)[None]
    want = F.relu(F.batch_norm(t, torch.tensor(mean), torch.tensor(var), None,
                               torch.tensor(beta), False, 0.0, 1e-3))       # slim: no gamma
    _close(tfops.bn_relu(x, beta, mean, var), want[0].permute(1, 2, 0).numpy())
    pool = F.max_pool2d(t, 2)[0].permute(1, 2, 0).numpy()
    assert np.array_equal(tfops.max_pool_2x2(x), pool)
    odd = rng.normal(size=(7, 9, 2)).astype(np.float32)                    # VALID drops the rest
    assert np.array_equal(tfops.max_pool_2x2(odd),
                          F.max_pool2d(torch.tensor(odd).permute(2, 0, 1)[None], 2)[0]
                          .permute(1, 2, 0).numpy())


def test_fc_and_softmax_match_torch():
    rng = np.random.default_rng(4)
    from oracle import heads as oheads
    x = rng.normal(size=(9, 20)).astype(np.float32)
    w = rng.normal(size=(20, 5)).astype(np.float32)
    b = rng.normal(size=5).astype(np.float32)
    want = F.relu(F.linear(torch.tensor(x), torch.tensor(w).T, torch.tensor(b))).numpy()
    _close(oheads.fc(x, w, b, True), want)
    logits = rng.normal(size=(50, 2)).astype(np.float32)
    _close(tfops.softmax2(logits), F.softmax(torch.tensor(logits), 1).numpy(), 1e-6)


def test_crop_and_resize_matches_torch_grid_sample_inside_the_image():
    """tf.image.crop_and_resize samples at in_y = y1 (H-1) + iy (y2-y1)(H-1)/(ch-1): the corner-aligned
    bilinear grid, which is torch's grid_sample(align_corners=True) on the normalised coordinates
    2 in / (size-1) - 1.  (The two differ only in how they extrapolate: TF zeroes a sample whose centre
    lies outside the image, torch blends with zeros; boxes here stay inside.)"""
    rng = np.random.default_rng(11)
    h, w, c, n, ch, cw = 23, 31, 5, 40, 7, 7
    img = rng.normal(size=(h, w, c)).astype(np.float32)
    y1, x1 = rng.uniform(0, 0.6, n), rng.uniform(0, 0.6, n)
    boxes = np.stack([y1, x1, y1 + rng.uniform(0.05, 0.39, n), x1 + rng.uniform(0.05, 0.39, n)], 1).astype(np.float32)
    boxes[0] = [0, 0, 1, 1]
    got = tfops.crop_and_resize(img, boxes, ch, cw)
    iy = np.arange(ch) / (ch - 1)
    ix = np.arange(cw) / (cw - 1)
    gy = 2 * (boxes[:, None, 0] + iy[None] * (boxes[:, 2] - boxes[:, 0])[:, None]) - 1        # (n, ch)
    gx = 2 * (boxes[:, None, 1] + ix[None] * (boxes[:, 3] - boxes[:, 1])[:, None]) - 1        # (n, cw)
    grid = np.stack([np.broadcast_to(gx[:, None, :], (n, ch, cw)), np.broadcast_to(gy[:, :, None], (n, ch, cw))], -1)
    t = torch.tensor(img).permute(2, 0, 1)[None].expand(n, -1, -1, -1)
    want = F.grid_sample(t, torch.tensor(grid, dtype=torch.float32), mode='bilinear', padding_mode='zeros',
                         align_corners=True).permute(0, 2, 3, 1).numpy()
    _close(got, want, 2e-5)


def test_nms_matches_a_quadratic_restatement():
    """tf.image.non_max_suppression as three nested python facts -- sort by score, keep a box unless an
    earlier kept one overlaps it by more than the threshold, stop at max_output -- against the oracle's
    vectorised form, incl. corner order normalisation and the zero-area rule of TF's IoU."""
    rng = np.random.default_rng(12)
    n = 400
    c = rng.uniform(0, 1, size=(n, 2))
    hw = rng.uniform(0.0, 0.15, size=(n, 2))
    boxes = np.concatenate([c - hw, c + hw], 1).astype(np.float32)
    boxes[::7] = boxes[::7][:, [2, 3, 0, 1]]              # flipped corners
    boxes[5] = [0.3, 0.3, 0.3, 0.5]                        # zero area
    scores = rng.uniform(size=n).astype(np.float32)

    def iou(a, b):
        ay0, ay1, ax0, ax1 = min(a[0], a[2]), max(a[0], a[2]), min(a[1], a[3]), max(a[1], a[3])
        by0, by1, bx0, bx1 = min(b[0], b[2]), max(b[0], b[2]), min(b[1], b[3]), max(b[1], b[3])
        aa, ab = np.float32(ay1 - ay0) * np.float32(ax1 - ax0), np.float32(by1 - by0) * np.float32(bx1 - bx0)
        if aa <= 0 or ab <= 0:
            return np.float32(0)
        ih = max(np.float32(0), np.float32(min(ay1, by1) - max(ay0, by0)))
        iw = max(np.float32(0), np.float32(min(ax1, bx1) - max(ax0, bx0)))
        inter = np.float32(ih * iw)
        return np.float32(inter / np.float32(np.float32(aa + ab) - inter))
    for thr, k in ((0.5, 60), (0.01, 400), (0.8, 100)):
        order = sorted(range(n), key=lambda i: (-scores[i], i))
        keep = []
        for i in order:
            if len(keep) >= k:
                break
            if all(not (iou(boxes[i], boxes[j]) > np.float32(thr)) for j in keep):
                keep.append(i)
        assert list(tfops.non_max_suppression_fast(boxes, scores, k, thr)) == keep


def test_resize_bilinear_matches_torch_grid_sample_at_the_legacy_coordinates():
    """TF-1.3's resize_bilinear (align_corners=False, no half-pixel centres) samples at src = dst * in / out,
    clamped at the last row / column: torch's grid_sample(align_corners=True, padding_mode='border') on exactly
    those coordinates is an independent bilinear interpolation."""
    rng = np.random.default_rng(5)
    h, w, c, oh, ow = 19, 45, 3, 12, 40          # the image path scales down (375x1242 -> 360x1200)
    img = rng.uniform(0, 255, size=(h, w, c)).astype(np.float32)
    got = tfops.resize_bilinear(img, oh, ow)
    sy = np.arange(oh, dtype=np.float64) * h / oh
    sx = np.arange(ow, dtype=np.float64) * w / ow
    gy, gx = 2 * sy / (h - 1) - 1, 2 * sx / (w - 1) - 1
    grid = np.stack([np.broadcast_to(gx[None, :], (oh, ow)), np.broadcast_to(gy[:, None], (oh, ow))], -1)[None]
    want = F.grid_sample(torch.tensor(img, dtype=torch.float64).permute(2, 0, 1)[None], torch.tensor(grid),
                         mode='bilinear', padding_mode='border', align_corners=True)[0].permute(1, 2, 0).numpy()
    _close(got, want, 2e-6)
    up = tfops.resize_bilinear(img, 2 * h, 2 * w)     # the plain-VGG path scales up: the clamp at the edge
    assert np.array_equal(up[::2, ::2], img)


def test_correlation_matches_an_unfold_formulation():
    """The same op written the other way round: for every displacement of the (2r+1)^2 grid, the padded second
    map shifted by it times the first, mean over channels (float64) -- against the oracle's restatement of the
    CUDA kernel's index arithmetic."""
    rng = np.random.default_rng(9)
    h, w, c, d, s2, pad = 13, 17, 8, 4, 2, 4
    a = rng.normal(size=(h, w, c)).astype(np.float32)
    b = rng.normal(size=(h, w, c)).astype(np.float32)
    got = tfops.correlation(a, b, d, s2, pad)
    r = d // s2
    ta = torch.tensor(a, dtype=torch.float64).permute(2, 0, 1)
    tb = F.pad(torch.tensor(b, dtype=torch.float64).permute(2, 0, 1), (d, d, d, d))      # zeros around
    want = []
    for p in range(-r, r + 1):
        for o in range(-r, r + 1):
            shifted = tb[:, d + p * s2:d + p * s2 + h, d + o * s2:d + o * s2 + w]
            want.append((ta * shifted).mean(0))
    want = torch.stack(want, -1).numpy()
    assert got.shape == want.shape == (h, w, (2 * r + 1) ** 2)     # pad == max_displacement: same size
    _close(got, want, 2e-6)
