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
