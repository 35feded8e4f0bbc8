"""CPU: the restated Pillow coefficient tables (loans_amd/common/datasets/resample.py) reproduce Pillow's LANCZOS resize
bit for bit -- against the committed vectors (tests/golden/resample_lanczos.npz, made by Pillow) and against the Pillow
installed here.  This is the one piece of the path whose third-party implementation IS present, so its parity is
pinned by the real thing."""
import os

import numpy as np
import pytest

from tests.resample_util import two_pass_u8

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'resample_lanczos.npz')


def _cases():
    g = np.load(GOLDEN)
    n = 0
    while 'src_%d' % n in g.files:
        yield g['src_%d' % n], g['dst_%d' % n], g['bil_%d' % n]
        n += 1


def test_tables_reproduce_the_golden_vectors():
    cases = list(_cases())
    assert len(cases) >= 7
    for src, dst, bil in cases:
        np.testing.assert_array_equal(two_pass_u8(src, dst.shape[0], dst.shape[1]), dst)
        np.testing.assert_array_equal(two_pass_u8(src, bil.shape[0], bil.shape[1], 'bilinear'), bil)


@pytest.mark.parametrize("shape", [(37, 53, 20, 24), (64, 64, 224, 224), (480, 640, 224, 224), (100, 30, 30, 100),
                                   (7, 5, 3, 2), (300, 1000, 75, 75), (50, 60, 50, 33), (1, 1, 4, 4), (5, 5, 1, 1)])
def test_tables_against_installed_pillow(shape):
    from PIL import Image
    H, W, oh, ow = shape
    a = np.random.RandomState(H * 1000 + W).randint(0, 256, (H, W, 3)).astype(np.uint8)
    ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.LANCZOS))
    np.testing.assert_array_equal(two_pass_u8(a, oh, ow), ref)
    ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
    np.testing.assert_array_equal(two_pass_u8(a, oh, ow, 'bilinear'), ref)


def test_table_properties():
    from loans_amd.common.datasets.resample import PRECISION_BITS, lanczos_coeffs
    b, k, ks = lanczos_coeffs(640, 224)
    assert k.shape == (224, ks) and b.shape == (224, 2) and ks == 2 * 9 + 1
    assert (b[:, 0] >= 0).all() and (b[:, 0] + b[:, 1] <= 640).all() and (b[:, 1] <= ks).all()
    # normalised windows: the fixed-point weights of a row sum to 1 within rounding
    assert np.abs(k.sum(axis=1) - (1 << PRECISION_BITS)).max() <= ks
    # identity size: a pure copy
    b, k, ks = lanczos_coeffs(17, 17)
    assert ks == 7 and all(k[i].max() == (1 << PRECISION_BITS) and (k[i] != 0).sum() == 1 for i in range(17))
