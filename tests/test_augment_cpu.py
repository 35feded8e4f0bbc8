"""CPU: documented semantics of the three imgaug operations the reference's training dataset applies
(common/datasets/image_dataset.py:57-70: Fliplr, AddToHueAndSaturation(per_channel), CropAndPad(percent, pad_mode constant /
edge)) on CONSTRUCTED images, for the restated form in loans_amd/common/datasets/augment.py.  imgaug and OpenCV are not
installable here, so this pins the restatement to the operations' published behaviour (8-bit HSV with H in [0, 180), additions
that saturate, hue that wraps, sides in imgaug's (top, right, bottom, left) order) -- not to imgaug's bytes: PARITY UNPINNED for
this branch.  The GPU stages are compared with this NumPy form byte for byte in tests/test_gpu_resample.py."""
import numpy as np

from loans_amd.common.datasets import augment as A

NONE = [[0] * 8] * 3


def _one(row):
    return [row] + NONE[:2]


def _hsv(rgb):
    h, s, v = A._rgb2hsv(np.asarray(rgb, np.uint8).reshape(1, 1, 3))
    return int(h[0, 0]), int(s[0, 0]), int(v[0, 0])


def test_hsv_of_the_primaries_is_opencvs_8_bit_convention():
    """H in [0, 180) (degrees / 2), S and V in [0, 255]"""
    assert _hsv((255, 0, 0)) == (0, 255, 255)
    assert _hsv((0, 255, 0)) == (60, 255, 255)
    assert _hsv((0, 0, 255)) == (120, 255, 255)
    assert _hsv((255, 255, 0)) == (30, 255, 255)
    assert _hsv((0, 255, 255)) == (90, 255, 255)
    assert _hsv((255, 0, 255)) == (150, 255, 255)
    assert _hsv((128, 128, 128)) == (0, 0, 128)
    assert _hsv((0, 0, 0)) == (0, 0, 0)
    assert _hsv((200, 100, 100)) == (0, 128, 200)              # S = 255 * (200 - 100) / 200, rounded
    # and back: every (h, s, v) the forward map produces for these returns the colour
    for rgb in [(255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 0), (40, 200, 90), (128, 128, 128)]:
        h, s, v = A._rgb2hsv(np.asarray(rgb, np.uint8).reshape(1, 1, 3))
        back = A._hsv2rgb(h, s, v)[0, 0]
        assert np.abs(back.astype(int) - np.asarray(rgb)).max() <= 2, (rgb, back)


def test_hue_wraps_at_180():
    """magenta has H = 150: +20 stays below 180 (170: towards red), +40 wraps to 10 (red-orange), never to a clipped 179.
    imgaug 0.2.6 (requirements.txt) implements the operation as WithColorspace(HSV, WithChannels([0, 1], Add(v))), and its Add
    CLIPS the uint8 channel to [0, 255]: a sum above 179 survives the addition and wraps in the conversion back (OpenCV takes
    the hue modulo 180), a sum below 0 is clipped to 0 -- a negative shift of red (H = 0) leaves it red"""
    magenta = np.zeros((3, 3, 3), np.uint8)
    magenta[..., 0] = magenta[..., 2] = 240
    out20 = A.apply_host(magenta, _one([2, 20, 0, 0, 0, 0, 0, 0]))[0, 0]
    assert _hsv(out20)[0] in (169, 170, 171)
    out40 = A.apply_host(magenta, _one([2, 40, 0, 0, 0, 0, 0, 0]))[0, 0]
    assert _hsv(out40)[0] in (9, 10, 11) and out40[0] == 240 and out40[2] == 0       # red-orange: R at full value, no blue
    red = np.zeros((2, 2, 3), np.uint8)
    red[..., 0] = 250
    back = A.apply_host(red, _one([2, -20, 0, 0, 0, 0, 0, 0]))[0, 0]
    np.testing.assert_array_equal(back, [250, 0, 0])                                  # clipped at H = 0, not wrapped to 160
    orange = np.zeros((2, 2, 3), np.uint8)
    orange[..., 0], orange[..., 1] = 240, 120                                         # H = 15
    assert _hsv(A.apply_host(orange, _one([2, -10, 0, 0, 0, 0, 0, 0]))[0, 0])[0] in (4, 5, 6)
    np.testing.assert_array_equal(A.apply_host(orange, _one([2, -20, 0, 0, 0, 0, 0, 0]))[0, 0], [240, 0, 0])   # 15 - 20 -> 0
    # on a random frame: every saturated pixel's hue moves by the shift modulo 180 (the reference draws shifts in [-20, 20], so
    # the uint8 sum never reaches the upper clip at 255), value and saturation stay
    img = np.random.RandomState(0).randint(0, 256, (6, 7, 3)).astype(np.uint8)
    h0, s0, v0 = A._rgb2hsv(img)
    h1, s1, v1 = A._rgb2hsv(A.apply_host(img, _one([2, 20, 0, 0, 0, 0, 0, 0])))
    dh = np.abs(h1 - (h0 + 20) % 180)
    assert (np.minimum(dh, 180 - dh)[s0 > 40] <= 2).all() and (h0 + 20 >= 180).any()
    assert np.abs(v1 - v0).max() <= 1 and np.abs(s1 - s0)[v0 > 60].max() <= 4


def test_saturation_saturates():
    """AddToHueAndSaturation adds in uint8 with clipping: a fully saturated colour is unchanged by +20, a weakly saturated one
    loses ALL colour under -20 (S clips at 0: a grey of the same value), value is never touched"""
    pure = np.zeros((2, 2, 3), np.uint8)
    pure[..., 1] = 180                                                    # S = 255
    np.testing.assert_array_equal(A.apply_host(pure, _one([2, 0, 20, 0, 0, 0, 0, 0])), pure)
    pale = np.full((2, 2, 3), 200, np.uint8)
    pale[..., 2] = 190                                                    # S = 255 * 10 / 200 = 13
    assert _hsv(pale[0, 0])[1] == 13
    grey = A.apply_host(pale, _one([2, 0, -20, 0, 0, 0, 0, 0]))
    np.testing.assert_array_equal(grey, np.full((2, 2, 3), 200, np.uint8))
    more = A.apply_host(pale, _one([2, 0, 20, 0, 0, 0, 0, 0]))[0, 0]
    assert _hsv(more)[1] in (32, 33, 34) and more.max() == 200            # V unchanged, S = 13 + 20
    # per_channel=True: the two shifts are independent parameters
    a = A.apply_host(pale, _one([2, 15, 0, 0, 0, 0, 0, 0]))
    b = A.apply_host(pale, _one([2, 0, 15, 0, 0, 0, 0, 0]))
    assert not np.array_equal(a, b)


def test_pad_mode_edge_against_constant():
    """CropAndPad pads (positive amounts) with zeros or with the border pixels and resizes back to the frame's size: on a frame
    whose border rows differ from its interior the two modes differ exactly in the padded band"""
    H, W = 20, 40
    img = np.full((H, W, 3), 100, np.uint8)
    img[0] = 250                                 # top border row bright
    img[:, -1] = 10                              # right border column dark
    pad = [3, 3, 2, 0, 4]                        # op 3 = crop-and-pad; (top, right, bottom, left) = imgaug's order
    const = A.apply_host(img, _one(pad + [0, 0, 0])).astype(int)
    edge = A.apply_host(img, _one(pad + [1, 0, 0])).astype(int)
    assert const.shape == edge.shape == (H, W, 3)
    # constant: the padded top band is black; edge: it repeats the bright top row
    assert const[0, W // 2, 0] == 0 and edge[0, W // 2, 0] >= 240
    # the left band (4 of 44 virtual columns): black against the interior's 100
    assert const[H // 2, 0, 0] == 0 and edge[H // 2, 0, 0] == 100
    # the right band repeats the dark border column under edge fill
    assert const[H // 2, -1, 0] == 0 and edge[H // 2, -1, 0] == 10
    # no padding at the bottom: the last row is image content in both
    assert const[-1, W // 2, 0] == edge[-1, W // 2, 0] == 100
    # away from the bands the two modes agree
    np.testing.assert_array_equal(const[6:, 6:-4], edge[6:, 6:-4])


def test_crop_sides_and_order():
    """negative amounts crop: each side removes ITS rows / columns (top, right, bottom, left), the rest is stretched back"""
    H, W = 32, 32
    img = np.zeros((H, W, 3), np.uint8)
    img[:8] = 200                                # a bright band at the top
    img[:, :4, 1] = 90                           # a green stripe at the left
    top = A.apply_host(img, _one([3, -8, 0, 0, 0, 0, 0, 0]))
    assert top[:, 8:].max() <= 1                                          # the bright band is gone entirely
    bottom = A.apply_host(img, _one([3, 0, 0, -8, 0, 0, 0, 0]))
    assert bottom[:9, 8:, 0].min() >= 199 and bottom[12:, 8:].max() <= 1  # still there, stretched from 8 to ~10.7 rows
    left = A.apply_host(img, _one([3, 0, 0, 0, -4, 0, 0, 0]))
    assert left[16:, :, 1].max() <= 1                                     # the stripe is gone
    right = A.apply_host(img, _one([3, 0, -4, 0, 0, 0, 0, 0]))
    assert right[16:, :4, 1].min() >= 89
    # a crop never removes the whole frame (sample_params keeps at least one row / column)
    import random
    for _ in range(200):
        rows = A.sample_params(random.Random(_), 3, 3, 1.0)
        for row in rows:
            if row[0] == 3:
                assert 3 + row[1] + row[3] >= 1 and 3 + row[2] + row[4] >= 1


def test_flip_is_an_involution_and_operations_compose_in_order():
    img = np.random.RandomState(3).randint(0, 256, (9, 11, 3)).astype(np.uint8)
    flip = [1, 0, 0, 0, 0, 0, 0, 0]
    np.testing.assert_array_equal(A.apply_host(img, [flip, flip, NONE[0]]), img)
    # random_order: pad-left-then-flip differs from flip-then-pad-left (the band ends up on the other side)
    pad_left = [3, 0, 0, 0, 3, 0, 0, 0]
    a = A.apply_host(img, [pad_left, flip, NONE[0]])
    b = A.apply_host(img, [flip, pad_left, NONE[0]])
    assert (a[:, -1] == 0).all() and (b[:, 0] == 0).all() and not np.array_equal(a, b)
