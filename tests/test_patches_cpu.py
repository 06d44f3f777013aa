"""Row f4 (body-part patch pipeline, reference training/dataset.py:751-927): the CPU restatement (oracle/ref_patches.py) against
properties any implementation of cv2.warpPerspective's bilinear fixed point must have, and the product's host-side geometry
(training/patch_pipeline.py) against the restatement.  cv2 is not installed here and the reference ships no fixture:
parity with OpenCV itself is UNPINNED -- these tests pin the two in-tree implementations to each other and to exact cases."""
import numpy as np
import pytest

from oracle import ref_patches as RP


def _img(seed=0, h=256, w=256):
    return np.random.default_rng(seed).integers(0, 256, [h, w, 3], dtype=np.uint8)


def test_identity_and_integer_translation_are_exact():
    img = _img()
    assert np.array_equal(RP.warp_perspective(img, np.eye(3), (256, 256)), img)
    shift = np.array([[1, 0, 5], [0, 1, -3], [0, 0, 1.0]])
    ref = np.zeros_like(img)
    ref[0:253, 5:256] = img[3:256, 0:251]
    assert np.array_equal(RP.warp_perspective(img, shift, (256, 256), RP.BORDER_CONSTANT), ref)
    rep = RP.warp_perspective(img, shift, (256, 256), RP.BORDER_REPLICATE)
    assert np.array_equal(rep[0:253, 5:256], ref[0:253, 5:256]) and np.array_equal(rep[10, 0], img[13, 0])


def test_half_pixel_shift_is_the_rounded_mean():
    img = _img(1)
    half = np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1.0]])              # dst x = src x + 0.5: every pixel samples x - 0.5
    out = RP.warp_perspective(img, half, (256, 256), RP.BORDER_REPLICATE)
    mean = (img[:, :-1].astype(int) + img[:, 1:].astype(int) + 1) >> 1      # (a * 2^14 + b * 2^14 + 2^14) >> 15
    assert np.array_equal(out[:, 1:], mean.astype(np.uint8))


def test_perspective_transform_maps_the_quadrilateral():
    src = np.float32([[40, 30], [35, 200], [180, 220], [200, 50]])
    dst = np.float32([[0, 0], [0, 64], [64, 64], [64, 0]])
    m = RP.get_perspective_transform(src, dst)
    for s, d in zip(src, dst):
        q = m @ np.array([s[0], s[1], 1.0])
        assert np.allclose(q[:2] / q[2], d, atol=1e-9)
    assert np.allclose(RP.invert3x3(m) @ m / (RP.invert3x3(m) @ m)[2, 2], np.eye(3), atol=1e-9)


def test_fixed_point_stays_within_the_quantisation_of_a_float_bilinear():
    """Smooth image: 1/32-pixel coordinates and 15-bit weights stay within two grey levels of a float64 bilinear warp."""
    ys, xs = np.mgrid[0:256, 0:256]
    img = np.stack([(xs + ys) // 2, (xs * 3 // 4 + 30), 255 - ys], -1).astype(np.uint8)
    src = np.float32([[40, 30], [35, 200], [180, 220], [200, 50]])
    m = RP.get_perspective_transform(src, np.float32([[0, 0], [0, 64], [64, 64], [64, 0]]))
    out = RP.warp_perspective(img, m, (64, 64), RP.BORDER_REPLICATE).astype(float)
    inv = np.linalg.inv(m)
    gy, gx = np.mgrid[0:64, 0:64]
    p = inv @ np.stack([gx.ravel(), gy.ravel(), np.ones(64 * 64)])
    fx, fy = (p[0] / p[2]).reshape(64, 64), (p[1] / p[2]).reshape(64, 64)
    x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    ax, ay = (fx - x0)[..., None], (fy - y0)[..., None]
    g = lambda y, x: img[np.clip(y, 0, 255), np.clip(x, 0, 255)].astype(float)
    ref = g(y0, x0) * (1 - ay) * (1 - ax) + g(y0, x0 + 1) * (1 - ay) * ax + g(y0 + 1, x0) * ay * (1 - ax) + g(y0 + 1, x0 + 1) * ay * ax
    assert np.abs(out - ref).max() <= 2.0


def _joints(seed, drop=()):
    rng = np.random.default_rng(seed)
    j = np.zeros([18, 3])
    j[:, 0] = rng.uniform(30, 160, 18)
    j[:, 1] = rng.uniform(15, 240, 18)
    j[:, 2] = rng.uniform(0.3, 1.0, 18)
    for name in drop:
        j[RP.ORDER.index(name), 2] = 0.0
    return j


@pytest.mark.parametrize('drop', [(), ('cnose',), ('lknee',), ('rknee', 'rhip'), ('lelbow', 'rwrist'), ('lshoulder',)])
def test_host_geometry_equals_the_restatement(drop):
    """training/patch_pipeline.py (the product's host side: quadrilaterals, 8 x 8 solves, adjugate inverse) against oracle/ref_patches.py,
    with every fallback of get_crop (missing nose, knee, hip, arm joints)."""
    from training import patch_pipeline as PP
    j = _joints(len(drop) + 3, drop)
    fwd, back, valid = PP.part_matrices(j[None], 256, 256)
    ref = RP.part_transforms(j, 256, 256)
    for k, (m, m_inv) in enumerate(ref):
        assert bool(valid[0, k]) == (m is not None)
        if m is not None:
            assert np.array_equal(fwd[0, k], m) and np.array_equal(back[0, k], m_inv)
            assert np.array_equal(PP.adjugate_inverse(m), RP.invert3x3(m))
        quad = PP.part_quadrilateral(j, PP.BODY_PARTS[k], 256)
        rq = RP.part_quadrilateral(j, RP.PARTS[k], 256)
        assert (quad is None) == (rq is None) and (quad is None or np.array_equal(quad, rq))


def test_normalize_shapes_and_compositing_order():
    img, low = _img(2), _img(3)
    mask = (np.random.default_rng(4).integers(0, 2, [256, 256, 1], dtype=np.uint8) * 255).repeat(3, 2)
    full = np.full([256, 256, 3], 255, np.uint8)
    out = RP.normalize(img, low, full, mask, _joints(5))
    assert [o.shape if hasattr(o, 'shape') else len(o) for o in out] == [(64, 64, 30), (64, 64, 12), (256, 256, 3), (256, 256, 3), (10, 3, 3), 4,
                                                                         (64, 64, 30), (64, 64, 12)]
    # with an all-255 mask a pixel inside a part's quadrilateral shows that part's round trip; outside every part it stays black
    covered = np.zeros([256, 256], bool)
    for hm in out[5]:
        covered |= hm[..., 0].astype(bool)
    assert covered.any() and (out[2][~(out[2].any(-1))] == 0).all()
