"""Row f4: the HIP patch-pipeline kernels (csrc/patches.hip, training/patch_pipeline.py) against the CPU restatement
(oracle/ref_patches.py) -- BIT-EXACT, uint8.  The restatement follows OpenCV's published fixed-point algorithm; parity with
cv2 itself is UNPINNED (no OpenCV here, no fixture in the reference): see tests/test_patches_cpu.py for what pins the restatement."""
import numpy as np
import pytest
import torch

from oracle import ref_patches as RP

pytestmark = pytest.mark.gpu


def _img(seed, h=256, w=256, c=3):
    return np.random.default_rng(seed).integers(0, 256, [h, w, c], dtype=np.uint8)


def _joints(seed, drop=()):
    rng = np.random.default_rng(seed)
    j = np.zeros([18, 3])
    j[:, 0] = rng.uniform(20, 170, 18)
    j[:, 1] = rng.uniform(10, 245, 18)
    j[:, 2] = rng.uniform(0.3, 1.0, 18)
    for name in drop:
        j[RP.ORDER.index(name), 2] = 0.0
    return j


@pytest.mark.parametrize('border', ['constant', 'replicate'])
def test_warp_perspective_is_bit_exact(border):
    from training import patch_pipeline as PP
    rng = np.random.default_rng(7)
    imgs = np.stack([_img(s) for s in range(3)])
    mats, idx = [], []
    for k in range(12):
        quad = np.float32(rng.uniform(-40, 300, [4, 2]))                  # quadrilaterals reaching outside the image
        mats.append(RP.get_perspective_transform(quad, np.float32([[0, 0], [0, 64], [64, 64], [64, 0]])))
        idx.append(k % 3)
    mats += [np.eye(3), np.array([[1, 0, 0.5], [0, 1, -7.25], [0, 0, 1.0]]), np.zeros([3, 3])]      # identity, sub-pixel shift, singular
    idx += [0, 1, 2]
    out = PP.warp_perspective(torch.from_numpy(imgs).cuda(), np.stack(mats), (64, 64), border, idx).cpu().numpy()
    code = RP.BORDER_REPLICATE if border == 'replicate' else RP.BORDER_CONSTANT
    for k, (m, i) in enumerate(zip(mats, idx)):
        assert np.array_equal(out[k], RP.warp_perspective(imgs[i], m, (64, 64), code)), k


def test_warp_back_to_the_full_image_and_odd_sizes():
    from training import patch_pipeline as PP
    patch = _img(11, 64, 64)
    quad = np.float32([[60, 40], [50, 210], [190, 230], [205, 60]])
    m_inv = RP.get_perspective_transform(np.float32([[0, 0], [0, 64], [64, 64], [64, 0]]), quad)
    out = PP.warp_perspective(torch.from_numpy(patch[None]).cuda(), m_inv[None], (256, 256), 'constant').cpu().numpy()[0]
    assert np.array_equal(out, RP.warp_perspective(patch, m_inv, (256, 256), RP.BORDER_CONSTANT))
    odd = _img(12, 97, 131, 1)                                               # one channel, sizes that are no multiple of the 64-column block
    m = np.array([[0.9, 0.1, 3.3], [-0.05, 1.1, -2.0], [1e-4, -2e-4, 1.0]])
    out = PP.warp_perspective(torch.from_numpy(odd[None]).cuda(), m[None], (77, 150), 'replicate').cpu().numpy()[0]
    assert np.array_equal(out, RP.warp_perspective(odd, m, (150, 77), RP.BORDER_REPLICATE))


@pytest.mark.parametrize('drops', [((), ('cnose',), ('lknee', 'rwrist')), (('rhip', 'rknee'), ('lshoulder',), ())])
def test_normalize_batch_equals_the_per_sample_restatement(drops):
    """dataset.py:838-927 for a batch of three samples (with missing key points: every fallback and the all-zero parts)."""
    from training import patch_pipeline as PP
    n = len(drops)
    up = np.stack([_img(20 + i) for i in range(n)])
    low = np.stack([_img(30 + i) for i in range(n)])
    rng = np.random.default_rng(3)

    def blob_mask(seed):                    # garment-like masks: 255 inside a few discs, 0 outside (the == 255 test needs exact values)
        r = np.random.default_rng(seed)
        ys, xs = np.mgrid[0:256, 0:256]
        m = np.zeros([256, 256], bool)
        for _ in range(6):
            cy, cx, rad = r.uniform(30, 226), r.uniform(30, 226), r.uniform(20, 70)
            m |= (ys - cy) ** 2 + (xs - cx) ** 2 < rad ** 2
        return (m[..., None] * np.uint8(255)).repeat(3, 2)
    um = np.stack([blob_mask(40 + i) for i in range(n)])
    lm = np.stack([blob_mask(50 + i) for i in range(n)])
    joints = np.stack([_joints(60 + i, d) for i, d in enumerate(drops)])
    cu = lambda a: torch.from_numpy(a).cuda()
    got = PP.normalize_batch(cu(up), cu(low), cu(um), cu(lm), joints)
    for i in range(n):
        ref = RP.normalize(up[i], low[i], um[i], lm[i], joints[i])
        for k in (0, 1, 2, 3, 6, 7):
            assert np.array_equal(got[k][i].cpu().numpy(), ref[k]), (i, k)
        assert np.array_equal(got[4][i].numpy(), ref[4])
        for h in range(4):
            assert np.array_equal(got[5][i, h].cpu().numpy(), ref[5][h]), (i, 'hand mask', h)
    assert got[2].any() and got[5].any()


def test_throughput_of_a_training_batch():
    """A batch of 16 samples (448 warps in the reference's formulation) is a handful of launches: sanity bound on the time."""
    import time
    from training import patch_pipeline as PP
    n = 16
    cu = lambda a: torch.from_numpy(a).cuda()
    up, low = cu(np.stack([_img(i) for i in range(n)])), cu(np.stack([_img(100 + i) for i in range(n)]))
    full = torch.full([n, 256, 256, 3], 255, dtype=torch.uint8, device='cuda')
    joints = np.stack([_joints(i) for i in range(n)])
    PP.normalize_batch(up, low, full, full, joints)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        PP.normalize_batch(up, low, full, full, joints)
    torch.cuda.synchronize()
    per_batch = (time.perf_counter() - t0) / 5
    print(f'normalize_batch(16): {1000 * per_batch:.2f} ms')
    assert per_batch < 0.25
