"""Body-part patch pipeline of the data loader on the GPU (reference training/dataset.py:751-836 ``get_crop`` and :838-927
``normalize``), batched: the reference warps ten body-part quadrilaterals of every sample into (W/4) x (H/4) patches, warps
them back and composites them where the warped garment mask is 255 -- about 28 ``cv2.warpPerspective`` calls per sample on the
host (the reason the reference's loader, not its networks, bounds real-data throughput).  Here the ten (or
fourteen) forward warps of a whole batch are ONE launch of ``pasta_warp_perspective_u8`` per source tensor and the warp-back
+ mask test + compositing of all parts one launch of ``pasta_patch_composite_u8`` per garment.

The projective matrices are 8 x 8 solves on eighteen key points -- host work, numpy float64, as in the reference.  Image
tensors stay uint8 HWC as the reference's arrays.  Numerics: OpenCV's fixed-point bilinear interpolation restated from its
published algorithm (csrc/patches.hip); OpenCV is not installed here, so parity with cv2 is UNPINNED;
the kernels are held bit for bit to oracle/ref_patches.py in tests/test_patches_gpu.py."""
import ctypes

import numpy as np
import torch

from torch_utils.ops import _native

# dataset.py:846-861
BODY_PARTS = (('lshoulder', 'lhip', 'rhip', 'rshoulder'), ('lshoulder', 'rshoulder', 'cnose'), ('lshoulder', 'lelbow'), ('lelbow', 'lwrist'),
              ('rshoulder', 'relbow'), ('relbow', 'rwrist'), ('lhip', 'lknee'), ('lknee', 'lankle'), ('rhip', 'rknee'), ('rknee', 'rankle'))
JOINT_ORDER = ('cnose', 'cneck', 'rshoulder', 'relbow', 'rwrist', 'lshoulder', 'lelbow', 'lwrist', 'rhip', 'rknee', 'rankle', 'lhip', 'lknee',
               'lankle', 'reye', 'leye', 'rear', 'lear')
_JOINT = {name: i for i, name in enumerate(JOINT_ORDER)}
LOWER_FROM = 6              # parts 6..9 (the legs) are also cut from the lower garment (dataset.py:890)
ARM_PARTS = (2, 3, 4, 5)    # whose warped-back masks the reference returns as denorm_hand_masks (:906-910)


def _seen(joints, names):
    return all(joints[_JOINT[n], 2] >= 0.1 for n in names)              # dataset.py:748-749


def _box_around(p, q, half_width_ratio):
    """Rectangle around the segment p -> q, ``half_width_ratio`` of its length to either side (corner order of dataset.py:821-829)."""
    seg = q - p
    normal = np.array([-seg[1], seg[0]], dtype=seg.dtype) * half_width_ratio
    return np.float32([p + normal, p - normal, q - normal, q + normal])


def part_quadrilateral(joints, part, image_height, aspect=0.5, x_pad=32):
    """Source quadrilateral [4, 2] float32 of one body part, or None when its key points are missing (dataset.py:751-829).
    ``joints`` [18, 3] = (x, y, confidence) in the unpadded 192-wide image; ``x_pad`` shifts into the padded square."""
    names = list(part)
    if not _seen(joints, names):
        if names[0] in ('lhip', 'rhip') and names[1] in ('lknee', 'rknee') and names[0][0] == names[1][0]:
            names = names[:1]                                   # thigh without its knee: straight down from the hip
        elif names == ['lshoulder', 'rshoulder', 'cnose']:
            names = ['lshoulder', 'rshoulder', 'rshoulder']     # head without the nose: a square above the shoulders
        if not _seen(joints, names):
            return None
    pts = np.float32([[joints[_JOINT[n], 0], joints[_JOINT[n], 1]] for n in names])
    pts[:, 0] = pts[:, 0] + x_pad                       # in float32, after the conversion (dataset.py:780)
    if len(pts) == 4:
        return pts
    if len(pts) == 1:
        return _box_around(pts[0], np.float32([pts[0][0], image_height - 1]), aspect / 2.0)
    if len(pts) == 2:
        return _box_around(pts[0], pts[1], aspect / 2.0)
    if names[2] == 'rshoulder':
        seg = pts[1] - pts[0]
        normal = np.array([-seg[1], seg[0]])
        if normal[1] > 0.0:
            normal = -normal
        return np.float32([pts[0] + normal, pts[0], pts[1], pts[1] + normal])
    neck = 0.5 * (pts[0] + pts[1])
    top = np.float32(neck + 2 * (pts[2] - neck))
    a, b, c, d = _box_around(top, np.float32(neck), 0.5)
    return np.float32([b, c, d, a])


def perspective_matrix(src, dst):
    """3 x 3 float64 map taking the four ``src`` points onto the four ``dst`` points (cv2.getPerspectiveTransform's system)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    rows, rhs = [], []
    for (x, y), (u, v) in zip(src, dst):
        rows.append([x, y, 1, 0, 0, 0, -x * u, -y * u])
        rhs.append(u)
    for (x, y), (u, v) in zip(src, dst):
        rows.append([0, 0, 0, x, y, 1, -x * v, -y * v])
        rhs.append(v)
    try:
        h = np.linalg.solve(np.array(rows), np.array(rhs))
    except np.linalg.LinAlgError:
        h = np.zeros(8)
    return np.append(h, 1.0).reshape(3, 3)


def adjugate_inverse(m):
    """Inverse of a 3 x 3 float64 matrix by cofactors (what cv2.warpPerspective applies to its argument); zeros if singular."""
    m = np.asarray(m, np.float64)
    c = np.empty([3, 3])
    c[0, 0] = m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]; c[0, 1] = m[0, 2] * m[2, 1] - m[0, 1] * m[2, 2]; c[0, 2] = m[0, 1] * m[1, 2] - m[0, 2] * m[1, 1]
    c[1, 0] = m[1, 2] * m[2, 0] - m[1, 0] * m[2, 2]; c[1, 1] = m[0, 0] * m[2, 2] - m[0, 2] * m[2, 0]; c[1, 2] = m[0, 2] * m[1, 0] - m[0, 0] * m[1, 2]
    c[2, 0] = m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]; c[2, 1] = m[0, 1] * m[2, 0] - m[0, 0] * m[2, 1]; c[2, 2] = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    det = (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0]) +
           m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))
    return c * (1.0 / det) if det != 0 else np.zeros([3, 3])


def part_matrices(joints, width, height, box_factor=2):
    """For a batch of key points [N, 18, 3]: (M [N,10,3,3], M_inv [N,10,3,3], valid [N,10]) float64 / bool -- image -> patch and
    patch -> image maps of the ten parts (dataset.py:831-836); zeros where a part is missing."""
    joints = np.asarray(joints, np.float64)
    n = joints.shape[0]
    pw, ph = width // 2 ** box_factor, height // 2 ** box_factor
    corners = np.float32([[0, 0], [0, ph], [pw, ph], [pw, 0]])
    fwd, back, valid = np.zeros([n, 10, 3, 3]), np.zeros([n, 10, 3, 3]), np.zeros([n, 10], bool)
    for i in range(n):
        for k, part in enumerate(BODY_PARTS):
            quad = part_quadrilateral(joints[i], part, height)
            if quad is not None:
                fwd[i, k], back[i, k], valid[i, k] = perspective_matrix(quad, corners), perspective_matrix(corners, quad), True
    return fwd, back, valid


def _u8(t):
    assert t.dtype == torch.uint8 and t.is_cuda and t.ndim == 4 and t.shape[-1] == 3, 'uint8 [N, H, W, 3] tensors on the GPU'
    return t.contiguous()


def warp_perspective(src, matrices, out_hw, border='constant', src_index=None, valid=None):
    """Batched ``cv2.warpPerspective(src[i], M, (w, h), borderMode=...)`` for uint8 [*, H, W, C] GPU tensors: ``matrices``
    [B, 3, 3] float64 (host), one output per matrix; ``src_index`` [B] picks the source of each (default: i)."""
    _native.require_gpu(src, 'warp_perspective')
    b = int(len(matrices))
    inv = np.ascontiguousarray(np.stack([adjugate_inverse(m) for m in matrices]).reshape(b, 9))
    dev = src.device
    inv_t = torch.from_numpy(inv).to(dev)
    idx_t = torch.as_tensor(np.asarray(src_index, np.int32), device=dev) if src_index is not None else None
    val_t = torch.as_tensor(np.asarray(valid, np.uint8), device=dev) if valid is not None else None
    oh, ow = out_hw
    dst = torch.empty([b, oh, ow, src.shape[-1]], dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _native.check(_native.lib().pasta_warp_perspective_u8(_native.ptr(src), _native.ptr(idx_t), _native.ptr(inv_t), _native.ptr(val_t),
                                                              _native.ptr(dst), b, int(src.shape[1]), int(src.shape[2]), oh, ow,
                                                              int(src.shape[-1]), 1 if border == 'replicate' else 0, _native.stream()))
    return dst


def normalize_batch(upper_img, lower_img, upper_mask, lower_mask, joints, box_factor=2):
    """``normalize`` (dataset.py:838-927) for a batch on the GPU.  Images and 3-channel masks: uint8 [N, H, W, 3] CUDA tensors;
    ``joints`` [N, 18, 3] (host).  Returns the reference's tuple, batched:
    (norm_img [N,h,w,30], norm_img_lower [N,h,w,12], denorm_upper [N,H,W,3], denorm_lower [N,H,W,3], M_invs [N,10,3,3] float32,
     denorm_hand_masks [N,4,H,W,1], clothes_masks [N,h,w,30], clothes_masks_lower [N,h,w,12])."""
    upper_img, lower_img, upper_mask, lower_mask = _u8(upper_img), _u8(lower_img), _u8(upper_mask), _u8(lower_mask)
    n, height, width, _ = upper_img.shape
    ph, pw = height // 2 ** box_factor, width // 2 ** box_factor
    fwd, back, valid = part_matrices(joints, width, height, box_factor)
    dev = upper_img.device
    sample = np.repeat(np.arange(n, dtype=np.int32), 10)
    flat_valid = valid.reshape(-1)
    # image -> patch (BORDER_REPLICATE): all ten parts of every sample in one launch per source tensor
    warp = lambda src, sel: warp_perspective(src, fwd.reshape(-1, 3, 3)[sel], (ph, pw), 'replicate', sample[sel], flat_valid[sel])
    everything = np.arange(n * 10)
    legs = everything.reshape(n, 10)[:, LOWER_FROM:].reshape(-1)
    p_img = warp(upper_img, everything).reshape(n, 10, ph, pw, 3)
    p_mask = warp(upper_mask, everything).reshape(n, 10, ph, pw, 3)
    p_img_l = warp(lower_img, legs).reshape(n, 10 - LOWER_FROM, ph, pw, 3)
    p_mask_l = warp(lower_mask, legs).reshape(n, 10 - LOWER_FROM, ph, pw, 3)

    # patch -> image (BORDER_CONSTANT) + mask test + compositing, all parts in order, one launch per garment
    def composite(patches, masks, part_ids, want_part_masks):
        p = len(part_ids)
        inv = np.ascontiguousarray(np.stack([adjugate_inverse(back[i, k]) for i in range(n) for k in part_ids]).reshape(n * p, 9))
        inv_t = torch.from_numpy(inv).to(dev)
        val_t = torch.as_tensor(np.ascontiguousarray(valid[:, part_ids]).astype(np.uint8), device=dev)
        out = torch.empty([n, height, width, 3], dtype=torch.uint8, device=dev)
        pm = torch.empty([n, p, height, width], dtype=torch.uint8, device=dev) if want_part_masks else None
        with torch.cuda.device(dev):
            _native.check(_native.lib().pasta_patch_composite_u8(_native.ptr(patches.contiguous()), _native.ptr(masks.contiguous()), _native.ptr(inv_t),
                                                                 _native.ptr(val_t), _native.ptr(out), _native.ptr(pm), n, p, ph, pw, height, width,
                                                                 _native.stream()))
        return out, pm
    den_u, part_masks = composite(p_img, p_mask, list(range(10)), True)
    den_l, _ = composite(p_img_l, p_mask_l, list(range(LOWER_FROM, 10)), False)
    hwc = lambda t: t.permute(0, 2, 3, 1, 4).reshape(n, ph, pw, -1)            # parts concatenated along the channel axis (:918-921)
    m_invs = torch.from_numpy(np.where(valid[..., None, None], back, 0.0).astype(np.float32))
    hand_masks = part_masks[:, list(ARM_PARTS)].unsqueeze(-1)
    return hwc(p_img), hwc(p_img_l), den_u, den_l, m_invs, hand_masks, hwc(p_mask), hwc(p_mask_l)
