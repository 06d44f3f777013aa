"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline): CPU restatement of the reference's
body-part patch pipeline -- training/dataset.py:751-836 (``get_crop``: a quadrilateral per body part from the pose keypoints
and its perspective transforms) and :838-927 (``normalize``: ten parts warped into w/4 x h/4 patches, warped back and
composited where the warped garment mask is 255).

The reference delegates the numerics to OpenCV (``cv2.getPerspectiveTransform``, ``cv2.warpPerspective`` with the default
bilinear interpolation on uint8 images).  OpenCV is NOT installed in this container and the reference ships no fixture of
this pipeline, so this file restates OpenCV 4.x's published algorithm (modules/imgproc/src/imgwarp.cpp) from its
description: the 8 x 8 linear system of getPerspectiveTransform; warpPerspective = invert M in double (closed-form 3 x 3
adjugate), source coordinates in 1/32-pixel fixed point (``cvRound`` = round half to even of ``(X0 + M0 * x1) * 32 / W``
evaluated per 64-column block as WarpPerspectiveInvoker does), bilinear weights ``(32 - a)(32 - b) * 32`` summing to 2^15,
``(sum + 2^14) >> 15``, BORDER_REPLICATE by clamping the tap coordinates, BORDER_CONSTANT by substituting 0 for taps
outside.  PARITY UNPINNED: nothing the reference produced can be compared here; the HIP kernels are held to THIS restatement
bit for bit (tests/test_patches_gpu.py), and DESIGN.md says so."""
import numpy as np

INTER_BITS = 5
TAB = 1 << INTER_BITS            # 32 sub-pixel positions
COEF_BITS = 15
BLOCK_W, BLOCK_H = 64, 16        # WarpPerspectiveInvoker's tile for images of at least 64 x 16

BORDER_CONSTANT, BORDER_REPLICATE = 0, 1

PARTS = (("lshoulder", "lhip", "rhip", "rshoulder"), ("lshoulder", "rshoulder", "cnose"), ("lshoulder", "lelbow"), ("lelbow", "lwrist"),
         ("rshoulder", "relbow"), ("relbow", "rwrist"), ("lhip", "lknee"), ("lknee", "lankle"), ("rhip", "rknee"), ("rknee", "rankle"))
ORDER = ('cnose', 'cneck', 'rshoulder', 'relbow', 'rwrist', 'lshoulder', 'lelbow', 'lwrist', 'rhip', 'rknee', 'rankle', 'lhip', 'lknee',
         'lankle', 'reye', 'leye', 'rear', 'lear')


def get_perspective_transform(src, dst):
    """The 3 x 3 map with M @ (x, y, 1) ~ (u, v, 1) for four point pairs (imgwarp.cpp, getPerspectiveTransform): float64."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    a = np.zeros([8, 8])
    b = np.zeros([8])
    for i in range(4):
        x, y = src[i]
        u, v = dst[i]
        a[i, 0:3] = (x, y, 1)
        a[i, 6:8] = (-x * u, -y * u)
        a[i + 4, 3:6] = (x, y, 1)
        a[i + 4, 6:8] = (-x * v, -y * v)
        b[i], b[i + 4] = u, v
    try:
        sol = np.linalg.solve(a, b)
    except np.linalg.LinAlgError:            # degenerate quadrilateral: OpenCV's solver returns zeros
        sol = np.zeros([8])
    return np.append(sol, 1.0).reshape(3, 3)


def invert3x3(m):
    """cv::invert on a 3 x 3 double matrix: adjugate / determinant; all zeros when singular."""
    m = np.asarray(m, np.float64)
    det = (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0]) +
           m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))
    if det == 0:
        return np.zeros([3, 3])
    d = 1.0 / det
    t = np.empty([3, 3])
    t[0, 0] = (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) * d
    t[0, 1] = (m[0, 2] * m[2, 1] - m[0, 1] * m[2, 2]) * d
    t[0, 2] = (m[0, 1] * m[1, 2] - m[0, 2] * m[1, 1]) * d
    t[1, 0] = (m[1, 2] * m[2, 0] - m[1, 0] * m[2, 2]) * d
    t[1, 1] = (m[0, 0] * m[2, 2] - m[0, 2] * m[2, 0]) * d
    t[1, 2] = (m[0, 2] * m[1, 0] - m[0, 0] * m[1, 2]) * d
    t[2, 0] = (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]) * d
    t[2, 1] = (m[0, 1] * m[2, 0] - m[0, 0] * m[2, 1]) * d
    t[2, 2] = (m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]) * d
    return t


def source_coordinates(minv, width, height):
    """Fixed-point source coordinates (X, Y in 1/32 pixel) of every destination pixel, given the inverted matrix."""
    minv = np.asarray(minv, np.float64)
    xs = np.arange(width)
    x_block = (xs // BLOCK_W) * BLOCK_W                    # the block's first column: the products are formed block-relative
    x1 = (xs - x_block).astype(np.float64)
    ys = np.arange(height, dtype=np.float64)[:, None]
    xb = x_block.astype(np.float64)[None, :]
    x0 = minv[0, 0] * xb + minv[0, 1] * ys + minv[0, 2]
    y0 = minv[1, 0] * xb + minv[1, 1] * ys + minv[1, 2]
    w0 = minv[2, 0] * xb + minv[2, 1] * ys + minv[2, 2]
    w = w0 + minv[2, 0] * x1[None, :]
    with np.errstate(divide='ignore', invalid='ignore'):
        w = np.where(w != 0, TAB / w, 0.0)
    fx = np.clip((x0 + minv[0, 0] * x1[None, :]) * w, -2147483648.0, 2147483647.0)
    fy = np.clip((y0 + minv[1, 0] * x1[None, :]) * w, -2147483648.0, 2147483647.0)
    return np.rint(fx).astype(np.int64), np.rint(fy).astype(np.int64)      # rint: round half to even, as cvRound


def warp_perspective(src, m, dsize, border=BORDER_CONSTANT):
    """cv2.warpPerspective(src, m, dsize, flags=INTER_LINEAR, borderMode=border) for uint8 [H, W, C] images (border value 0)."""
    src = np.asarray(src)
    assert src.dtype == np.uint8 and src.ndim == 3
    sh, sw, _ = src.shape
    dw, dh = dsize
    X, Y = source_coordinates(invert3x3(m), dw, dh)
    # remap works on short coordinates: saturate
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    ax = (X & (TAB - 1)).astype(np.int64)
    ay = (Y & (TAB - 1)).astype(np.int64)
    w00 = (TAB - ay) * (TAB - ax) * 32
    w01 = (TAB - ay) * ax * 32
    w10 = ay * (TAB - ax) * 32
    w11 = ay * ax * 32
    s = src.astype(np.int64)

    def tap(yy, xx):
        if border == BORDER_REPLICATE:
            return s[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)]
        inside = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        v = s[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)]
        return np.where(inside[..., None], v, 0)
    acc = (tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None] + tap(sy + 1, sx) * w10[..., None] +
           tap(sy + 1, sx + 1) * w11[..., None])
    return np.clip((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255).astype(np.uint8)


def _valid(conf):
    return bool(np.all(np.asarray(conf) >= 0.1))          # dataset.py:748-749


def part_quadrilateral(joints, bpart, o_h, ar=0.5, x_pad=32):
    """dataset.py:751-829: the source quadrilateral (4 x 2 float32) of one body part, or None when its joints are missing.
    ``joints`` [18, 3] = (x, y, confidence) in the unpadded image; ``x_pad`` = the white border added left of it."""
    joints = np.asarray(joints)
    names = list(bpart)
    idx = [ORDER.index(b) for b in names]
    if not _valid(joints[idx][:, 2]):
        if names[:2] == ["lhip", "lknee"]:
            names = ["lhip"]
        elif names[:2] == ["rhip", "rknee"]:
            names = ["rhip"]
        elif names == ["lshoulder", "rshoulder", "cnose"]:
            names = ["lshoulder", "rshoulder", "rshoulder"]
        idx = [ORDER.index(b) for b in names]
        if not _valid(joints[idx][:, 2]):
            return None
    pts = np.float32(joints[idx][:, :2])
    pts[:, 0] = pts[:, 0] + x_pad
    if len(pts) == 1:                                  # hip only: a segment straight down to the image's last row
        pts = np.float32([pts[0], np.float32([pts[0][0], o_h - 1])])
    if len(pts) == 4:
        return pts
    if len(pts) == 3:
        if names == ["lshoulder", "rshoulder", "rshoulder"]:        # no nose: a square above the shoulder line
            seg = pts[1] - pts[0]
            normal = np.array([-seg[1], seg[0]])
            if normal[1] > 0.0:
                normal = -normal
            return np.float32([pts[0] + normal, pts[0], pts[1], pts[1] + normal])
        neck = 0.5 * (pts[0] + pts[1])
        ends = np.float32([neck + 2 * (pts[2] - neck), neck])
        seg = ends[1] - ends[0]
        normal = np.array([-seg[1], seg[0]])
        a, b = ends[0] + 0.5 * normal, ends[0] - 0.5 * normal
        c, d = ends[1] - 0.5 * normal, ends[1] + 0.5 * normal
        return np.float32([b, c, d, a])
    seg = pts[1] - pts[0]
    normal = np.array([-seg[1], seg[0]])
    alpha = ar / 2.0
    return np.float32([pts[0] + alpha * normal, pts[0] - alpha * normal, pts[1] - alpha * normal, pts[1] + alpha * normal])


def part_transforms(joints, o_w, o_h, box_factor=2):
    """For the ten parts: (M, M_inv) float64 3 x 3 or (None, None) (dataset.py:831-836: patch corners (0,0), (0,h), (w,h), (w,0))."""
    w, h = o_w // 2 ** box_factor, o_h // 2 ** box_factor
    dst = np.float32([[0.0, 0.0], [0.0, 1.0], [1.0, 1.0], [1.0, 0.0]]) * np.float32([[w, h]])
    out = []
    for bpart in PARTS:
        quad = part_quadrilateral(joints, bpart, o_h)
        out.append((None, None) if quad is None else (get_perspective_transform(quad, dst), get_perspective_transform(dst, quad)))
    return out


def normalize(upper_img, lower_img, upper_mask, lower_mask, joints, box_factor=2):
    """dataset.py:838-927 for one sample.  Images and 3-channel masks uint8 [H, W, 3].  Returns the reference's tuple:
    (img [h,w,30], img_lower [h,w,12], denorm_upper [H,W,3], denorm_lower [H,W,3], M_invs [10,3,3] float32,
     denorm_hand_masks (4 x [H,W,1]), clothes_masks [h,w,30], clothes_masks_lower [h,w,12])."""
    o_h, o_w = upper_img.shape[:2]
    h, w = o_h // 2 ** box_factor, o_w // 2 ** box_factor
    mats = part_transforms(joints, o_w, o_h, box_factor)
    imgs, imgs_lower, masks, masks_lower, m_invs, hand_masks = [], [], [], [], [], []
    den_u, den_l = np.zeros_like(upper_img), np.zeros_like(upper_img)
    for ii, (m, m_inv) in enumerate(mats):
        p_img = np.zeros([h, w, 3], np.uint8)
        p_img_l, p_mask, p_mask_l = p_img.copy(), p_img.copy(), p_img.copy()
        back_mask = None
        if m is not None:
            p_img = warp_perspective(upper_img, m, (w, h), BORDER_REPLICATE)
            p_mask = warp_perspective(upper_mask, m, (w, h), BORDER_REPLICATE)
            back = warp_perspective(p_img, m_inv, (o_w, o_h), BORDER_CONSTANT)
            back_mask = (warp_perspective(p_mask, m_inv, (o_w, o_h), BORDER_CONSTANT)[..., 0:1] == 255).astype(np.uint8)
            den_u = back * back_mask + den_u * (1 - back_mask)
            if ii >= 6:
                p_img_l = warp_perspective(lower_img, m, (w, h), BORDER_REPLICATE)
                p_mask_l = warp_perspective(lower_mask, m, (w, h), BORDER_REPLICATE)
                back_l = warp_perspective(p_img_l, m_inv, (o_w, o_h), BORDER_CONSTANT)
                bm_l = (warp_perspective(p_mask_l, m_inv, (o_w, o_h), BORDER_CONSTANT)[..., 0:1] == 255).astype(np.uint8)
                den_l = back_l * bm_l + den_l * (1 - bm_l)
            m_invs.append(np.float32(m_inv))
        else:
            m_invs.append(np.zeros([3, 3], np.float32))
        if 2 <= ii <= 5:
            hand_masks.append(back_mask if back_mask is not None else np.zeros([o_h, o_w, 1], np.uint8))
        imgs.append(p_img)
        masks.append(p_mask)
        if ii >= 6:
            imgs_lower.append(p_img_l)
            masks_lower.append(p_mask_l)
    return (np.concatenate(imgs, 2), np.concatenate(imgs_lower, 2), den_u, den_l, np.stack(m_invs), hand_masks,
            np.concatenate(masks, 2), np.concatenate(masks_lower, 2))
