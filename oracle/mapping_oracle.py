"""CPU ORACLE (test infrastructure, NOT product code) for the mapping build: ctypes binding of
``oracle/mapping_oracle.c`` plus a numpy restatement of the integer assembly around it
(MapImages post-processing, lexargunique, ImageMapping.from_dense, select_points).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Reference citations are relative to /root/reference/torch_points3d/.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle_mapping.so")

CAMERA_CODE = {"s3dis_equirectangular": 0, "scannet": 1, "kitti360_perspective": 2, "kitti360_fisheye": 3}


class Camera(ctypes.Structure):
    """Same layout as struct dvo_camera (mapping_oracle.c) == struct dva_camera (include/dva.h)."""
    _fields_ = [
        ("model", ctypes.c_int32), ("img_w", ctypes.c_int32), ("img_h", ctypes.c_int32),
        ("crop_top", ctypes.c_int32), ("crop_bottom", ctypes.c_int32),
        ("r_min", ctypes.c_float), ("r_max", ctypes.c_float),
        ("img_xyz", ctypes.c_float * 3), ("rot", ctypes.c_float * 9), ("trans", ctypes.c_float * 3),
        ("fx", ctypes.c_float), ("fy", ctypes.c_float), ("mx", ctypes.c_float), ("my", ctypes.c_float),
        ("fisheye", ctypes.c_float * 7),
        ("r_min_d", ctypes.c_double), ("r_max_d", ctypes.c_double),
        ("voxel", ctypes.c_double), ("k_swell", ctypes.c_double), ("d_swell", ctypes.c_double),
        ("exact", ctypes.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(LIB_PATH)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        cam = ctypes.POINTER(Camera)
        L.dvo_camera_projection.restype = i64
        L.dvo_camera_projection.argtypes = [vp, i64, cam, vp, vp, vp, vp, vp]
        L.dvo_splat.restype = None
        L.dvo_splat.argtypes = [vp, vp, vp, vp, i64, cam, vp]
        L.dvo_zbuffer.restype = i64
        L.dvo_zbuffer.argtypes = [vp, vp, vp, vp, i64, cam, vp, vp, vp]
        L.dvo_visibility.restype = i64
        L.dvo_visibility.argtypes = [vp, i64, cam, vp, vp, vp, vp, vp, vp, vp]
        L.dvo_mapping_features.restype = ctypes.c_int32
        L.dvo_mapping_features.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, cam, i64, vp]
        L.dvo_splat_area.restype = i64
        L.dvo_splat_area.argtypes = [vp, i64]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def rotation_from_opk(opk):
    """core/multimodal/visibility.py:57-90 (float32 cos/sin, float32 3x3 products)."""
    opk = np.asarray(opk, dtype=np.float32)
    co, so = np.cos(opk[0]), np.sin(opk[0])
    cp, sp = np.cos(opk[1]), np.sin(opk[1])
    ck, sk = np.cos(opk[2]), np.sin(opk[2])
    M_o = np.array([[1.0, 0.0, 0.0], [0.0, co, -so], [0.0, so, co]], dtype=np.float32)
    M_p = np.array([[cp, 0.0, sp], [0.0, 1.0, 0.0], [-sp, 0.0, cp]], dtype=np.float32)
    M_k = np.array([[ck, -sk, 0.0], [sk, ck, 0.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    return np.dot(M_o, np.dot(M_p, M_k))


def make_camera(camera, img_size, img_xyz, crop_top=0, crop_bottom=0, r_min=0.5, r_max=30.0, voxel=0.1,
                k_swell=1.0, d_swell=1000, exact=False, img_opk=None, img_extrinsic=None,
                img_intrinsic_pinhole=None, img_intrinsic_fisheye=None):
    """Host-side per-image scalar preparation, exactly as the reference does it before its loops."""
    c = Camera()
    c.model = CAMERA_CODE[camera]
    c.img_w, c.img_h = int(img_size[0]), int(img_size[1])
    c.crop_top, c.crop_bottom = int(crop_top), int(crop_bottom)
    c.r_min, c.r_max = float(r_min), float(r_max)
    c.r_min_d, c.r_max_d = float(r_min), float(r_max)
    c.voxel, c.k_swell, c.d_swell = float(voxel), float(k_swell), float(d_swell)
    c.exact = int(bool(exact))
    c.img_xyz[:] = np.asarray(img_xyz, dtype=np.float32).tolist()
    rot, trans = np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32)
    if camera == "s3dis_equirectangular":
        rot = rotation_from_opk(np.zeros(3) if img_opk is None else img_opk)
    elif camera == "scannet":
        M = np.linalg.inv(np.ascontiguousarray(np.asarray(img_extrinsic, dtype=np.float32)))  # :232
        rot, trans = M[:3, :3].copy(), M[:3, 3].copy()
    else:
        E = np.asarray(img_extrinsic, dtype=np.float32)
        rot, trans = E[:3, :3].copy(), E[:3, 3].copy()
    c.rot[:] = rot.astype(np.float32).reshape(-1).tolist()
    c.trans[:] = trans.astype(np.float32).tolist()
    if img_intrinsic_pinhole is not None:
        K = np.asarray(img_intrinsic_pinhole, dtype=np.float32)
        c.fx, c.fy, c.mx, c.my = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    if img_intrinsic_fisheye is not None:
        c.fisheye[:] = np.asarray(img_intrinsic_fisheye, dtype=np.float32).tolist()
    return c


def camera_projection(xyz, cam, mask=None):
    """visibility.py:478-538 -> (idx_1, dist, x_proj, y_proj)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    idx1 = np.empty(n, np.int64)
    dist = np.empty(n, np.float32)
    xp, yp = np.empty(n, np.float64), np.empty(n, np.float64)
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    m = lib().dvo_camera_projection(_p(xyz), n, ctypes.byref(cam), _p(mask), _p(idx1), _p(dist), _p(xp), _p(yp))
    return idx1[:m], dist[:m], xp[:m], yp[:m]


def splat(xp, yp, dist, xyz_sel, cam):
    m = xp.shape[0]
    out = np.empty((m, 4), np.int32)
    xyz_sel = np.ascontiguousarray(xyz_sel, dtype=np.float32)
    lib().dvo_splat(_p(xp), _p(yp), _p(dist), _p(xyz_sel), m, ctypes.byref(cam), _p(out))
    return out


def visibility(xyz, cam, mask=None):
    """VisibilityModel.__call__ (visibility.py:1699-1757) without the features."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    cap = max(n, cam.img_w * (cam.img_h - cam.crop_top - cam.crop_bottom)) if not cam.exact else n
    idx = np.empty(cap, np.int64)
    xpix, ypix = np.empty(cap, np.int64), np.empty(cap, np.int64)
    depth = np.empty(cap, np.float32)
    xp, yp = np.empty(cap, np.float64), np.empty(cap, np.float64)
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    q = lib().dvo_visibility(_p(xyz), n, ctypes.byref(cam), _p(mask), _p(idx), _p(xpix), _p(ypix), _p(depth),
                             _p(xp), _p(yp))
    return dict(idx=idx[:q], x=xpix[:q], y=ypix[:q], depth=depth[:q], x_proj=xp[:q], y_proj=yp[:q])


def mapping_features(xyz, vis, cam, linearity=None, planarity=None, scattering=None, normals=None):
    """visibility.py:1548-1582."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    q = vis["idx"].shape[0]
    arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float32)
            for a in (linearity, planarity, scattering, normals)]
    ncol = 2 + sum(a is not None for a in arrs)
    out = np.empty((q, ncol), np.float32)
    got = lib().dvo_mapping_features(_p(xyz), _p(vis["idx"]), _p(vis["depth"]), _p(vis["y_proj"]),
                                     *[_p(a) for a in arrs], ctypes.byref(cam), q, _p(out))
    assert got == ncol
    return out


# ----------------------------------------------------------------------------------------------
# integer assembly (numpy): utils/multimodal.py, core/multimodal/csr.py, core/multimodal/image.py
# ----------------------------------------------------------------------------------------------

def composite(*cols):
    """utils/multimodal.py:182-232 CompositeNDArray: key = sum_i a_i * prod_{j>i}(max_j + 1)."""
    cols = [np.asarray(c).astype(np.int64) for c in cols]
    if cols[0].shape[0] == 0:
        return np.zeros(0, np.int64)
    maxs = [int(np.abs(c).max()) + 1 for c in cols]
    assert np.prod([float(m) for m in maxs]) < np.iinfo(np.int64).max
    key = np.zeros_like(cols[0])
    for i, c in enumerate(cols):
        base = 1
        for m in maxs[i + 1:]:
            base *= m
        key = key + c * base
    return key


def lexargunique(*cols):
    """utils/multimodal.py:308-312: index of the first occurrence of every distinct key, key-sorted."""
    return np.unique(composite(*cols), return_index=True)[1]


def lexargsort_stable(*cols):
    """utils/multimodal.py:314-318 uses an unstable argsort; a stable one is one admissible order."""
    return np.argsort(composite(*cols), kind="stable")


def sorted_to_pointers(idx):
    """core/multimodal/csr.py:158-172."""
    return np.concatenate([[0], np.where(idx[1:] > idx[:-1])[0] + 1, [idx.shape[0]]]).astype(np.int64)


def from_dense(point_ids, image_ids, pixels, features, num_points):
    """core/multimodal/image.py:1728-1795 -> dict(pointers, images, atom_pointers, pixels, features)."""
    order = lexargsort_stable(point_ids, image_ids)
    point_ids, image_ids, pixels, features = point_ids[order], image_ids[order], pixels[order], features[order]
    comp = composite(point_ids, image_ids)
    atom_ptr = sorted_to_pointers(comp)
    last = atom_ptr[1:] - 1
    v_img, v_pt = image_ids[last], point_ids[last]
    sizes = np.diff(atom_ptr)
    v_feat = (np.add.reduceat(features.astype(np.float64), atom_ptr[:-1], axis=0) / sizes[:, None]).astype(np.float32)
    view_ptr = sorted_to_pointers(v_pt)
    groups = v_pt[view_ptr[1:] - 1]
    num_points = max(int(num_points), int(groups.max()) + 1)
    # insert_empty_groups (csr.py:197-229)
    starts = np.concatenate([[-1], groups])
    ends = np.concatenate([groups, [num_points]])
    pointers = np.repeat(view_ptr, ends - starts)
    return dict(pointers=pointers.astype(np.int64), images=v_img.astype(np.int64), atom_pointers=atom_ptr,
                pixels=pixels, features=v_feat)


def map_images(xyz, cams, ref_size, proj_upscale, linearity=None, planarity=None, scattering=None, normals=None):
    """MapImages._process (core/data_transform/multimodal/image.py:238-353, :372-417) with identity candidate
    order, no crop offsets and downscale 1 (the settings of every shipped data config)."""
    image_ids, point_ids, features, pixels = [], [], [], []
    for i_img, cam in enumerate(cams):
        vis = visibility(xyz, cam)
        if vis["idx"].shape[0] == 0:
            continue
        ft = mapping_features(xyz, vis, cam, linearity, planarity, scattering, normals)
        pid = vis["idx"]
        px, py = vis["x"] // proj_upscale, vis["y"] // proj_upscale
        keep = (px >= 0) & (py >= 0) & (px < ref_size[0]) & (py < ref_size[1])
        px, py, pid, ft = px[keep], py[keep], pid[keep], ft[keep]
        u = lexargunique(pid, px, py)
        image_ids.append(i_img)
        point_ids.append(pid[u])
        features.append(ft[u])
        pixels.append(np.stack((px[u], py[u]), axis=1).astype(np.int16))
    seen = np.unique(np.asarray(image_ids, dtype=np.int64))
    ids = np.searchsorted(seen, np.asarray(image_ids, dtype=np.int64))
    ids = np.repeat(ids, [p.shape[0] for p in point_ids])
    dense = dict(point_ids=np.concatenate(point_ids), image_ids=ids, pixels=np.concatenate(pixels),
                 features=np.concatenate(features))
    mapping = from_dense(dense["point_ids"], dense["image_ids"], dense["pixels"], dense["features"], xyz.shape[0])
    return seen, dense, mapping
