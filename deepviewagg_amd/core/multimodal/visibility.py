"""Point -> pixel visibility models backed by the gfx950 mapping-build kernels.

Drop-in mirror of the classes ``MapImages`` resolves by name with
``getattr(visibility_module, self.method)`` (reference:
torch_points3d/core/data_transform/multimodal/image.py:214-215) and calls as
``visi_model(xyz, img_xyz, img_opk=..., img_intrinsic_pinhole=..., img_intrinsic_fisheye=...,
img_extrinsic=..., img_mask=..., linearity=..., planarity=..., scattering=..., normals=...)``
(:273-285).  Semantics follow the reference's CPU/numba path (core/multimodal/visibility.py:478-538,
:630-953, :1073-1195, :1548-1582, :1699-1757), which its authors designate as the reliable one
(README.md:122-123).  The computation itself runs on the HIP device through
``dva_visibility`` / ``dva_mapping_features`` (include/dva.h); there is no CPU implementation here.

Tie-breaks depend on the ORDER of the candidate points in ``xyz`` (first point wins a depth tie,
highest index wins a shared centre pixel in exact mode), exactly as in the reference.
"""
import ctypes
import os

import numpy as np
import torch

from ... import _lib
from ..._lib import DvaCamera, check, ptr, stream_of

CAMERAS = ('s3dis_equirectangular', 'scannet', 'kitti360_perspective', 'kitti360_fisheye')
# DVA_VIS_SINGLE_VIA_BATCH=1: a single-camera call runs as a batch of one on the tiled LDS z-buffer build.  Measured (round 4,
# S3DIS setting, 200 k candidates): 0.37 ms per image against 0.28 ms for the single-image kernels of dva_visibility (64-bit
# atomic z-buffer plane) -- the batched build has more launches (binning, scans) than one image amortises -- so the
# single-image kernels stay the default; the two implementations check each other in tests/test_gpu_mapping.py.
SINGLE_VIA_BATCH = os.environ.get('DVA_VIS_SINGLE_VIA_BATCH', '0') == '1'


def _np32(x):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float32)


def pose_to_rotation_matrix(opk):
    """Inverse rotation matrix of an (omega, phi, kappa) pose, float32 like the reference
    (visibility.py:57-90). Host-side scalar work done once per image."""
    opk = _np32(opk)
    co, so = np.cos(opk[0]), np.sin(opk[0])
    cp, sp = np.cos(opk[1]), np.sin(opk[1])
    ck, sk = np.cos(opk[2]), np.sin(opk[2])
    m_o = np.array([[1.0, 0.0, 0.0], [0.0, co, -so], [0.0, so, co]], dtype=np.float32)
    m_p = np.array([[cp, 0.0, sp], [0.0, 1.0, 0.0], [-sp, 0.0, cp]], dtype=np.float32)
    m_k = np.array([[ck, -sk, 0.0], [sk, ck, 0.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    return np.dot(m_o, np.dot(m_p, m_k))


class VisibilityModel:
    """Base class: camera description + the ``__call__`` contract (visibility.py:1677-1761)."""

    def __init__(self, img_size=(1024, 512), crop_top=0, crop_bottom=0, r_max=30, r_min=0.5,
                 camera='s3dis_equirectangular'):
        self.img_size = img_size
        self.crop_top = crop_top
        self.crop_bottom = crop_bottom
        self.r_max = r_max
        self.r_min = r_min
        self.camera = camera

    # -- host-side per-image preparation -------------------------------------------------------
    def _camera_struct(self, img_xyz, img_opk, img_intrinsic_pinhole, img_intrinsic_fisheye, img_extrinsic):
        if self.camera not in CAMERAS:
            raise ValueError(f"Unknown camera '{self.camera}'")  # visibility.py:526-527
        c = DvaCamera()
        c.model = _lib.CAMERA_CODE[self.camera]
        c.img_w, c.img_h = int(self.img_size[0]), int(self.img_size[1])
        c.crop_top, c.crop_bottom = int(self.crop_top), int(self.crop_bottom)
        c.r_min, c.r_max = float(self.r_min), float(self.r_max)
        c.r_min_d, c.r_max_d = float(self.r_min), float(self.r_max)
        c.voxel = float(getattr(self, 'voxel', 0.1))
        c.k_swell = float(getattr(self, 'k_swell', 1.0))
        c.d_swell = float(getattr(self, 'd_swell', 1000))
        c.exact = int(bool(getattr(self, 'exact', False)))
        c.img_xyz[:] = _np32(img_xyz).reshape(3).tolist()
        rot, trans = np.eye(3, dtype=np.float32), np.zeros(3, dtype=np.float32)
        if self.camera == 's3dis_equirectangular':
            rot = pose_to_rotation_matrix(np.zeros(3) if img_opk is None else img_opk)
        else:
            ext = np.eye(4, dtype=np.float32) if img_extrinsic is None else _np32(img_extrinsic)
            if self.camera == 'scannet':
                # camera_to_world = inv(extrinsic) (visibility.py:232-236)
                ext = np.linalg.inv(np.ascontiguousarray(ext))
            rot, trans = ext[:3, :3].copy(), ext[:3, 3].copy()
        c.rot[:] = rot.astype(np.float32).reshape(-1).tolist()
        c.trans[:] = trans.astype(np.float32).tolist()
        k = np.eye(4, dtype=np.float32) if img_intrinsic_pinhole is None else _np32(img_intrinsic_pinhole)
        c.fx, c.fy, c.mx, c.my = float(k[0, 0]), float(k[1, 1]), float(k[0, 2]), float(k[1, 2])
        fe = np.ones(7, dtype=np.float32) if img_intrinsic_fisheye is None else _np32(img_intrinsic_fisheye)
        c.fisheye[:] = fe.reshape(7).tolist()
        return c

    def __call__(self, xyz, img_xyz, linearity=None, planarity=None, scattering=None, normals=None,
                 img_opk=None, img_intrinsic_pinhole=None, img_intrinsic_fisheye=None, img_extrinsic=None,
                 img_mask=None, **kwargs):
        """Visibility of the point cloud ``xyz`` [n, 3] from one camera.

        :return: dict(idx LongTensor[q] (index into xyz), x, y LongTensor[q] pixel coordinates in the
          non-cropped projection map, depth FloatTensor[q], features FloatTensor[q, F])
        """
        if SINGLE_VIA_BATCH:
            # one camera = a batch of one: the tiled LDS z-buffer of dva_visibility_batch instead of the 64-bit atomic plane
            # of dva_visibility (round 4; rows identical: tests/test_gpu_mapping.py)
            def one(a):
                return None if a is None else torch.as_tensor(a)[None]
            out = self.batch(xyz, torch.as_tensor(img_xyz)[None], linearity=linearity, planarity=planarity,
                             scattering=scattering, normals=normals, img_opk=one(img_opk),
                             img_intrinsic_pinhole=one(img_intrinsic_pinhole),
                             img_intrinsic_fisheye=one(img_intrinsic_fisheye), img_extrinsic=one(img_extrinsic),
                             img_mask=img_mask)
            out.pop('image')
            out.pop('row_ptr')
            if out['idx'].shape[0] == 0:       # visibility.py:1721-1729
                out.pop('x_proj')
                out.pop('y_proj')
                out['features'] = torch.empty((0,), dtype=torch.float, device=xyz.device)
            return out
        lib = _lib.load()
        in_device = xyz.device
        if not torch.cuda.is_available():
            raise _lib.DvaError("the mapping build runs on a HIP device; none is visible "
                                "(deepviewagg_amd has no CPU fallback)")
        dev = in_device if xyz.is_cuda else torch.device('cuda', torch.cuda.current_device())
        assert img_mask is None or tuple(img_mask.shape) == tuple(self.img_size), \
            f'Expected img_mask to be a torch.BoolTensor of shape img_size={self.img_size} but got ' \
            f'size={None if img_mask is None else tuple(img_mask.shape)}.'
        cam = self._camera_struct(img_xyz, img_opk, img_intrinsic_pinhole, img_intrinsic_fisheye, img_extrinsic)

        xyz_d = xyz.detach().to(dev, torch.float32).contiguous()
        n = xyz_d.shape[0]
        mask_d = None if img_mask is None else img_mask.to(dev).to(torch.uint8).contiguous()
        hc = cam.img_h - cam.crop_top - cam.crop_bottom
        cap = n if cam.exact else max(n, cam.img_w * hc)
        cap = max(cap, 1)
        idx = torch.empty(cap, dtype=torch.int64, device=dev)
        x_pix = torch.empty(cap, dtype=torch.int64, device=dev)
        y_pix = torch.empty(cap, dtype=torch.int64, device=dev)
        depth = torch.empty(cap, dtype=torch.float32, device=dev)
        x_proj = torch.empty(cap, dtype=torch.float64, device=dev)
        y_proj = torch.empty(cap, dtype=torch.float64, device=dev)
        n_out = torch.zeros(1, dtype=torch.int64, device=dev)
        ws_bytes = lib.dva_visibility_workspace_bytes(ctypes.byref(cam), n)
        if ws_bytes < 0:
            check(int(ws_bytes), "dva_visibility_workspace_bytes")
        ws = torch.empty(int(ws_bytes), dtype=torch.uint8, device=dev)
        check(lib.dva_visibility(ptr(xyz_d), n, ctypes.byref(cam), ptr(mask_d), ptr(idx), ptr(x_pix),
                                 ptr(y_pix), ptr(depth), ptr(x_proj), ptr(y_proj), ptr(n_out), ptr(ws),
                                 int(ws_bytes), stream_of(xyz_d)), "dva_visibility")
        q = int(n_out.item())   # one host sync per image, like the reference's numpy round trip

        out = {}
        if q == 0:
            # visibility.py:1721-1729
            out['idx'] = torch.empty((0,), dtype=torch.long, device=in_device)
            out['x'] = torch.empty((0,), dtype=torch.long, device=in_device)
            out['y'] = torch.empty((0,), dtype=torch.long, device=in_device)
            out['depth'] = torch.empty((0,), dtype=torch.float, device=in_device)
            out['features'] = torch.empty((0,), dtype=torch.float, device=in_device)
            return out

        def dev32(a):
            return None if a is None else a.detach().to(dev, torch.float32).contiguous()
        lin, pla, sca, nrm = dev32(linearity), dev32(planarity), dev32(scattering), dev32(normals)
        ncol = 2 + sum(a is not None for a in (lin, pla, sca, nrm))
        feats = torch.empty((q, ncol), dtype=torch.float32, device=dev)
        got = ctypes.c_int32(0)
        check(lib.dva_mapping_features(ptr(xyz_d), ptr(idx), ptr(depth), ptr(y_proj), ptr(lin), ptr(pla),
                                       ptr(sca), ptr(nrm), ctypes.byref(cam), q, ptr(feats),
                                       ctypes.byref(got), stream_of(xyz_d)), "dva_mapping_features")
        assert got.value == ncol
        out['idx'] = idx[:q].to(in_device)
        out['x'] = x_pix[:q].to(in_device)
        out['y'] = y_pix[:q].to(in_device)
        out['depth'] = depth[:q].to(in_device)
        out['features'] = feats.to(in_device)
        # float projections of the mapped points (not in the reference's dict; used by parity tests)
        out['x_proj'] = x_proj[:q].to(in_device)
        out['y_proj'] = y_proj[:q].to(in_device)
        return out

    # -- batched build: B cameras of this setting against the same cloud --------------------------------------
    def _camera_array(self, img_xyz, img_opk, img_intrinsic_pinhole, img_intrinsic_fisheye, img_extrinsic):
        """numpy structured array of B ``dva_camera`` (same bytes as B x ``_camera_struct``), filled without a
        Python loop over the images."""
        if self.camera not in CAMERAS:
            raise ValueError(f"Unknown camera '{self.camera}'")
        pos = _np32(img_xyz).reshape(-1, 3)
        B = pos.shape[0]
        cams = np.zeros(B, dtype=np.dtype(DvaCamera))
        cams['model'] = _lib.CAMERA_CODE[self.camera]
        cams['img_w'], cams['img_h'] = int(self.img_size[0]), int(self.img_size[1])
        cams['crop_top'], cams['crop_bottom'] = int(self.crop_top), int(self.crop_bottom)
        cams['r_min'], cams['r_max'] = float(self.r_min), float(self.r_max)
        cams['r_min_d'], cams['r_max_d'] = float(self.r_min), float(self.r_max)
        cams['voxel'] = float(getattr(self, 'voxel', 0.1))
        cams['k_swell'] = float(getattr(self, 'k_swell', 1.0))
        cams['d_swell'] = float(getattr(self, 'd_swell', 1000))
        cams['exact'] = int(bool(getattr(self, 'exact', False)))
        cams['img_xyz'] = pos
        rot = np.tile(np.eye(3, dtype=np.float32), (B, 1, 1))
        trans = np.zeros((B, 3), dtype=np.float32)
        if self.camera == 's3dis_equirectangular':
            opk = np.zeros((B, 3), dtype=np.float32) if img_opk is None else _np32(img_opk).reshape(B, 3)
            co, so = np.cos(opk[:, 0]), np.sin(opk[:, 0])
            cp, sp = np.cos(opk[:, 1]), np.sin(opk[:, 1])
            ck, sk = np.cos(opk[:, 2]), np.sin(opk[:, 2])
            one, zero = np.ones(B, np.float32), np.zeros(B, np.float32)
            m_o = np.stack([one, zero, zero, zero, co, -so, zero, so, co], 1).reshape(B, 3, 3).astype(np.float32)
            m_p = np.stack([cp, zero, sp, zero, one, zero, -sp, zero, cp], 1).reshape(B, 3, 3).astype(np.float32)
            m_k = np.stack([ck, -sk, zero, sk, ck, zero, zero, zero, one], 1).reshape(B, 3, 3).astype(np.float32)
            # one image at a time through np.dot, like pose_to_rotation_matrix: float32 dot products in the same
            # order (a batched matmul may accumulate differently)
            rot = np.stack([np.dot(m_o[b], np.dot(m_p[b], m_k[b])) for b in range(B)])
        else:
            ext = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1)) if img_extrinsic is None \
                else _np32(img_extrinsic).reshape(B, 4, 4)
            if self.camera == 'scannet':
                ext = np.stack([np.linalg.inv(np.ascontiguousarray(e)) for e in ext])
            rot, trans = ext[:, :3, :3].copy(), ext[:, :3, 3].copy()
        cams['rot'] = rot.astype(np.float32).reshape(B, 9)
        cams['trans'] = trans.astype(np.float32)
        k = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1)) if img_intrinsic_pinhole is None \
            else _np32(img_intrinsic_pinhole).reshape(B, 4, 4) if _np32(img_intrinsic_pinhole).size == B * 16 \
            else _np32(img_intrinsic_pinhole).reshape(B, 3, 3)
        cams['fx'], cams['fy'], cams['mx'], cams['my'] = k[:, 0, 0], k[:, 1, 1], k[:, 0, 2], k[:, 1, 2]
        fe = np.ones((B, 7), dtype=np.float32) if img_intrinsic_fisheye is None \
            else _np32(img_intrinsic_fisheye).reshape(B, 7)
        cams['fisheye'] = fe
        return cams

    def batch(self, xyz, img_xyz, linearity=None, planarity=None, scattering=None, normals=None, img_opk=None,
              img_intrinsic_pinhole=None, img_intrinsic_fisheye=None, img_extrinsic=None, img_mask=None, **kwargs):
        """``__call__`` for B cameras of this setting at once (``img_xyz`` [B, 3]; ``img_opk`` [B, 3],
        ``img_intrinsic_pinhole`` [B, 4, 4] | [B, 3, 3], ``img_intrinsic_fisheye`` [B, 7], ``img_extrinsic``
        [B, 4, 4] where the camera model has them): the kernels of the single-image build with an image axis
        (``dva_visibility_batch``), ONE host synchronisation (the total row count) instead of one per image.

        :return: the dict of ``__call__`` with the rows of all images concatenated in image order, plus ``image``
          LongTensor[q] (image of every row) and ``row_ptr`` LongTensor[B + 1] (first row of every image).  Row for
          row identical to B single calls (tests/test_gpu_mapping.py)."""
        lib = _lib.load()
        in_device = xyz.device
        if not torch.cuda.is_available():
            raise _lib.DvaError("the mapping build runs on a HIP device; none is visible "
                                "(deepviewagg_amd has no CPU fallback)")
        dev = in_device if xyz.is_cuda else torch.device('cuda', torch.cuda.current_device())
        assert img_mask is None or tuple(img_mask.shape) == tuple(self.img_size), \
            f'Expected img_mask to be a torch.BoolTensor of shape img_size={self.img_size} but got ' \
            f'size={None if img_mask is None else tuple(img_mask.shape)}.'
        cams = self._camera_array(img_xyz, img_opk, img_intrinsic_pinhole, img_intrinsic_fisheye, img_extrinsic)
        B = cams.shape[0]
        cam0 = DvaCamera.from_buffer_copy(cams[0].tobytes())
        cams_d = torch.from_numpy(cams.view(np.uint8).reshape(B, -1).copy()).to(dev)
        xyz_d = xyz.detach().to(dev, torch.float32).contiguous()
        n = xyz_d.shape[0]
        mask_d = None if img_mask is None else img_mask.to(dev).to(torch.uint8).contiguous()
        hc = cam0.img_h - cam0.crop_top - cam0.crop_bottom
        cap = max((n if cam0.exact else max(n, cam0.img_w * hc)) * B, 1)
        idx = torch.empty(cap, dtype=torch.int64, device=dev)
        x_pix = torch.empty(cap, dtype=torch.int64, device=dev)
        y_pix = torch.empty(cap, dtype=torch.int64, device=dev)
        depth = torch.empty(cap, dtype=torch.float32, device=dev)
        x_proj = torch.empty(cap, dtype=torch.float64, device=dev)
        y_proj = torch.empty(cap, dtype=torch.float64, device=dev)
        row_ptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        n_out = torch.zeros(1, dtype=torch.int64, device=dev)
        ws_bytes = lib.dva_visibility_batch_workspace_bytes(ctypes.byref(cam0), n, B)
        if ws_bytes < 0:
            check(int(ws_bytes), "dva_visibility_batch_workspace_bytes")
        ws = torch.empty(int(ws_bytes), dtype=torch.uint8, device=dev)
        st = stream_of(xyz_d)
        check(lib.dva_visibility_batch(ptr(xyz_d), n, ctypes.byref(cam0), ptr(cams_d), B, ptr(mask_d), ptr(idx),
                                       ptr(x_pix), ptr(y_pix), ptr(depth), ptr(x_proj), ptr(y_proj), ptr(row_ptr),
                                       ptr(n_out), ptr(ws), int(ws_bytes), st), "dva_visibility_batch")
        q = int(n_out.item())          # the one host synchronisation of the batch
        del ws

        def dev32(a):
            return None if a is None else a.detach().to(dev, torch.float32).contiguous()
        lin, pla, sca, nrm = dev32(linearity), dev32(planarity), dev32(scattering), dev32(normals)
        ncol = 2 + sum(a is not None for a in (lin, pla, sca, nrm))
        feats = torch.empty((q, ncol), dtype=torch.float32, device=dev)
        row_image = torch.empty(q, dtype=torch.int32, device=dev)
        got = ctypes.c_int32(0)
        check(lib.dva_mapping_features_batch(ptr(xyz_d), ptr(idx), ptr(depth), ptr(y_proj), ptr(lin), ptr(pla),
                                             ptr(sca), ptr(nrm), ptr(cams_d), ptr(row_ptr), B, q, ptr(feats),
                                             ptr(row_image), ctypes.byref(got), st), "dva_mapping_features_batch")
        assert got.value == ncol
        return {'idx': idx[:q].to(in_device), 'x': x_pix[:q].to(in_device), 'y': y_pix[:q].to(in_device),
                'depth': depth[:q].to(in_device), 'features': feats.to(in_device),
                'x_proj': x_proj[:q].to(in_device), 'y_proj': y_proj[:q].to(in_device),
                'image': row_image.long().to(in_device), 'row_ptr': row_ptr.to(in_device)}

    def __repr__(self):
        attr_repr = ', '.join([f'{k}={v}' for k, v in self.__dict__.items()])
        return f'{self.__class__.__name__}({attr_repr})'


class SplattingVisibility(VisibilityModel):
    """Z-buffered splatting visibility (visibility.py:1764-1776): every point is splatted as a box
    whose size follows its voxel footprint and distance; the closest point wins each pixel; with
    ``exact=True`` only the centre pixel of each winning point is kept."""

    def __init__(self, voxel=0.1, k_swell=1.0, d_swell=1000, exact=False, **kwargs):
        super().__init__(**kwargs)
        self.voxel = voxel
        self.k_swell = k_swell
        self.d_swell = d_swell
        self.exact = exact


class _ProjectionVisibility(VisibilityModel):
    """Visibility models that decide per PROJECTED point (no splatting): the reference's ``VisibilityModel.__call__``
    (visibility.py:1699-1757) with ``camera_projection`` on the device (``dva_camera_projection``), a subclass
    ``_visibility(x_proj, y_proj, dist)`` -> (indices into the projected points, x_pix, y_pix) and the mapping features of
    the kept points (``dva_mapping_features``).  ``x`` / ``y`` of the result are what the reference returns for these
    models: the FLOAT projections of the kept points (visibility.py:1381, :1496), not integer pixels."""

    def _visibility(self, x_proj, y_proj, dist, **kwargs):
        raise NotImplementedError

    def batch(self, xyz, img_xyz, linearity=None, planarity=None, scattering=None, normals=None, img_opk=None,
              img_intrinsic_pinhole=None, img_intrinsic_fisheye=None, img_extrinsic=None, img_mask=None, **kwargs):
        """The ``batch`` contract of ``VisibilityModel`` (rows of all images concatenated, ``image``, ``row_ptr``) as a
        loop over the images, like the reference's MapImages (core/data_transform/multimodal/image.py:238-353): these
        models have no batched kernels."""
        B = torch.as_tensor(img_xyz).reshape(-1, 3).shape[0]

        def one(a, b):
            return None if a is None else torch.as_tensor(a)[b]

        def per_image(b):
            # per-image keyword arguments arrive as lists (depth_map_path / depth_map of DepthBasedVisibility: one file
            # per image, reference core/data_transform/multimodal/image.py:262-285); anything else is shared
            return {k: (v[b] if isinstance(v, (list, tuple)) and len(v) == B and k in ('depth_map_path', 'depth_map') else v)
                    for k, v in kwargs.items()}
        outs = [self(xyz, torch.as_tensor(img_xyz).reshape(-1, 3)[b], linearity=linearity, planarity=planarity,
                     scattering=scattering, normals=normals, img_opk=one(img_opk, b),
                     img_intrinsic_pinhole=one(img_intrinsic_pinhole, b),
                     img_intrinsic_fisheye=one(img_intrinsic_fisheye, b), img_extrinsic=one(img_extrinsic, b),
                     img_mask=img_mask, **per_image(b)) for b in range(B)]
        counts = torch.tensor([o['idx'].shape[0] for o in outs], dtype=torch.long)
        row_ptr = torch.cat((torch.zeros(1, dtype=torch.long), counts.cumsum(0))).to(xyz.device)
        keep = [o for o in outs if o['idx'].shape[0] > 0]
        res = {k: (torch.cat([o[k] for o in keep]) if keep else outs[0][k]) for k in ('idx', 'x', 'y', 'depth', 'features')}
        res['image'] = torch.repeat_interleave(torch.arange(B), counts).to(xyz.device)
        res['row_ptr'] = row_ptr
        return res

    def __call__(self, xyz, img_xyz, linearity=None, planarity=None, scattering=None, normals=None,
                 img_opk=None, img_intrinsic_pinhole=None, img_intrinsic_fisheye=None, img_extrinsic=None,
                 img_mask=None, **kwargs):
        lib = _lib.load()
        in_device = xyz.device
        if not torch.cuda.is_available():
            raise _lib.DvaError("the mapping build runs on a HIP device; none is visible "
                                "(deepviewagg_amd has no CPU fallback)")
        dev = in_device if xyz.is_cuda else torch.device('cuda', torch.cuda.current_device())
        assert img_mask is None or tuple(img_mask.shape) == tuple(self.img_size)
        cam = self._camera_struct(img_xyz, img_opk, img_intrinsic_pinhole, img_intrinsic_fisheye, img_extrinsic)
        xyz_d = xyz.detach().to(dev, torch.float32).contiguous()
        n = xyz_d.shape[0]
        mask_d = None if img_mask is None else img_mask.to(dev).to(torch.uint8).contiguous()
        cap = max(n, 1)
        idx1 = torch.empty(cap, dtype=torch.int64, device=dev)
        dist = torch.empty(cap, dtype=torch.float32, device=dev)
        x_proj = torch.empty(cap, dtype=torch.float64, device=dev)
        y_proj = torch.empty(cap, dtype=torch.float64, device=dev)
        n_out = torch.zeros(1, dtype=torch.int64, device=dev)
        ws_bytes = lib.dva_visibility_workspace_bytes(ctypes.byref(cam), n)
        if ws_bytes < 0:
            check(int(ws_bytes), "dva_visibility_workspace_bytes")
        ws = torch.empty(int(ws_bytes), dtype=torch.uint8, device=dev)
        st = stream_of(xyz_d)
        check(lib.dva_camera_projection(ptr(xyz_d), n, ctypes.byref(cam), ptr(mask_d), ptr(idx1), ptr(dist),
                                        ptr(x_proj), ptr(y_proj), ptr(n_out), ptr(ws), int(ws_bytes), st),
              "dva_camera_projection")
        m = int(n_out.item())
        del ws
        out = {}
        if m == 0:          # visibility.py:1721-1729
            for k, dt in (('idx', torch.long), ('x', torch.long), ('y', torch.long), ('depth', torch.float),
                          ('features', torch.float)):
                out[k] = torch.empty((0,), dtype=dt, device=in_device)
            return out
        idx1, dist, x_proj, y_proj = idx1[:m], dist[:m], x_proj[:m], y_proj[:m]
        idx2, x_pix, y_pix = self._visibility(x_proj, y_proj, dist, **kwargs)
        idx = idx1[idx2].contiguous()
        dist, y_kept = dist[idx2].contiguous(), y_proj[idx2].contiguous()
        q = idx.shape[0]

        def dev32(a):
            return None if a is None else a.detach().to(dev, torch.float32).contiguous()
        lin, pla, sca, nrm = dev32(linearity), dev32(planarity), dev32(scattering), dev32(normals)
        ncol = 2 + sum(a is not None for a in (lin, pla, sca, nrm))
        feats = torch.empty((q, ncol), dtype=torch.float32, device=dev)
        if q > 0:
            got = ctypes.c_int32(0)
            check(lib.dva_mapping_features(ptr(xyz_d), ptr(idx), ptr(dist), ptr(y_kept), ptr(lin), ptr(pla), ptr(sca),
                                           ptr(nrm), ctypes.byref(cam), q, ptr(feats), ctypes.byref(got), st),
                  "dva_mapping_features")
            assert got.value == ncol
        out['idx'], out['x'], out['y'] = idx.to(in_device), x_pix.to(in_device), y_pix.to(in_device)
        out['depth'], out['features'] = dist.to(in_device), feats.to(in_device)
        return out


def read_s3dis_depth_map(path, img_size=None, empty=-1):
    """S3DIS depth panorama (16-bit PNG, 1/512 m, 2^16 - 1 = missing) as a float [W, H] tensor in metres
    (visibility.py:1326-1355)."""
    from PIL import Image
    im = Image.open(path)
    if img_size is not None:
        im = im.resize(tuple(int(v) for v in img_size), resample=Image.NEAREST)
    im = torch.from_numpy(np.array(im).astype(np.int32)).t()
    empty_mask = im == 2 ** 16 - 1
    im = im / 512
    im[empty_mask] = empty
    return im


class DepthBasedVisibility(_ProjectionVisibility):
    """A projected point is visible when its distance lies within ``depth_threshold`` of the depth map's value at its
    pixel (visibility.py:1356-1383, :1779-1787).  The map comes from ``depth_map_path`` (S3DIS format, as in the reference)
    or, already loaded, as ``depth_map`` float [W, H]."""

    def __init__(self, depth_threshold=0.05, **kwargs):
        super().__init__(**kwargs)
        self.depth_threshold = depth_threshold

    def _visibility(self, x_proj, y_proj, dist, depth_map_path=None, depth_map=None, **kwargs):
        if depth_map is None:
            assert depth_map_path is not None, 'Please provide depth_map_path.'      # visibility.py:1374
            depth_map = read_s3dis_depth_map(depth_map_path, img_size=self.img_size, empty=-1)
        depth_map = depth_map.to(x_proj.device)
        dist_real = depth_map[x_proj.long(), y_proj.long()]
        indices = torch.where((dist_real - dist).abs() <= self.depth_threshold)[0]
        return indices, x_proj[indices], y_proj[indices]


class BiasuttiVisibility(_ProjectionVisibility):
    """Biasutti et al., "Visibility estimation in point clouds with variable density" (visibility.py:1390-1496, :1790-1803):
    alpha = exp(-((d - d_min) / (d_max - d_min))^2) over the k nearest projected points in IMAGE coordinates (wrapped in x
    by ``margin`` pixels for panoramas), visible when alpha >= ``threshold`` (default: the mean alpha).  The neighbour
    search is the exact hash-grid K-NN of ``dva_knn`` on (x, y, 0) -- the reference's KeOps argKmin over all pairs, fp32
    squared distances, ties to the lower index."""

    def __init__(self, k=75, margin=None, threshold=None, **kwargs):
        super().__init__(**kwargs)
        self.k = k
        self.margin = margin
        self.threshold = threshold

    def _neighbors(self, x_proj, y_proj):
        from ... import ops
        n = x_proj.shape[0]
        xy = torch.stack((x_proj.float(), y_proj.float())).t()
        x_width, x_margin = self.img_size[0], self.margin
        wrap = x_margin is not None and x_margin > 0 and x_width is not None and x_width > 0
        if wrap:
            off = torch.tensor([[float(x_width), 0.0]], device=xy.device)
            idx_left = torch.where(x_proj <= x_margin)[0]
            idx_right = torch.where(x_proj >= (x_width - x_margin))[0]
            search = torch.cat((xy, xy[idx_left] + off, xy[idx_right] - off))
        else:
            search = xy
        k = min(int(self.k), search.shape[0])
        pts = torch.cat((search, torch.zeros((search.shape[0], 1), device=xy.device)), 1).contiguous()
        nbr, _ = ops.knn(pts, k)                        # self-search over the search set; the queries are its first n rows
        nbr = nbr[:n].long()
        if wrap:
            n_left = idx_left.shape[0]
            is_left = (nbr >= n) & (nbr < n + n_left)
            nbr[is_left] = idx_left[nbr[is_left] - n]
            is_right = nbr >= n + n_left
            nbr[is_right] = idx_right[nbr[is_right] - n - n_left]
        return nbr

    def _visibility(self, x_proj, y_proj, dist, **kwargs):
        neighbors = self._neighbors(x_proj, y_proj)
        dist_nn = dist[neighbors]
        dist_min, dist_max = dist_nn.min(dim=1).values, dist_nn.max(dim=1).values
        alpha = torch.exp(-((dist - dist_min) / (dist_max - dist_min)) ** 2)
        threshold = alpha.mean() if self.threshold is None else self.threshold
        indices = torch.where(alpha >= threshold)[0]
        return indices, x_proj[indices], y_proj[indices]
