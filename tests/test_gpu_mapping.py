"""-m gpu: the HIP mapping build (dva_visibility / dva_mapping_features / lex ops, through the C ABI)
against the reference's golden vectors and against the C oracle at full projection-map size.

Bit-exact: point indices, pixel coordinates, depths, lex orders.  Float projections: the GPU's
float64 libm (OCML) vs glibc may differ in the last bit of a float64 intermediate -> compared at
1e-9 px; mapping features at 2.5e-7.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, t
from oracle import mapping_oracle as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VIS = ["vis_equirect_exact", "vis_equirect_dense", "vis_equirect_rot_crop_mask", "vis_equirect_bigsplat",
       "vis_pinhole_scannet", "vis_pinhole_kitti", "vis_fisheye_kitti", "vis_equirect_empty"]


def model_of(g):
    from deepviewagg_amd.core.multimodal.visibility import SplattingVisibility
    return SplattingVisibility(
        img_size=tuple(int(v) for v in g["img_size"]), crop_top=int(g["crop"][0]), crop_bottom=int(g["crop"][1]),
        r_max=float(g["r_max"]), r_min=float(g["r_min"]), camera=str(g["camera"]), voxel=float(g["voxel"]),
        k_swell=float(g["k_swell"]), d_swell=float(g["d_swell"]), exact=bool(g["exact"]))


def call_kwargs(g, dev):
    kw = {k: t(g[k], dev) for k in ("img_opk", "img_extrinsic", "img_intrinsic_pinhole", "img_intrinsic_fisheye")
          if k in g}
    if "img_mask" in g:
        kw["img_mask"] = t(g["img_mask"], dev)
    return kw


@pytest.fixture
def single_image_kernels():
    """Single-camera calls on the single-image kernels of dva_visibility (64-bit atomic z-buffer plane) instead of a batch
    of one on the tiled build: the two implementations check each other."""
    from deepviewagg_amd.core.multimodal import visibility as V
    old, V.SINGLE_VIA_BATCH = V.SINGLE_VIA_BATCH, False
    yield
    V.SINGLE_VIA_BATCH = old


@pytest.fixture
def batch_of_one():
    """Single-camera calls as a batch of one on the tiled z-buffer build (DVA_VIS_SINGLE_VIA_BATCH=1)."""
    from deepviewagg_amd.core.multimodal import visibility as V
    old, V.SINGLE_VIA_BATCH = V.SINGLE_VIA_BATCH, True
    yield
    V.SINGLE_VIA_BATCH = old


@pytest.mark.parametrize("name", VIS)
def test_visibility_golden_batch_of_one(name, batch_of_one):
    test_visibility_golden(name)


@pytest.mark.parametrize("name", VIS)
def test_visibility_golden(name):
    g = load_golden(name)
    model = model_of(g)
    out = model(t(g["xyz"], DEV), t(g["img_xyz"], DEV), linearity=t(g["linearity"], DEV),
                planarity=t(g["planarity"], DEV), scattering=t(g["scattering"], DEV),
                normals=t(g["normals"], DEV), **call_kwargs(g, DEV))
    for k in ("idx", "x", "y"):
        assert out[k].dtype == torch.int64
        assert np.array_equal(out[k].cpu().numpy(), g[k]), k
    assert np.array_equal(out["depth"].cpu().numpy(), g["depth"])
    if len(g["idx"]):
        np.testing.assert_allclose(out["features"].cpu().numpy(), g["features"], rtol=0, atol=2.5e-7)
    else:
        assert out["features"].shape == (0,)
    # CPU inputs are accepted (uploaded, computed on the device, returned on the CPU) like the
    # reference's use_cuda path
    out2 = model(t(g["xyz"]), t(g["img_xyz"]), **call_kwargs(g, "cpu"))
    assert out2["idx"].device.type == "cpu" and np.array_equal(out2["idx"].numpy(), g["idx"])


def room_cloud(n, rng, size=(8.0, 6.0, 3.0)):
    face = rng.integers(0, 6, n)
    uvw = rng.random((n, 3))
    uvw[np.arange(n), face // 2] = face % 2
    return (uvw * np.array(size) + np.clip(rng.normal(0, 1e-3, (n, 3)), -0.05, 0.05)).astype(np.float32)


@pytest.mark.parametrize("exact", [True, False])
def test_visibility_full_size_vs_oracle(exact):
    """S3DIS settings: 2048x1024 projection map, 200k candidates, voxel 2 cm."""
    from deepviewagg_amd.core.multimodal.visibility import SplattingVisibility
    rng = np.random.default_rng(0)
    xyz = room_cloud(200_000, rng)
    cam_xyz = np.array([3.1, 2.2, 1.4], dtype=np.float32)
    opk = np.array([0.02, -0.01, 0.7], dtype=np.float32)
    kw = dict(img_size=(2048, 1024), crop_top=0, crop_bottom=0, r_max=8.0, r_min=0.05, voxel=0.02,
              k_swell=1.0, d_swell=1000, exact=exact)
    cam = M.make_camera("s3dis_equirectangular", kw["img_size"], cam_xyz, r_min=kw["r_min"], r_max=kw["r_max"],
                        voxel=kw["voxel"], k_swell=1.0, d_swell=1000, exact=exact, img_opk=opk)
    ref = M.visibility(xyz, cam)
    model = SplattingVisibility(camera="s3dis_equirectangular", **kw)
    out = model(t(xyz, DEV), t(cam_xyz, DEV), img_opk=t(opk, DEV))
    assert len(ref["idx"]) > 10000
    for k in ("idx", "x", "y", "depth"):
        assert np.array_equal(out[k].cpu().numpy(), ref[k]), k
    np.testing.assert_allclose(out["x_proj"].cpu().numpy(), ref["x_proj"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["y_proj"].cpu().numpy(), ref["y_proj"], rtol=0, atol=1e-9)
    # property: in exact mode every mapped pixel is unique and equals the truncated projection
    if exact:
        key = out["x"].cpu().numpy() * 1024 + out["y"].cpu().numpy()
        assert len(np.unique(key)) == len(key)
        assert np.array_equal(out["x"].cpu().numpy(), out["x_proj"].cpu().numpy().astype(np.int64))
    # run-to-run determinism (atomics must not change the result)
    out2 = model(t(xyz, DEV), t(cam_xyz, DEV), img_opk=t(opk, DEV))
    assert torch.equal(out["idx"], out2["idx"]) and torch.equal(out["x"], out2["x"])


def test_lex_ops_golden_and_random():
    from deepviewagg_amd.utils import multimodal as U
    g = load_golden("lex_csr")
    a, b, c = t(g["a"], DEV), t(g["b"], DEV), t(g["c"], DEV)
    assert np.array_equal(U.CompositeTensor(a, b, c).data.cpu().numpy(), g["composite"])
    assert np.array_equal(U.lexargunique(a, b, c).cpu().numpy(), g["argunique"])
    ua, ub, uc = U.lexunique(a, b, c)
    assert np.array_equal(ua.cpu().numpy(), g["unique_a"]) and uc.dtype == torch.int16
    assert np.array_equal(uc.cpu().numpy(), g["unique_c"])
    sa, sb, sc = U.lexsort(a, b, c)
    assert np.array_equal(sa.cpu().numpy(), g["sort_a"]) and np.array_equal(sc.cpu().numpy(), g["sort_c"])
    order = U.lexargsort(a, b, c)
    assert np.array_equal(g["composite"][order.cpu().numpy()], g["argsort_keys"])
    # 2M random triples vs numpy (reference's own cross-check pattern, utils/multimodal.py:326-379)
    rng = np.random.default_rng(1)
    cols = [rng.integers(0, 1000, 2_000_000) for _ in range(3)]
    dev_cols = [t(x, DEV) for x in cols]
    key = M.composite(*cols)
    assert np.array_equal(U.lexargunique(*dev_cols).cpu().numpy(), np.unique(key, return_index=True)[1])
    assert np.array_equal(U.lexargsort(*dev_cols).cpu().numpy(), np.argsort(key, kind="stable"))
    # empty input
    e = torch.zeros(0, dtype=torch.long, device=DEV)
    assert U.lexargunique(e, e).shape == (0,)


# ---------------------------------------------------------------------------------------------------------------
# Batched build (dva_visibility_batch): B images of one setting in one set of launches
# ---------------------------------------------------------------------------------------------------------------
def _batch_inputs(g, B, slot, dev):
    """B cameras of the fixture's setting: the fixture's own camera at position ``slot``, the others perturbed."""
    rng = np.random.default_rng(B * 10 + slot)
    kw = call_kwargs(g, dev)
    pos = t(g["img_xyz"], dev).float().view(1, 3).repeat(B, 1)
    pos = pos + torch.from_numpy(rng.normal(0, 0.3, (B, 3)).astype(np.float32)).to(dev)
    pos[slot] = t(g["img_xyz"], dev).float()
    out = {}
    for k, v in kw.items():
        if k == "img_mask":
            out[k] = v
            continue
        vb = v.float().unsqueeze(0).repeat(B, *([1] * v.dim())).clone()
        if k == "img_opk":
            vb += torch.from_numpy(rng.normal(0, 0.2, tuple(vb.shape)).astype(np.float32)).to(dev)
        elif k == "img_extrinsic":
            vb[:, :3, 3] += torch.from_numpy(rng.normal(0, 0.3, (B, 3)).astype(np.float32)).to(dev)
        vb[slot] = v.float()
        out[k] = vb
    return pos, out


@pytest.mark.parametrize("name", VIS)
def test_visibility_batch_golden(name, single_image_kernels):
    """Every reference fixture through the batched build: the fixture's camera sits among four perturbed cameras of
    the same setting; its rows must be the fixture's (bit-exact indices, pixels, depths), and every image's rows must
    equal the single-image build of that camera."""
    g = load_golden(name)
    model = model_of(g)
    B, slot = 5, 3
    pos, kw = _batch_inputs(g, B, slot, DEV)
    attrs = dict(linearity=t(g["linearity"], DEV), planarity=t(g["planarity"], DEV),
                 scattering=t(g["scattering"], DEV), normals=t(g["normals"], DEV))
    xyz = t(g["xyz"], DEV)
    out = model.batch(xyz, pos, **attrs, **kw)
    rp = out["row_ptr"].cpu().numpy()
    assert rp[0] == 0 and rp[-1] == out["idx"].shape[0] and (np.diff(rp) >= 0).all()
    assert np.array_equal(out["image"].cpu().numpy(), np.repeat(np.arange(B), np.diff(rp)))
    a, b = rp[slot], rp[slot + 1]
    for k in ("idx", "x", "y"):
        assert out[k].dtype == torch.int64
        assert np.array_equal(out[k][a:b].cpu().numpy(), g[k]), k
    assert np.array_equal(out["depth"][a:b].cpu().numpy(), g["depth"])
    if len(g["idx"]):
        np.testing.assert_allclose(out["features"][a:b].cpu().numpy(), g["features"], rtol=0, atol=2.5e-7)
    for i in range(B):
        kw_i = {k: (v if k == "img_mask" else v[i]) for k, v in kw.items()}
        one = model(xyz, pos[i], **attrs, **kw_i)
        a, b = rp[i], rp[i + 1]
        for k in ("idx", "x", "y", "depth"):
            assert torch.equal(out[k][a:b], one[k]), (i, k)
        if b > a:
            assert torch.equal(out["features"][a:b], one["features"]), i
            assert torch.equal(out["x_proj"][a:b], one["x_proj"]) and torch.equal(out["y_proj"][a:b], one["y_proj"])


@pytest.mark.parametrize("exact", [True, False])
def test_visibility_batch_full_size_equals_single(exact, single_image_kernels):
    """S3DIS settings (2048 x 1024 projection map, 100 k candidates), 6 cameras: the batch is the concatenation of the
    single-image builds, bit for bit; image 0 is also held to the C oracle."""
    from deepviewagg_amd.core.multimodal.visibility import SplattingVisibility
    rng = np.random.default_rng(1)
    xyz = room_cloud(100_000, rng)
    cams = np.array([[3.1, 2.2, 1.4], [5.0, 3.0, 1.2], [2.0, 4.5, 1.6], [6.5, 1.5, 1.5], [4.0, 3.0, 2.0], [1.0, 1.0, 1.0]],
                    dtype=np.float32)
    opk = rng.normal(0, 0.3, (6, 3)).astype(np.float32)
    kw = dict(img_size=(2048, 1024), crop_top=0, crop_bottom=0, r_max=8.0, r_min=0.05, voxel=0.02, k_swell=1.0,
              d_swell=1000, exact=exact)
    model = SplattingVisibility(camera="s3dis_equirectangular", **kw)
    xyz_d = torch.from_numpy(xyz).to(DEV)
    out = model.batch(xyz_d, torch.from_numpy(cams).to(DEV), img_opk=torch.from_numpy(opk).to(DEV))
    rp = out["row_ptr"].cpu().numpy()
    for i in range(6):
        one = model(xyz_d, torch.from_numpy(cams[i]).to(DEV), img_opk=torch.from_numpy(opk[i]).to(DEV))
        a, b = rp[i], rp[i + 1]
        assert b - a == one["idx"].shape[0]
        for k in ("idx", "x", "y", "depth", "features"):
            assert torch.equal(out[k][a:b], one[k]), (i, k)
    cam0 = M.make_camera("s3dis_equirectangular", kw["img_size"], cams[0], r_min=kw["r_min"], r_max=kw["r_max"],
                         voxel=kw["voxel"], k_swell=1.0, d_swell=1000, exact=exact, img_opk=opk[0])
    ref = M.visibility(xyz, cam0)
    for k in ("idx", "x", "y"):
        assert np.array_equal(out[k][rp[0]:rp[1]].cpu().numpy(), ref[k]), k


@pytest.mark.parametrize("exact", [True, False])
def test_visibility_batch_tiled_zbuffer_fallbacks(exact, single_image_kernels):
    """The tiled z-buffer of the batched build under stress: camera 0 stands inside a dense cluster of points (thousands
    of splat boxes cover more than 16 screen tiles -- beyond the 4096 slots of the per-image large-box list -- and the
    mid-sized ones overflow the tile lists' capacity of 4 entries per candidate): those survivors fall back to the
    atomic plane, which the tile kernel merges.  Every image must still equal its single-image build (the plain atomic
    z-buffer) bit for bit, and image 0 the C oracle."""
    from deepviewagg_amd.core.multimodal.visibility import SplattingVisibility
    rng = np.random.default_rng(7)
    cams = np.array([[3.1, 2.2, 1.4], [5.0, 3.0, 1.2], [2.0, 4.5, 1.6]], dtype=np.float32)


    def shell(n, r0, r1):
        u = rng.normal(0, 1, (n, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        return (cams[0] + u * rng.uniform(r0, r1, (n, 1))).astype(np.float32)
    near = shell(6000, 0.20, 0.30)           # boxes of 5 .. 8 tiles per axis: "large"
    mid = shell(20000, 0.40, 0.48)           # boxes of 3 .. 4 tiles per axis: 9 .. 16 list entries each
    xyz = np.concatenate([room_cloud(20_000, rng), near, mid]).astype(np.float32)
    xyz = xyz[rng.permutation(len(xyz))]
    opk = rng.normal(0, 0.3, (3, 3)).astype(np.float32)
    kw = dict(img_size=(1024, 512), crop_top=0, crop_bottom=0, r_max=8.0, r_min=0.05, voxel=0.15, k_swell=1.0,
              d_swell=1000, exact=exact)
    # the scene really exercises the fallbacks (lower bounds from the splat formula, SURVEY.md A.1 step 4: w_y = a H / pi,
    # w_x >= a W / (2 pi 1.001), a = (1 + k exp(-d / ln d_swell)) voxel / d): large boxes of image 0 and list entries
    d = np.linalg.norm(xyz - cams[0], axis=1)
    d = d[(d > 0.05) & (d < 8.0)]
    ang = (1 + np.exp(-d / np.log(1000))) * 0.15 / d
    ty, tx = np.maximum(ang * 512 / np.pi / 32, 1), np.maximum(ang * 1024 / (2 * np.pi * 1.001) / 32, 1)
    assert int((tx * ty > 16).sum()) > 4096                      # more than ZT_BIGCAP large boxes
    assert float(np.where(tx * ty <= 16, tx * ty, 0).sum()) > 4 * len(xyz)      # more entries than one image's capacity
    model = SplattingVisibility(camera="s3dis_equirectangular", **kw)
    xyz_d = torch.from_numpy(xyz).to(DEV)
    out = model.batch(xyz_d, torch.from_numpy(cams).to(DEV), img_opk=torch.from_numpy(opk).to(DEV))
    rp = out["row_ptr"].cpu().numpy()
    for i in range(3):
        one = model(xyz_d, torch.from_numpy(cams[i]).to(DEV), img_opk=torch.from_numpy(opk[i]).to(DEV))
        a, b = rp[i], rp[i + 1]
        assert b - a == one["idx"].shape[0] and b > a
        for k in ("idx", "x", "y", "depth"):
            assert torch.equal(out[k][a:b], one[k]), (i, k)
    # a batch of ONE image: its list capacity (4 entries per candidate) overflows as well
    solo = model.batch(xyz_d, torch.from_numpy(cams[:1]).to(DEV), img_opk=torch.from_numpy(opk[:1]).to(DEV))
    for k in ("idx", "x", "y", "depth"):
        assert torch.equal(solo[k], out[k][rp[0]:rp[1]]), k
    cam0 = M.make_camera("s3dis_equirectangular", kw["img_size"], cams[0], r_min=kw["r_min"], r_max=kw["r_max"],
                         voxel=kw["voxel"], k_swell=1.0, d_swell=1000, exact=exact, img_opk=opk[0])
    ref = M.visibility(xyz, cam0)
    for k in ("idx", "x", "y"):
        assert np.array_equal(out[k][rp[0]:rp[1]].cpu().numpy(), ref[k]), k
