"""Online mapping-selection transforms (reference core/data_transform/multimodal/image.py:615-959) on
device-resident mappings, against direct restatements of the reference statements."""
import numpy as np
import pytest
import torch

from conftest import load_golden, t
from test_gpu_data import make_image_data, canonical_pixels

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Data:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def scene():
    g = load_golden("gather")
    x = t(g["x"], DEV)
    sd = make_image_data(g, "", x, g["mapping_size"], DEV)
    n = sd.mappings.num_groups
    gen = torch.Generator().manual_seed(0)
    return g, sd, Data(pos=torch.rand(n, 3, generator=gen).to(DEV), mapping_index=torch.arange(n, device=DEV),
                       num_nodes=n)


def views_as_set(m):
    """{(point, image, pixel tuple...)} -- a canonical form of a mapping."""
    ptr = m.pointers.cpu().numpy()
    img = m.images.cpu().numpy()
    aptr = m.values[1].pointers.cpu().numpy()
    pix = m.pixels.cpu().numpy()
    out = set()
    for p in range(len(ptr) - 1):
        for v in range(ptr[p], ptr[p + 1]):
            out.add((p, int(img[v]), tuple(sorted(map(tuple, pix[aptr[v]:aptr[v + 1]].tolist())))))
    return out


def test_select_mapping_from_point_id():
    from deepviewagg_amd.core.data_transform.multimodal.image import SelectMappingFromPointId
    g, sd, data = scene()
    full = views_as_set(sd.mappings)
    keep = torch.tensor([5, 2, 9, 0], device=DEV)
    data.mapping_index = keep
    data.num_nodes = 4
    data, out = SelectMappingFromPointId()(data, sd)
    assert torch.equal(data.mapping_index.cpu(), torch.arange(4))
    got = views_as_set(out.mappings)
    exp = set()
    # images without any remaining view are dropped and the others renumbered (select_points contract)
    for new_p, old_p in enumerate(keep.tolist()):
        exp |= {(new_p, i, px) for (p, i, px) in full if p == old_p}
    old_imgs = sorted({i for (_, i, _) in exp})
    renum = {o: k for k, o in enumerate(old_imgs)}
    assert got == {(p, renum[i], px) for (p, i, px) in exp}
    assert out.num_views == len(old_imgs)


def test_pick_images_from_mapping_area_and_k_images():
    from deepviewagg_amd.core.data_transform.multimodal.image import PickImagesFromMappingArea, PickKImages
    g, sd, data = scene()
    m = sd.mappings
    aptr = m.values[1].pointers
    pixel_idx = m.images.repeat_interleave(aptr[1:] - aptr[:-1]).cpu()
    areas = torch.bincount(pixel_idx, minlength=sd.num_views).float()
    for ratio, n_max in [(0.0, None), (0.02, 2), (0.5, None)]:
        thr = sd.img_size[0] * sd.img_size[1] * ratio
        order = areas.argsort().flip(0)
        exp = order[areas[order] > thr][: (n_max or sd.num_views)]
        _, out = PickImagesFromMappingArea(area_ratio=ratio, n_max=n_max)(data, sd)
        assert out.num_views == exp.shape[0]
        assert torch.equal(out.pos.cpu(), sd.pos.cpu()[exp])
    # bounding-box variant
    pix = m.pixels.int().cpu()
    exp_area = torch.zeros(sd.num_views)
    for i in range(sd.num_views):
        sel = pix[pixel_idx == i]
        if sel.shape[0]:
            exp_area[i] = float((sel[:, 0].max() - sel[:, 0].min()) * (sel[:, 1].max() - sel[:, 1].min()))
    order = exp_area.argsort().flip(0)
    exp = order[exp_area[order] > 0]
    _, out = PickImagesFromMappingArea(area_ratio=0.0, use_bbox=True)(data, sd)
    assert sorted(out.pos.cpu()[:, 0].tolist()) == sorted(sd.pos.cpu()[exp][:, 0].tolist())
    _, out = PickKImages(2)(data, sd)
    assert out.num_views == len(range(0, sd.num_views, 2))


def test_pick_mappings_from_features_jitter_and_bbox():
    from deepviewagg_amd.core.data_transform.multimodal.image import (
        PickMappingsFromMappingFeatures, JitterMappingFeatures, DropImagesOutsideDataBoundingBox)
    g, sd, data = scene()
    f = sd.mappings.features.clone()
    lo = float(f[:, 0].median())
    _, out = PickMappingsFromMappingFeatures(feat=0, lower=lo)(data, sd.clone())
    assert out.mappings.num_items == int((f[:, 0] > lo).sum())
    assert bool((out.mappings.features[:, 0] > lo).all())
    torch.manual_seed(0)
    before = sd.mappings.features.clone()
    _, out = JitterMappingFeatures(sigma=0.5, clip=0.03)(data, sd)
    d = (out.mappings.features - before).abs()
    assert float(d.max()) <= 0.03 + 1e-6 and float(d.max()) > 0
    sd.pos = torch.tensor([[0.5, 0.5, 0.5]] * sd.num_views, device=DEV)
    sd.pos[0] = torch.tensor([9.0, 9.0, 9.0], device=DEV)
    _, out = DropImagesOutsideDataBoundingBox(margin=0.2)(data, sd)
    assert out.num_views == sd.num_views - 1


def test_pick_images_from_memory_credit():
    from deepviewagg_amd.core.data_transform.multimodal.image import PickImagesFromMemoryCredit
    from deepviewagg_amd.core.multimodal.image import ImageData
    g, sd, data = scene()
    size = sd.img_size[0] * sd.img_size[1]
    np.random.seed(0)
    _, out = PickImagesFromMemoryCredit(credit=2 * size, k_coverage=2)(data, ImageData([sd]))
    assert isinstance(out, ImageData) and out.num_views == 2
    np.random.seed(0)
    _, out = PickImagesFromMemoryCredit(img_size=list(sd.img_size), n_img=sd.num_views + 3)(data, ImageData([sd]))
    assert out.num_views == sd.num_views
    with pytest.raises(ValueError):
        PickImagesFromMemoryCredit()
