"""CPU ORACLE (test infrastructure, NOT product code): plain restatements of the two visibility models that decide per
projected point, given the output of ``camera_projection`` (x_proj, y_proj fp64, dist fp32):

  visibility_from_depth_map   /root/reference/torch_points3d/core/multimodal/visibility.py:1356-1383
  k_nn_image_system           :1390-1455   (KeOps argKmin over all pairs; ties -> lower index, like oracle/shims/pykeops)
  visibility_biasutti         :1458-1496

Pinned by tests/golden/vis_biasutti*.npz / vis_depth_map.npz, written by oracle/gen_golden.py from the reference's own
classes (tests/test_oracle_golden.py).  Only tests/ may import this module."""
import numpy as np


def depth_map_visibility(x_proj, y_proj, dist, depth_map, depth_threshold):
    """depth_map float [W, H] (metres, empty pixels < 0): indices of the projected points within the threshold."""
    real = depth_map[x_proj.astype(np.int64), y_proj.astype(np.int64)].astype(np.float32)
    return np.where(np.abs(real - dist.astype(np.float32)) <= depth_threshold)[0]


def knn_image(x_proj, y_proj, k, x_margin=None, x_width=None):
    """Brute-force k nearest neighbours in image coordinates (fp32 squared distances dx^2 + dy^2, ascending, ties to the
    lower index); with a margin the image is wrapped along x and copies map back to their originals."""
    xy = np.stack((x_proj.astype(np.float32), y_proj.astype(np.float32)), 1)
    n = xy.shape[0]
    wrap = x_margin is not None and x_margin > 0 and x_width is not None and x_width > 0
    if wrap:
        off = np.array([[x_width, 0]], dtype=np.float32)
        left = np.where(x_proj <= x_margin)[0]
        right = np.where(x_proj >= (x_width - x_margin))[0]
        search = np.concatenate((xy, xy[left] + off, xy[right] - off))
    else:
        search = xy
    out = np.empty((n, k), dtype=np.int64)
    for lo in range(0, n, 512):
        d = xy[lo:lo + 512, None, :] - search[None, :, :]
        d2 = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
        out[lo:lo + 512] = np.argsort(d2, axis=1, kind='stable')[:, :k]
    if wrap:
        nl = left.shape[0]
        is_l = (out >= n) & (out < n + nl)
        out[is_l] = left[out[is_l] - n]
        is_r = out >= n + nl
        out[is_r] = right[out[is_r] - n - nl]
    return out


def biasutti_visibility(x_proj, y_proj, dist, img_size, k=75, margin=None, threshold=None):
    nbr = knn_image(x_proj, y_proj, k, margin, img_size[0])
    dist = dist.astype(np.float32)
    dnn = dist[nbr]
    dmin, dmax = dnn.min(1), dnn.max(1)
    with np.errstate(invalid='ignore', divide='ignore'):
        alpha = np.exp(-((dist - dmin) / (dmax - dmin)) ** 2).astype(np.float32)
    # torch's mean of a tensor with NaN is NaN (then nothing passes): same here
    thr = np.float32(alpha.mean(dtype=np.float32)) if threshold is None else threshold
    return np.where(alpha >= thr)[0], alpha
