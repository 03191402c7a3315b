"""Host mirror of the MATCHING stage of ``feabas.stitcher.Stitcher`` -- the caller directly above the NCC hot path
(SURVEY.md sec.3): which tiles overlap (``find_overlaps``, stitcher.py:418-437) and, for a list of overlaps, the strips
that are cropped, matched and shifted back into tile coordinates (``subprocess_match_list_of_overlaps``,
stitcher.py:478-613).  The reference handles one pair at a time inside a process pool; here all pairs of a section go
through ``matcher.stitching_matcher_batch`` (shape buckets -> device batches).

Image IO stays outside: ``images`` is a sequence of tile arrays already in memory, or any object with the
``crop(bbox, index, return_index=False)`` method of ``dal.StaticImageLoader``.
"""
import numpy as np
from scipy.ndimage import binary_dilation

from . import common
from .matcher import stitching_matcher_batch

MARGIN_RATIO_SWITCH = 2              # feabas/stitcher.py:32


def bbox_intersections(bboxes0, bboxes1):
    """feabas/common.py:699-704"""
    bboxes0 = np.asarray(bboxes0); bboxes1 = np.asarray(bboxes1)
    xy_min = np.maximum(bboxes0[..., :2], bboxes1[..., :2])
    xy_max = np.minimum(bboxes0[..., -2:], bboxes1[..., -2:])
    return np.concatenate((xy_min, xy_max), axis=-1), np.min(xy_max - xy_min, axis=-1)


def find_overlaps(bboxes, tile_size=None):
    """stitcher.py:418-437: every pair (k, hit), hit < k, whose bounding boxes intersect (closed boxes, like the rtree
    query: touching counts), sorted along the z-order of the overlap centres.  tile_size = (height, width) of an average
    tile (Stitcher.average_tile_size); default: the mean box size."""
    bboxes = np.asarray(bboxes)
    n = bboxes.shape[0]
    pairs = []
    for k in range(1, n):
        b = bboxes[k]
        prev = bboxes[:k]
        hit = (prev[:, 0] <= b[2]) & (prev[:, 2] >= b[0]) & (prev[:, 1] <= b[3]) & (prev[:, 3] >= b[1])
        pairs.extend((k, int(h)) for h in np.flatnonzero(hit))
    if not pairs:
        return np.empty((0, 2), dtype=np.int32)
    overlaps = np.array(pairs)
    bbox_ov, _ = bbox_intersections(bboxes[overlaps[:, 0]], bboxes[overlaps[:, 1]])
    ctr = common.bbox_centers(bbox_ov)
    if tile_size is None:
        sz = common.bbox_sizes(bboxes).mean(axis=0)              # (height, width)
    else:
        sz = np.asarray(tile_size, dtype=np.float64)
    step = sz[::-1] / 2
    idx = np.round((ctr - ctr.min(axis=0)) / step)
    return overlaps[common.z_order(idx)]


class _ArrayLoader:
    """tiles held as arrays: crop(bbox, index) like dal.StaticImageLoader for boxes inside the tile"""

    def __init__(self, images, bboxes):
        self.images = images
        self.bboxes = np.asarray(bboxes)

    def crop(self, bbox, index, return_index=False):
        x0, y0 = self.bboxes[index][:2]
        xa, ya, xb, yb = (int(v) for v in bbox)
        return self.images[index][ya - y0:yb - y0, xa - x0:xb - x0]


def match_list_of_overlaps(overlaps, images, bboxes, **kwargs):
    """stitcher.py:478-613.  overlaps [M, 2] tile indices, bboxes [N, 4] integer tile boxes (xmin, ymin, xmax, ymax);
    kwargs as the reference's: ``min_overlap_width``, ``maskout_val``, ``margin`` (<= 2: a ratio of the overlap width),
    ``index_mapper``, ``matcher_config``; plus ``batch`` / ``threads`` of the device batches.
    Returns ``(matches, strains, brightness_contrast, err_raised)``: matches[(i, j)] = (xy0, xy1, weight) in tile
    coordinates, strains[(i, j)], brightness_contrast[(i, j)] (when the matcher computes photometric statistics).
    A pair the device path cannot take (e.g. an odd strip size with coarse_downsample = 0.5) is counted as an error, like
    the exceptions the reference catches per pair (stitcher.py:604-613), and the rest of the list is processed."""
    min_width = kwargs.get('min_overlap_width', 0)
    maskout_val = kwargs.get('maskout_val', None)
    index_mapper = kwargs.get('index_mapper', None)
    margin = kwargs.get('margin', 1.0)
    matcher_config = dict(kwargs.get('matcher_config', {}))
    overlaps = np.asarray(overlaps).reshape(-1, 2)
    bboxes = np.asarray(bboxes)
    if overlaps.shape[0] == 0:
        return {}, {}, {}, False
    loader = images if hasattr(images, 'crop') else _ArrayLoader(images, bboxes)
    bboxes_overlap, wds = bbox_intersections(bboxes[overlaps[:, 0]], bboxes[overlaps[:, 1]])
    jobs, pairs = [], []
    err_count = 0
    first_err = None
    for (idx0, idx1), bbox_ov, wd in zip(overlaps, bboxes_overlap, wds):
        if wd <= min_width:
            continue
        real_margin = int(margin * wd) if margin <= MARGIN_RATIO_SWITCH else int(margin)
        bbox_ov = bbox_ov + np.array([-real_margin, -real_margin, real_margin, real_margin])     # common.bbox_enlarge
        bbox0, bbox1 = bboxes[idx0], bboxes[idx1]
        bbox_ov0 = bbox_intersections(bbox_ov, bbox0)[0]
        bbox_ov1 = bbox_intersections(bbox_ov, bbox1)[0]
        img0 = loader.crop(bbox_ov0, int(idx0), return_index=False)
        img1 = loader.crop(bbox_ov1, int(idx1), return_index=False)
        masks = []
        for img in (img0, img1):
            mk = None
            if maskout_val is not None:
                bad = img == maskout_val
                if bad.ndim > 2:
                    bad = np.all(bad, axis=tuple(range(2, bad.ndim)))
                if np.any(bad):
                    mk = ~binary_dilation(bad, iterations=2)
            masks.append(mk)
        jobs.append((int(idx0), int(idx1), bbox_ov0[:2] - bbox0[:2], bbox_ov1[:2] - bbox1[:2]))
        pairs.append((img0, img1, masks[0], masks[1]))
    results = [None] * len(pairs)
    try:
        results = stitching_matcher_batch(pairs, batch=kwargs.get('batch', 32), threads=kwargs.get('threads', 2), **matcher_config)
    except NotImplementedError:
        # some pair is outside the device path: take the pairs one bucket at a time so that the others still match
        from .matcher import stitching_matcher
        for k, pr in enumerate(pairs):
            try:
                results[k] = stitching_matcher(pr[0], pr[1], mask0=pr[2], mask1=pr[3], **matcher_config)
            except Exception as err:              # noqa: BLE001 -- per-pair errors are counted, stitcher.py:604-613
                err_count += 1
                first_err = first_err or err
    matches, strains, brightness_contrast = {}, {}, {}
    for (idx0, idx1, off0, off1), res in zip(jobs, results):
        if res is None or res[0] is None:
            continue
        xy0, xy1, weight, strain, phtm = res
        if index_mapper is not None:
            idx0, idx1 = index_mapper[idx0], index_mapper[idx1]
        matches[(idx0, idx1)] = (xy0 + off0, xy1 + off1, weight)
        strains[(idx0, idx1)] = strain
        if phtm is not None:
            brightness_contrast[(idx0, idx1)] = phtm
    return matches, strains, brightness_contrast, err_count > 0
