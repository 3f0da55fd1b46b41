"""Box utilities of the region (bbox) branch — drop-in for models/box_ops.py:8-56.  Boxes are <= 128 x 4 floats per
step (not kernel work), so these are plain device-side tensor expressions, fp32, no host sync."""
import torch


def box_cxcywh_to_xyxy(x):
    """(cx, cy, w, h) -> (x0, y0, x1, y1); models/box_ops.py:8-12"""
    centre, half = x[..., :2], 0.5 * x[..., 2:]
    return torch.cat([centre - half, centre + half], dim=-1)


def box_xyxy_to_cxcywh(x):
    """models/box_ops.py:15-19"""
    lo, hi = x[..., :2], x[..., 2:]
    return torch.cat([(lo + hi) / 2, hi - lo], dim=-1)


def _area(b):
    return (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])


def _iou_union(b1, b2):
    """broadcasting IoU and union of xyxy boxes"""
    wh = (torch.minimum(b1[..., 2:], b2[..., 2:]) - torch.maximum(b1[..., :2], b2[..., :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = _area(b1) + _area(b2) - inter
    return inter / union, union


def box_iou(boxes1, boxes2):
    """pairwise [N, M] IoU and union; models/box_ops.py:23-38"""
    return _iou_union(boxes1[:, None, :], boxes2[None, :, :])


def _giou(b1, b2):
    iou, union = _iou_union(b1, b2)
    wh = (torch.maximum(b1[..., 2:], b2[..., 2:]) - torch.minimum(b1[..., :2], b2[..., :2])).clamp(min=0)
    hull = wh[..., 0] * wh[..., 1]
    return iou - (hull - union) / hull


def generalized_box_iou(boxes1, boxes2):
    """pairwise [N, M] generalised IoU of xyxy boxes; models/box_ops.py:41-56"""
    return _giou(boxes1[:, None, :], boxes2[None, :, :])


def generalized_box_iou_rowwise(boxes1, boxes2):
    """diag(generalized_box_iou(b1, b2)) without the N x N matrix - the only part the bbox loss reads (xvlm.py:559)"""
    return _giou(boxes1, boxes2)
