"""Image-level sharding of the AZ search over GPUs: one process per GPU, rank r owns the
images i with i % world == r, and proposals are exchanged with ONE fixed-size all-gather
per batch of images (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for tests).

The reference has no multi-GPU code at all (single process, caffe.set_device); what is
preserved is the result: after the gather, rank order == image order, so the gathered
list equals what test_proposals' serial loop would have produced
(lib/detect/test.py:492,508-513).  The search itself needs no communication -- images are
independent -- so there is no collective on the data path.

Record per image: [n, boxes[cap][4], scores[cap]] as float64 (n <= cap = NUM_PROPOSALS):
300 proposals -> 1501 doubles = 12 KB; latency-bound, so records are batched.
"""
import numpy as np


def record_len(cap):
    return 1 + 5 * cap


def pack_record(boxes, scores, cap):
    """boxes [n,4] f64, scores [n] f32 -> float64 [1 + 5*cap]."""
    n = boxes.shape[0]
    assert n <= cap
    rec = np.zeros(record_len(cap), dtype=np.float64)
    rec[0] = n
    rec[1:1 + 4 * n] = boxes.reshape(-1)
    rec[1 + 4 * cap:1 + 4 * cap + n] = scores
    return rec


def unpack_record(rec, cap):
    n = int(rec[0])
    boxes = rec[1:1 + 4 * n].reshape(n, 4).copy()
    scores = rec[1 + 4 * cap:1 + 4 * cap + n].astype(np.float32)
    return boxes, scores


def shard_indices(num_images, rank, world):
    """Images owned by `rank`: i % world == rank (weak scaling: fixed work per GPU)."""
    return list(range(rank, num_images, world))


def gather_proposals(local, cap, device=None, group=None):
    """local: list of (boxes, scores) for this rank's images, in local order.  Returns, on
    every rank, the list for ALL images in global image order (image i = local[i // world]
    of rank i % world).  Every rank must hold the same number of images."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [(b.copy(), s.copy()) for b, s in local]
    rl = record_len(cap)
    buf = np.stack([pack_record(b, s, cap) for b, s in local]) if local else np.zeros((0, rl))
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    # concatenated form (world * n_local rows): the layout both RCCL and gloo accept
    out = torch.empty((world * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    out = out.cpu().numpy().reshape(world, t.shape[0], t.shape[1])
    res = []
    for j in range(len(local)):
        for r in range(world):
            res.append(unpack_record(out[r, j], cap))
    return res
