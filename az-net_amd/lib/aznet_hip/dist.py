"""Image-level sharding of the AZ search over GPUs: one process per GPU, rank r owns the
images i with i % world == r, and proposals are exchanged with ONE fixed-size all-gather
per batch of images (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for tests).

The reference has no multi-GPU code at all (single process, caffe.set_device); what is
preserved is the result: after the gather, rank order == image order, so the gathered
list equals what test_proposals' serial loop would have produced
(lib/detect/test.py:492,508-513).  The search itself needs no communication -- images are
independent -- so there is no collective on the data path.

Two record formats, both one row per image:
  * device records (`DeviceGather`, the production path): the result block az_propose leaves in
    HBM (az_result_record_layout: int32 n, boxes f64[k][4], scores f32[k]; 11.3 KB for k = 300)
    is copied device-to-device into the RCCL send buffer on the ctx stream
    (az_propose_stage_result_dev) -- the exchanged bytes never visit the host before the gather;
  * host records (`gather_proposals`): [n, boxes[cap][4], scores[cap]] as float64, for callers
    that hold NumPy results (variable proposal counts: cfg.SEAR.FIXED_PROPOSAL_NUM = False,
    cfg.SEAR.APPEND_BOXES).  `cap` is agreed collectively (all-reduce MAX), so a rank with more
    boxes than another cannot make the collective hang.
Ranks may own different numbers of images (num_images % world != 0): short ranks send padding
rows (n = -1) that are dropped after the gather.
"""
import numpy as np


def record_len(cap):
    return 1 + 5 * cap


def pack_record(boxes, scores, cap):
    """boxes [n,4] f64, scores [n] f32 -> float64 [1 + 5*cap]."""
    n = boxes.shape[0]
    if n > cap:
        raise ValueError("pack_record: %d boxes do not fit a record of capacity %d" % (n, cap))
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


def _world(group=None):
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _agree_max(values, device=None, group=None):
    """Element-wise MAX of a small int vector over the ranks (one tiny all-reduce)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return [int(v) for v in t.cpu().tolist()]


def _interleave(rows_by_rank, n_rows, keep):
    """rows_by_rank[r][j] = image j * world + r  ->  global image order, padding dropped."""
    res = []
    for j in range(n_rows):
        for r in range(len(rows_by_rank)):
            item = keep(rows_by_rank[r][j])
            if item is not None:
                res.append(item)
    return res


def gather_proposals(local, cap=None, device=None, group=None):
    """local: list of (boxes [n,4] f64, scores [n] f32) for this rank's images, in local order.
    Returns, on every rank, the list for ALL images in global image order (image i =
    local[i // world] of rank i % world).  cap=None (or too small on any rank) is replaced by the
    largest box count over all ranks; ranks may hold different numbers of images."""
    import torch
    import torch.distributed as dist
    world = _world(group)
    if world == 1:
        return [(b.copy(), s.copy()) for b, s in local]
    n_max = max([b.shape[0] for b, _ in local] + [0])
    cap_all, rows = _agree_max([max(n_max, cap or 0, 1), len(local)], device=device, group=group)
    rl = record_len(cap_all)
    buf = np.zeros((rows, rl), dtype=np.float64)
    buf[:, 0] = -1.0                                       # padding rows of a short rank
    for j, (b, s) in enumerate(local):
        buf[j] = pack_record(b, s, cap_all)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    # concatenated form (world * rows rows): the layout both RCCL and gloo accept
    out = torch.empty((world * rows, rl), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    out = out.cpu().numpy().reshape(world, rows, rl)
    return _interleave(out, rows, lambda rec: None if rec[0] < 0 else unpack_record(rec, cap_all))


def unpack_device_record(raw, layout, k):
    """raw: uint8 [bytes] of one az_propose result record -> (boxes [n,4] f64, scores [n] f32),
    or None for a padding row (n < 0)."""
    nbytes, n_off, b_off, s_off = layout
    n = int(raw[n_off:n_off + 4].view(np.int32)[0])
    if n < 0:
        return None
    n = min(n, k)
    boxes = raw[b_off:b_off + 32 * k].view(np.float64).reshape(k, 4)[:n].copy()
    scores = raw[s_off:s_off + 4 * k].view(np.float32)[:n].copy()
    return boxes, scores


class DeviceGather(object):
    """Send buffers of `rows` result records in HBM + the all-gather over them.

        g = DeviceGather(ctx, num_proposals, rows, device)
        for j in range(n_local):
            ctx.propose_launch(params); g.stage(j); ctx.propose_fetch()
        everything = g.gather(n_local)       # all ranks' images, global image order

    stage(j) enqueues a device-to-device copy of the search's result record into row j on the ctx
    stream (complete when that search's propose_fetch returns); gather() marks the rows past n_local as padding,
    runs ONE all_gather_into_tensor on the device buffers and unpacks on the host.
    There are TWO buffer pairs (`buf` = 0 / 1): a caller that launches the first search of the next batch before it
    gathers the current one (propose_launch queues up to two searches) stages that search into the other pair."""

    def __init__(self, ctx, num_proposals, rows, device, group=None, always_collective=None, native=False):
        """native=True: the exchange is ONE ncclAllGather issued by the library itself on the ctx stream
        (az_gather_records; the communicator is made here -- rank 0's id reaches the others through torch.distributed,
        the control plane -- and bound to the RCCL this process already holds); False: torch.distributed's
        all_gather_into_tensor on torch's stream."""
        import torch
        import torch.distributed as dist
        from aznet_hip import ffi
        self.ctx, self.k, self.rows, self.group = ctx, int(num_proposals), int(rows), group
        self.native = bool(native)
        # a process group of ONE rank still runs the collective (RCCL on the one GPU) unless told otherwise: the
        # single-GPU run then exercises the code path the 8-GPU run takes
        self.collective = dist.is_initialized() if always_collective is None else bool(always_collective)
        # a "gloo" process group (several ranks sharing one GPU, a box without RCCL: the control plane only) cannot gather
        # device tensors: the staged records still land in the HBM send buffer device-to-device, one copy brings the
        # batch to the host and the all-gather runs on host tensors
        self.host_collective = (dist.is_initialized() and dist.get_backend(group) == "gloo"
                                and torch.device(device).type == "cuda")
        self.layout = ffi.AzContext.result_record_layout(self.k)
        self.rec_bytes = self.layout[0]
        self.device = device
        self.world = _world(group)
        self.bufs = [(torch.zeros((self.rows, self.rec_bytes), dtype=torch.uint8, device=device),
                      torch.empty((self.world * self.rows, self.rec_bytes), dtype=torch.uint8, device=device))
                     for _ in range(2)]
        self.send, self.recv = self.bufs[0]
        pad = np.zeros(self.rec_bytes, dtype=np.uint8)
        pad[self.layout[1]:self.layout[1] + 4] = np.array([-1], dtype=np.int32).view(np.uint8)
        self._pad = torch.from_numpy(pad).to(device)
        # the non-blocking form's side stream / pinned host buffers, and which buffer pair has an exchange in flight
        self._side = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        self._host = None
        self._busy = [False, False]
        if self.native:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
            uid = [ffi.AzContext.rccl_unique_id() if rank == 0 else None]
            if dist.is_initialized() and self.world > 1:
                dist.broadcast_object_list(uid, src=0, group=group)
            ctx.rccl_init(uid[0], self.world, rank)
            self.collective = True

    def stage(self, j, buf=0):
        assert 0 <= j < self.rows
        if self._busy[buf]:
            raise RuntimeError("DeviceGather.stage: buffer pair %d still has an exchange in flight (gather_end it first, "
                               "or stage into the other pair)" % buf)
        self.ctx.stage_result(self.bufs[buf][0].data_ptr() + j * self.rec_bytes, self.rec_bytes)

    def stage_batch(self, j0, n, buf=0):
        """Right behind ctx.batch_launch of n images: their records into rows j0 .. j0 + n - 1 (one strided copy)."""
        assert 0 <= j0 and j0 + n <= self.rows
        if self._busy[buf]:
            raise RuntimeError("DeviceGather.stage_batch: buffer pair %d still has an exchange in flight" % buf)
        self.ctx.batch_stage_results(self.bufs[buf][0].data_ptr() + j0 * self.rec_bytes, self.rec_bytes, n * self.rec_bytes)

    def _exchange(self, n_local, buf):
        import torch.distributed as dist
        send, recv = self.bufs[buf]
        if n_local < self.rows:
            send[n_local:] = self._pad
        if self.native:
            # the pad write (torch's current stream) before the collective (the context's collective stream), the collective
            # before whoever reads `recv` on torch's current stream: two event waits on the device, nothing on the host
            import torch
            cur = torch.cuda.current_stream(send.device)
            ms = self.ctx.comm_stream()
            ms.wait_stream(cur)
            self.ctx.gather_records(send.data_ptr(), recv.data_ptr(), self.rows * self.rec_bytes)
            cur.wait_stream(ms)
            return recv, self.world
        if self.host_collective:
            import torch
            hs = send.cpu()                                   # (synchronises torch's current stream: the pad write is in)
            hr = torch.empty((self.world * self.rows, self.rec_bytes), dtype=torch.uint8)
            dist.all_gather_into_tensor(hr, hs, group=self.group)
            return hr, self.world
        if self.world > 1 or self.collective:
            # stage() copies ran on the ctx stream and are complete: az_propose_stage_result_dev re-records the event
            # propose_fetch waits for BEHIND the staging copy, and the caller has fetched every search of the batch.  The
            # pad write above and the collective are ordered by torch's current stream, which all_gather_into_tensor
            # joins with RCCL's stream on both sides (async_op=False)
            dist.all_gather_into_tensor(recv, send, group=self.group)
            return recv, self.world
        return send, 1

    def gather(self, n_local, to_host=True, buf=0):
        """After the propose_fetch of the batch's last search."""
        import torch
        got, w = self._exchange(n_local, buf)
        if not to_host:
            if got.is_cuda:
                torch.cuda.current_stream(got.device).synchronize()
            return None
        raw = got.cpu().numpy().reshape(w, self.rows, self.rec_bytes)
        return _interleave(raw, self.rows, lambda rec: unpack_device_record(rec, self.layout, self.k))

    # ---- the same exchange without stalling the host ----------------------------------------------------------
    def gather_begin(self, n_local, buf=0):
        """Start the exchange of the batch staged in pair `buf` and return a handle; `gather_end(handle)` collects the
        result later (stage the next batch into the other pair meanwhile).  The collective and the device-to-host copy
        run on a side stream; nothing here blocks the host."""
        import torch
        if self._busy[buf]:
            raise RuntimeError("DeviceGather.gather_begin: buffer pair %d already has an exchange in flight" % buf)
        if self._host is None:
            self._host = [torch.empty(self.bufs[0][1].shape, dtype=torch.uint8).pin_memory() for _ in range(2)]
        host = self._host[buf]
        self._busy[buf] = True
        ev = torch.cuda.Event()
        self._side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._side):
            src, w = self._exchange(n_local, buf)
            host[:src.shape[0]].copy_(src, non_blocking=True)
            ev.record(self._side)
        return (ev, host, w, buf)

    def gather_end(self, handle):
        ev, host, w, buf = handle
        ev.synchronize()
        self._busy[buf] = False
        raw = host.numpy()[:w * self.rows].reshape(w, self.rows, self.rec_bytes)
        return _interleave(raw, self.rows, lambda rec: unpack_device_record(rec, self.layout, self.k))
