"""CPU, world_size 2 over gloo: the image sharding + fixed-size proposal gather that runs over
RCCL on the GPUs (aznet_hip.dist).  Rank order must reproduce serial image order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aznet_hip import dist as azdist


def _fake(i, cap):
    rng = np.random.RandomState(100 + i)
    n = cap if i % 3 else max(1, cap // 2 - i)         # ragged: some images yield fewer proposals
    return rng.uniform(0, 1000, (n, 4)), rng.uniform(0, 1, n).astype(np.float32)


def _worker(rank, world, port, num_images, cap, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = azdist.shard_indices(num_images, rank, world)
        local = [_fake(i, cap) for i in mine]
        # cap=None / a cap that only some ranks exceed: the capacity is agreed collectively (no hang, no assert)
        allp = azdist.gather_proposals(local, None if num_images % 2 else cap // 2)
        ok = len(allp) == num_images
        for i, (b, s) in enumerate(allp):
            rb, rs = _fake(i, cap)
            ok = ok and np.array_equal(b, rb) and np.array_equal(s, rs)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("num_images,cap", [(8, 300), (2, 16), (7, 300), (1, 16)])      # 7, 1: ragged shards
def test_gather_two_ranks(num_images, cap):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_images, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_record_roundtrip_and_sharding():
    b, s = _fake(1, 300)
    rec = azdist.pack_record(b, s, 300)
    assert rec.shape == (azdist.record_len(300),) and rec.dtype == np.float64
    b2, s2 = azdist.unpack_record(rec, 300)
    assert np.array_equal(b, b2) and np.array_equal(s, s2)
    e = azdist.pack_record(np.zeros((0, 4)), np.zeros(0, dtype=np.float32), 300)
    assert azdist.unpack_record(e, 300)[0].shape == (0, 4)
    assert azdist.shard_indices(8, 1, 4) == [1, 5] and azdist.shard_indices(3, 2, 4) == [2]
    # single process: gather is the identity
    assert len(azdist.gather_proposals([(b, s)], 300)) == 1
    with pytest.raises(ValueError):
        azdist.pack_record(b, s, 10)


def test_device_record_layout_and_unpack():
    """The device-resident result record of az_propose (az_result_record_layout) as DeviceGather unpacks it."""
    from aznet_hip import ffi
    k = 300
    layout = ffi.AzContext.result_record_layout(k)
    nbytes, n_off, b_off, s_off = layout
    hdr = b_off                                  # the counters block at the head of the record (1 KB since round 3)
    assert hdr % 256 == 0 and nbytes == hdr + 36 * k and s_off == hdr + 32 * k and 0 < n_off < hdr and n_off % 4 == 0
    b, s = _fake(2, k)
    n = b.shape[0]
    raw = np.zeros(nbytes, dtype=np.uint8)
    raw[n_off:n_off + 4] = np.array([n], dtype=np.int32).view(np.uint8)
    raw[b_off:b_off + 32 * n] = b.reshape(-1).view(np.uint8)
    raw[s_off:s_off + 4 * n] = s.view(np.uint8)
    b2, s2 = azdist.unpack_device_record(raw, layout, k)
    assert np.array_equal(b, b2) and np.array_equal(s, s2)
    raw[n_off:n_off + 4] = np.array([-1], dtype=np.int32).view(np.uint8)
    assert azdist.unpack_device_record(raw, layout, k) is None
    with pytest.raises(ffi.AzError):
        ffi.AzContext.result_record_layout(0)


# ---- world 8 (the node size of BASELINE config 5), ragged image counts, the DEVICE-record path on CPU tensors ---------------
def _worker8(rank, world, port, num_images, k, q):
    """What tools/prop_az.py does on every rank: rows = ceil(n / world) record slots, this rank's images staged into its
    slots (here: written by hand in the device-record layout), short ranks padded, ONE all-gather, rank-interleaved
    unpack -- over gloo with CPU tensors, through the same DeviceGather object the GPUs use."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from aznet_hip import ffi
        mine = azdist.shard_indices(num_images, rank, world)
        rows = (num_images + world - 1) // world

        class NoCtx(object):          # (records are written below, not staged from a search)
            def stage_result(self, *a):
                raise AssertionError("not used")
        g = azdist.DeviceGather(NoCtx(), k, rows, "cpu")
        nbytes, n_off, b_off, s_off = g.layout
        ok = True
        for buf in (0, 1):            # both buffer pairs
            send = g.bufs[buf][0]
            send.zero_()
            for j, i in enumerate(mine):
                b, s = _fake(i, k)
                raw = np.zeros(nbytes, dtype=np.uint8)
                raw[n_off:n_off + 4] = np.array([b.shape[0]], dtype=np.int32).view(np.uint8)
                raw[b_off:b_off + 32 * b.shape[0]] = b.reshape(-1).view(np.uint8)
                raw[s_off:s_off + 4 * b.shape[0]] = s.view(np.uint8)
                send[j] = torch.from_numpy(raw)
            allp = g.gather(len(mine), buf=buf)
            ok = ok and len(allp) == num_images
            for i, (b, s) in enumerate(allp):
                rb, rs = _fake(i, k)
                ok = ok and np.array_equal(b, rb) and np.array_equal(s, rs)
        # ... and the host-record path (variable proposal counts) on the same shards
        allh = azdist.gather_proposals([_fake(i, k) for i in mine], None)
        ok = ok and len(allh) == num_images and all(np.array_equal(allh[i][0], _fake(i, k)[0]) for i in range(num_images))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_images", [8, 13, 19, 3])          # 13, 19: ragged; 3: five ranks own nothing
def test_gather_eight_ranks_in_serial_image_order(num_images):
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, num_images, 40, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]
