#!/usr/bin/env python3
"""One rank of the image-sharded proposal exchange ON THE GPU (test infrastructure; started as a fresh process by
tests/test_gpu_rccl.py, one per rank, never re-exec'ed after GPU init).

    python tests/rccl_worker.py <rank> <world> <port> <n_images> <rows>

The rank joins an "nccl" (= RCCL) process group -- also when world == 1: a one-rank group still builds a
communicator and runs the collective on the GPU --, searches the images it owns (i % world == rank) with
az_propose_launch / az_propose_stage_result_dev / az_propose_fetch, exchanges the device-resident records with
DeviceGather (one all_gather_into_tensor per batch, padding rows for short ranks) and compares EVERY image of the
gathered list with a plain az_propose of that image on this rank.  The images cover: a healthy search, a search
whose fused levels overflow on first sight (err bit 8: rerun + restaging of the record, az_search.hip) and a search
whose one-pass premise fails (NaN zoom score, err bit 32: rerun through the level loop + restaging), and forced
whole-tree passes whose pruned trees miss a window (err bit 256: rerun + restaging).
Prints "RCCL_WORKER_OK <rank> <images checked>" on success."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "az-net_amd", "lib"))
sys.path.insert(0, os.path.join(HERE, ".."))


def image_case(i, synth):
    """(H, W, scale, Tz, map seed, head key, static_tree) of image i: shapes / thresholds vary; every third image
    uses the head with a NaN zoom bias (one-pass premise fails -> err bit 32 -> level-loop rerun, restaged); the
    800x1200 images walk their 2729-region tree level by level (static_tree=False), whose sixth level outgrows the
    fused level kernel's LDS tables on first sight (err bit 8 -> multi-launch rerun, restaged)."""
    shapes = [(600, 1000, 1.0), (375, 500, 1.6), (480, 640, 1.25), (800, 1200, 0.75)]
    H, W, scale = shapes[i % len(shapes)]
    Tz = [0.0, 0.45, 0.0, 0.0][i % 4]
    return H, W, scale, Tz, 100 + i, ("nan" if i % 3 == 2 else "ok"), (i % 4 != 3)


def params_of(i, ffi, synth, k):
    H, W, scale, Tz, seed, hk, static = image_case(i, synth)
    # every fifth image forces the whole-tree pass: with Tz = 0.45 the pruned tree needs windows that pass lacks
    # (err bit 256 -> rerun level by level, record restaged)
    return ffi.AzContext.make_params(H, W, scale, Tz, num_proposals=k, static_tree=static,
                                     full_spec=(True if i % 5 == 1 else None))


def main():
    rank, world, port, n_images, rows = (int(x) for x in sys.argv[1:6])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from aznet_hip import ffi, synth
    from aznet_hip import dist as azdist
    from aznet_hip.net import HipAZNet

    ndev = torch.cuda.device_count()
    local = rank % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        heads = {"ok": synth.make_head(seed=77, **synth.SMALL_DIMS), "nan": synth.make_head(seed=78, **synth.SMALL_DIMS)}
        heads["nan"]["bz"] = np.full(1, np.nan, dtype=np.float32)
        nets = {k: HipAZNet(h, device=local, name="rccl_" + k) for k, h in heads.items()}
        # (the expected results come from contexts of their own: the exchanging contexts must meet every image shape
        #  for the first time inside launch / stage / fetch, where a fallback rerun has to restage the record)
        refs = {k: HipAZNet(h, device=local, name="ref_" + k) for k, h in heads.items()}
        k = 300
        # (AZ_TEST_NATIVE_GATHER=1: the library's own ncclAllGather on the ctx stream; AZ_TEST_LANES=2: two lanes)
        native = os.environ.get("AZ_TEST_NATIVE_GATHER", "0") == "1"
        for n in nets.values():
            n.ctx.set_lanes(int(os.environ.get("AZ_TEST_LANES", "1")))
        gats = {key: azdist.DeviceGather(n.ctx, k, rows, dev, always_collective=True, native=native) for key, n in nets.items()}
        assert all(g.collective for g in gats.values())

        def fmap_of(i):
            H, W, scale, Tz, seed, hk, _ = image_case(i, synth)
            fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
            return synth.make_feature_map(seed, synth.SMALL_DIMS["C"], fh, fw)

        # every rank computes the expected result of EVERY image with a plain az_propose (same kernels, same bits)
        want = []
        for i in range(n_images):
            hk = image_case(i, synth)[5]
            refs[hk].set_conv(fmap_of(i))
            want.append(refs[hk].propose(params_of(i, ffi, synth, k), want_scores=True))
        # the exchange runs per head (a record buffer belongs to one context); images keep their global order
        checked = 0
        for hk, net in nets.items():
            ids = [i for i in range(n_images) if image_case(i, synth)[5] == hk]
            mine = ids[rank::world]
            n_batches = (max(len(ids[r::world]) for r in range(world)) + rows - 1) // rows
            got = []
            for b in range(n_batches):
                batch = mine[b * rows:(b + 1) * rows]
                for j, i in enumerate(batch):
                    net.set_conv(fmap_of(i))
                    net.ctx.propose_launch(params_of(i, ffi, synth, k))
                    gats[hk].stage(j, buf=b % 2)
                    net.ctx.propose_fetch()
                # all ranks' rows of this batch, rank-interleaved; odd batches through the non-blocking form
                # (side stream, double-buffered send buffer, pinned host copy)
                if b % 2:
                    res = gats[hk].gather_end(gats[hk].gather_begin(len(batch), buf=b % 2))
                else:
                    res = gats[hk].gather(len(batch), buf=b % 2)
                got.append(res)
            # global order of this head's images: batch b holds ids[b*rows*world : ...] interleaved by rank
            flat = [x for res in got for x in res]
            order = []
            for b in range(n_batches):
                for j in range(rows):
                    for r in range(world):
                        sub = ids[r::world][b * rows:(b + 1) * rows]
                        if j < len(sub):
                            order.append(sub[j])
            assert len(flat) == len(order) == len(ids), (len(flat), len(order), len(ids))
            for (boxes, scores), i in zip(flat, order):
                wb, ws = want[i]
                assert boxes.shape == wb.shape and np.array_equal(boxes, wb, equal_nan=True), "image %d boxes" % i
                assert np.array_equal(scores, ws, equal_nan=True), "image %d scores" % i
                checked += 1
        assert checked == n_images
        dist.barrier()
        print("RCCL_WORKER_OK %d %d" % (rank, checked))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
