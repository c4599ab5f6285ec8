"""Randomised cross-check of the search paths, under pytest so that the driver's GPU run executes it:
640 seeded random configurations (image shape, MAX_SIZE, Tz drawn from the image's own zoom scores,
BATCH_SIZE chunking, proposal count, threshold mode, DEDUP_BOXES on/off); the default path (speculative
levels 1-3, fused geometry, counting top-k) against the plain one (level by level, one launch per stage,
radix select) bit for bit, and every fourth case against the oracle's loop driven by the HIP head.
tests/stress_gpu.py holds the case generator and also runs stand-alone for longer soaks."""
import pytest

import stress_gpu

pytestmark = pytest.mark.gpu

BLOCK = 20


@pytest.fixture(scope="module")
def net():
    return stress_gpu.make_net()


@pytest.mark.parametrize("block", range(32))
def test_random_search_configurations(net, block):
    bad = []
    # (odd blocks on two lanes: the queued searches of a case -- default, pair, whole-tree over tree rows, closure -- then
    #  run four deep and overlap on the GPU)
    stress_gpu.LANES = 2 if block % 2 else 1
    net.ctx.set_lanes(stress_gpu.LANES)
    for case in range(block * BLOCK, (block + 1) * BLOCK):
        ok, desc, why = stress_gpu.run_case(net, case)
        if not ok:
            bad.append("%s | %s" % (desc, why))
    assert not bad, "\n".join(bad)
