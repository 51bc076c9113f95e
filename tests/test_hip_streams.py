"""engine._new_side_stream: the side streams really run BESIDE the step's stream (profiles/r05_stream_aliasing.md)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_side_streams_overlap_the_current_stream_and_each_other():
    """HIP runs two streams that share a hardware queue one after the other, and torch deals pool streams in turn — so the
    engine tests its candidates.  Whatever the pool's cursor (12 streams are taken first), the side streams it hands out must
    overlap the current stream; the first three also each other (4 hardware queues per priority by default: the step's
    stream + three); and asking again gives the same streams (found once per process)."""
    from mi355x_rec import engine as G
    if not hasattr(torch.cuda, "_sleep"):
        pytest.skip("torch.cuda._sleep is not available: the engine takes pool streams untested")
    dev = torch.device("cuda", 0)
    hold = [torch.cuda.Stream(device=dev) for _ in range(12)]
    for s in hold:
        with torch.cuda.stream(s):
            torch.zeros(1, device=dev)
    side = [G._tested_side_stream(dev, 0, k) for k in range(3)]
    cyc = G._SPIN[0]
    main = torch.cuda.current_stream(dev)
    assert len({s.cuda_stream for s in side} | {main.cuda_stream}) == 4
    for s in side:
        assert G._streams_overlap(main, s, cyc)
    for i in range(3):
        for j in range(i):
            assert G._streams_overlap(side[i], side[j], cyc)
    assert [G._tested_side_stream(dev, 0, k).cuda_stream for k in range(3)] == [s.cuda_stream for s in side]
    # the test itself tells one queue from two: a stream does not overlap itself
    assert not G._streams_overlap(side[0], side[0], cyc)
