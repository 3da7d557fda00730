"""GPU: round-6 additions outside the kernel files' own tests."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from bmt_amd import ops as o
    return o


def test_captured_scratch_belongs_to_the_step_that_captured_it(ops):
    """scratch requested by captured launches was keyed by (device, stream, capturing, name) -- a second captured step on the same
    stream shared the buffers of the first, and the first one's uncapture() freed what the second one's graph still wrote into.  Now the
    key carries the capturing step (ops.scratch_owner) and a step releases its own entries only."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eager = ops.stream_scratch("r6_probe", 1024, torch.float32, DEV)
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with ops.scratch_owner("step-a"), torch.cuda.graph(g1, stream=s):
            a = ops.stream_scratch("r6_probe", 1024, torch.float32, DEV)
            a.fill_(1.0)
        with ops.scratch_owner("step-b"), torch.cuda.graph(g2, stream=s):
            b = ops.stream_scratch("r6_probe", 1024, torch.float32, DEV)
            b.fill_(2.0)
        assert len({eager.data_ptr(), a.data_ptr(), b.data_ptr()}) == 3
        keys = [k for k in ops._SCRATCH if k[3] == "r6_probe"]
        assert sorted(str(k[2]) for k in keys) == ["None", "step-a", "step-b"]
        ops.release_scratch(owner="step-a")
        assert sorted(str(k[2]) for k in ops._SCRATCH if k[3] == "r6_probe") == ["None", "step-b"]
        g2.replay()
        s.synchronize()
        assert float(b.sum()) == 2048.0
        ops.release_scratch(owner="step-b")
        ops.release_scratch(capturing=False)
        assert not [k for k in ops._SCRATCH if k[3] == "r6_probe"]
