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


@pytest.mark.parametrize("B", [9, 20, 32, 37])
def test_balanced_sample_order_is_a_permutation_dealt_in_serpentine(ops, B):
    """bmt_pack_rows_ordered: the order the attention kernels walk a packed batch's samples in -- a permutation of 0 .. B - 1, ranked by valid
    length (ties by index) and dealt to the eight contiguous ranges (one per XCD of the sample-major work order) in serpentine order, each range
    longest first; the ranges' squared-length sums are closer to each other than those of the batch order"""
    g = torch.Generator().manual_seed(B)
    S = 300
    L = torch.randint(S // 2, S + 1, (B,), generator=g)
    L[B // 2] = L[0]                                             # a tie
    mask = (torch.arange(S)[None, :] < L[:, None])
    pk = ops.pack_rows(mask.view(B, 1, S).to(DEV))
    assert pk.order is not None
    got = pk.order.cpu().tolist()
    assert sorted(got) == list(range(B))
    ranked = sorted(range(B), key=lambda j: (-int(L[j]), j))
    groups = [[] for _ in range(8)]
    for r, j in enumerate(ranked):
        rnd_, pos = divmod(r, 8)
        groups[7 - pos if rnd_ % 2 else pos].append(j)
    assert got == [j for grp in groups for j in grp]
    if B % 8 == 0:
        n8 = B // 8
        cost = (L.double() ** 2)
        spread = lambda o: float(max(sum(cost[j] for j in o[i * n8:(i + 1) * n8]) for i in range(8)) / (cost.sum() / 8))
        assert spread(got) <= spread(list(range(B))) + 1e-9 and spread(got) < 1.1
