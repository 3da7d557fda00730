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


def test_captured_grouped_launch_reads_tables_written_at_capture_time(ops):
    """ops.gemm_bf16_grouped under a capture: the descriptor image is built on the host (bmt_gemm_bf16_grouped_image), copied into a table the
    launch owns on a stream that is not capturing, and the graph holds the product alone -- replays give what the eager launch (tables written
    by kernels in front of it) gives, also after the operands' CONTENTS changed (the addresses are what the tables hold)"""
    from tests.gpu_util import assert_close
    import math

    def rnd(*shape, seed=0):
        return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))
    shapes = [(2048, 256, 128), (700, 130, 70), (5000, 512, 1024)]
    dys = [(rnd(r, n, seed=3 + i) * 0.1).to(DEV) for i, (r, n, k) in enumerate(shapes)]
    xs = [rnd(r, k, seed=30 + i).to(DEV) for i, (r, n, k) in enumerate(shapes)]
    planes = lambda: [(ops.make_planes(dy, "bwd"), ops.make_planes(x, "bwd")) for dy, x in zip(dys, xs)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        accs = [torch.zeros(n, k, device=DEV) for (r, n, k) in shapes]
        pl = planes()
        ops.gemm_bf16_grouped([(a, b, c) for (a, b), c in zip(pl, accs)])          # eager: registers the pool a capture takes its tables from
        torch.cuda.synchronize()
        want = [a.clone() for a in accs]
        for a in accs:
            a.zero_()
        free_before = len(ops._const_tables[torch.cuda.current_device()]["free"])
        g = torch.cuda.CUDAGraph()
        with ops.scratch_owner("grouped-test"), torch.cuda.graph(g, stream=s):
            ops.gemm_bf16_grouped([(a, b, c) for (a, b), c in zip(pl, accs)])
        ops.finish_capture()
        assert len(ops._const_tables[torch.cuda.current_device()]["free"]) == free_before - 1
        g.replay()
        torch.cuda.synchronize()
        for a, w, (r, n, k) in zip(accs, want, shapes):
            assert_close(a, w, atol=2e-4 * math.sqrt(r), rtol=1e-5, name=f"captured grouped dW {n}x{k}")
        # new contents in the same buffers: the replay follows them
        for (a, b), dy, x in zip(pl, dys, xs):
            a.hi.copy_(ops.make_planes(dy * 2.0, "bwd").hi)
        for a in accs:
            a.zero_()
        g.replay()
        torch.cuda.synchronize()
        for a, w, (r, n, k) in zip(accs, want, shapes):
            assert_close(a, 2.0 * w, atol=4e-4 * math.sqrt(r), rtol=1e-5, name=f"captured grouped dW, new contents {n}x{k}")
        del g
        ops.release_const_tables("grouped-test")
        assert len(ops._const_tables[torch.cuda.current_device()]["free"]) == free_before


@pytest.mark.parametrize("B,H,S,dk,packed", [(12, 4, 300, 128, True), (3, 4, 200, 128, False), (24, 4, 800, 128, True)])
def test_attention_with_one_key_value_plane_shared_by_the_heads(ops, B, H, S, dk, packed):
    """bmt_attn_*_args.kv_shared: K = V = ONE plane of width d_k for all heads (the encoder's self-attention over an input narrower than a
    head: S_h = q'_h X^T, O'_h = P_h X) against the same kernels over that plane replicated into H column blocks: outputs, lse, dQ and the
    per-head dK / dV blocks agree to the bit (same operands, same order)"""
    g = torch.Generator().manual_seed(S + dk)
    D = H * dk
    L = torch.randint(S // 2, S + 1, (B,), generator=g)
    mask = (torch.arange(S)[None, :] < L[:, None]).view(B, 1, S)
    pk = ops.pack_rows(mask.to(DEV)) if packed else None
    n = int(L.sum()) if packed else B * S
    x = torch.randn(B * S, dk, generator=g) * 0.8
    q = torch.randn(B * S, D, generator=g) * 0.5
    do = torch.randn(B * S, D, generator=g) * 1e-2
    if not packed:
        do = (do.view(B, S, D) * mask.view(B, S, 1)).view(B * S, D)

    def f16only(t, pack):
        pl = ops.make_planes(t.to(DEV), "f16", pack=pack)
        return ops.Planes(None, None, pl.rows, pl.cols, fh=pl.fh, pack=pack)
    qP, xP = f16only(q, pk), f16only(x, pk)
    xrep = f16only(x.repeat(1, H), pk)
    m = None if packed else mask.to(DEV)
    o1, lse1 = ops.attn_fwd_planes(qP, xP, xP, B, S, S, D, m, H, precision=ops.PREC_F16, out_fmt="f16", kv_shared=True)
    o2, lse2 = ops.attn_fwd_planes(qP, xrep, xrep, B, S, S, D, m, H, precision=ops.PREC_F16, out_fmt="f16")
    torch.cuda.synchronize()
    rows = slice(0, n)
    assert torch.equal(o1.fh[rows], o2.fh[rows]) and torch.equal(o1.hi[rows], o2.hi[rows])
    for b in range(B):          # (lse keeps its padded [B, H, S] layout; positions past a sample's length are never written)
        assert torch.equal(lse1[b, :, :int(L[b])], lse2[b, :, :int(L[b])])
    dpl = ops.make_planes(do.to(DEV), "bwd", pack=pk)
    doP = ops.Planes(dpl.hi[:, :D].contiguous(), None, B * S, D, pack=pk)
    r1 = ops.attn_bwd_planes(qP, xP, xP, o1, doP, lse1, B, S, S, D, m, H, 0.0, (None, None, None), kv_shared=True)
    r2 = ops.attn_bwd_planes(qP, xrep, xrep, o2, doP, lse2, B, S, S, D, m, H, 0.0, (None, None, None))
    torch.cuda.synchronize()
    for name, (a, _), (b, _) in zip(("dq", "dk", "dv"), r1[:3], r2[:3]):
        assert torch.equal(a.hi[rows, :D], b.hi[rows, :D]), name
