"""GPU: every C-ABI entry point against the matching oracle function on the same seeded inputs.
Tolerances are stated per test: discrete outputs must be identical; fp32 contractions are compared with an
fp64 evaluation of the same formula at a few ulp of the accumulated magnitude."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import ogmm_oracle as O
from ogmm_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def raw_ops():
    assert torch.cuda.is_available()
    from ogmm_amd import ops as _ops
    return _ops


def dev(t):
    return t.cuda().contiguous()


def clouds(C, N, seed=0, kind="partial"):
    src, tgt, _, _ = synth.make_batch(seed, (C + 1) // 2, N, kind)
    return torch.cat([src, tgt], 0)[:C].transpose(1, 2).contiguous()      # [C,N,3]


# ------------------------------------------------------------------------------------------------ K1
class _KnnBoth:
    """ops.knn through BOTH forms: the plain kernels (ogmm_knn) and, where it applies, the fused head kernel (ogmm_knn_pos_head, round 5: scan B over scan A's
    marks; the 5-NN graph and the positional front end folded in).  Index tensors must be IDENTICAL (same distance bits, same insertion order, same tie
    resolution), the 5-NN graph must be knn(xyz, 5)'s and the hidden maps pos_hidden's, bit for bit -- so every kNN test below pins both forms."""

    def __init__(self, ops):
        self.ops = ops

    def knn(self, xyz_dev, k):
        ops = self.ops
        base = ops.knn(xyz_dev, k)
        C, N, _ = xyz_dev.shape
        if ops.knn_pos_head_supported(N, k):
            got = ops.knn_pos_head(xyz_dev, k)
            assert torch.equal(got, base), "the fused head kernel's %d-NN graph differs from ogmm_knn's in %d rows" % (k, int((got != base).any(-1).sum()))
            g = torch.Generator().manual_seed(N * 31 + k)
            pos = {key: (torch.rand(64, generator=g) * 2 - 0.5).cuda() for key in ("w_dis", "s_dis", "t_dis", "w_ang", "s_ang", "t_ang")}
            got, idx5, hd, ha = ops.knn_pos_head(xyz_dev, k, pos)
            ref5 = ops.knn(xyz_dev, 5)
            assert torch.equal(got, base) and torch.equal(idx5, ref5), "5-NN graph differs in %d rows" % int((idx5 != ref5).any(-1).sum())
            hd_r, ha_r = ops.pos_hidden(xyz_dev, ref5, 5, pos)
            assert torch.equal(hd, hd_r) and torch.equal(ha, ha_r), "positional hidden maps differ: %.3e / %.3e" % ((hd - hd_r).abs().max().item(), (ha - ha_r).abs().max().item())
            chw = xyz_dev.transpose(1, 2).contiguous()                     # [C,3,N]: the model's input layout
            xyz2 = ops.pack_clouds(chw, chw)
            assert torch.equal(xyz2[:C], xyz_dev) and torch.equal(xyz2[C:], xyz_dev)
        return base


@pytest.fixture(scope="module")
def ops(raw_ops):          # every test sees the operator module; its knn() checks both kernels
    class W:
        def __getattr__(self, name):
            return getattr(raw_ops, name)
    w = W()
    w.knn = _KnnBoth(raw_ops).knn
    return w


@pytest.mark.parametrize("N,fused", [(2304, True), (2305, False)])
def test_knn_head_boundary(ops, raw_ops, N, fused):
    """the largest cloud the fused head kernel takes (dynamic + static LDS = 64 KiB at N = 2304, k = 20) gives ogmm_knn's graph (the
    wrapper's knn() compares both forms bit for bit), and one point more reports "unsupported", so that the forward takes the three-kernel head."""
    assert bool(raw_ops.knn_pos_head_supported(N, 20)) == fused
    xyz = clouds(2, N, seed=N).cuda()
    idx = ops.knn(xyz, 20).cpu().long()
    ref = torch.topk(O.sq_dist_expanded(xyz.cpu(), xyz.cpu()), 20, dim=-1, largest=False, sorted=True)
    d = O.sq_dist_expanded(xyz.cpu(), xyz.cpu())
    assert torch.equal(torch.gather(d, 2, idx), ref[0])          # the same distance multiset per row, ascending: ties may swap indices only


@pytest.mark.parametrize("C,N,k", [(4, 1024, 20), (3, 200, 12), (2, 2048, 5), (2, 717, 20), (1, 64, 32), (2, 33, 1), (2, 2048, 20), (3, 1000, 8)])
def test_knn_identical_indices(ops, C, N, k):
    """Distance rows are bit-identical to the reference's, and the kept neighbour SET equals torch.topk's even when
    rank k is an exact tie (the expanded formula quantises distances to ~3e-8, so ties are not rare).  Only the order
    among equal distances inside the set may differ (torch: unspecified; ours: by index)."""
    xyz = clouds(C, N, seed=11)
    ref = O.knn_indices(xyz, k)
    got = ops.knn(dev(xyz), k).cpu().long()
    d = O.sq_dist_expanded(xyz, xyz)
    assert torch.equal(torch.gather(d, 2, got), torch.gather(d, 2, ref)), "distance rows differ"
    assert torch.equal(got.sort(-1)[0], ref.sort(-1)[0]), "kept sets differ in %d rows" % int((got.sort(-1)[0] != ref.sort(-1)[0]).any(-1).sum())


@pytest.mark.parametrize("kind,N,k", [("room", 2048, 20), ("room", 2048, 5), ("room", 1024, 20), ("room", 700, 12), ("room", 1200, 20), ("room", 2000, 32), ("room", 100, 20)])
def test_knn_boundary_ties_resolved_like_torch(ops, kind, N, k):
    """Axis-aligned room planes produce many exact ties at rank k; both torch code paths (heap select for k*64 <= N,
    introselect otherwise) must be reproduced."""
    xyz = clouds(4, N, seed=400, kind=kind)
    d = O.sq_dist_expanded(xyz, xyz)
    dk = d.topk(k + 1, dim=-1, largest=False)[0]
    n_tied = int((dk[:, :, k - 1] == dk[:, :, k]).sum())
    ref = O.knn_indices(xyz, k)
    got = ops.knn(dev(xyz), k).cpu().long()
    assert torch.equal(got.sort(-1)[0], ref.sort(-1)[0]), "sets differ (rows with a rank-k tie: %d)" % n_tied
    print("rows with an exact tie at rank k:", n_tied)


def test_knn_duplicate_points_ties(ops):
    """Exact duplicates tie at the clamped distance 1e-12; the neighbour SET must agree in distance terms."""
    xyz = clouds(1, 256, seed=5)
    xyz[0, 100:110] = xyz[0, 0:10]
    d = O.sq_dist_expanded(xyz, xyz)
    got = ops.knn(dev(xyz), 8).cpu().long()
    ref = O.knn_indices(xyz, 8)
    assert torch.equal(torch.gather(d, 2, got), torch.gather(d, 2, ref))      # same sorted distance values
    assert torch.equal(got.sort(-1)[0], ref.sort(-1)[0])


@pytest.mark.parametrize("side,k,shuffle", [(32, 18, False), (32, 18, True), (30, 15, True), (24, 10, True)])
def test_knn_lattice_ties_introselect_branch(ops, side, k, shuffle):
    """A square lattice (integer coordinates: every distance exact) puts whole shells of equidistant neighbours across rank k in almost
    every row, with N < 64 k: torch.topk's std::nth_element branch, which the tie-resolution kernel executes with workgroup-wide
    partitions.  The kept SET must be torch's, row by row (a different member of a tied shell is a different EdgeConv graph)."""
    g = torch.Generator().manual_seed(side * 100 + k)
    ii, jj = torch.meshgrid(torch.arange(side), torch.arange(side), indexing="ij")
    pts = torch.stack([ii.reshape(-1).float(), jj.reshape(-1).float(), torch.zeros(side * side)], dim=1)
    clouds_ = []
    for c in range(2):
        p = pts[torch.randperm(side * side, generator=g)] if shuffle else pts
        clouds_.append(p * (0.25 if c else 1.0))                 # exact in fp32 either way
    xyz = torch.stack(clouds_)
    N = side * side
    assert k * 64 > N
    d = O.sq_dist_expanded(xyz, xyz)
    dk = d.topk(k + 1, dim=-1, largest=False)[0]
    n_tied = int((dk[:, :, k - 1] == dk[:, :, k]).sum())
    assert n_tied > N                                             # most rows tie at rank k
    ref = O.knn_indices(xyz, k)
    got = ops.knn(dev(xyz), k).cpu().long()
    assert torch.equal(torch.gather(d, 2, got), torch.gather(d, 2, ref)), "distance rows differ"
    differ = (got.sort(-1)[0] != ref.sort(-1)[0]).any(-1)
    assert not bool(differ.any()), "kept sets differ in %d of %d rows (%d rows tie at rank k)" % (int(differ.sum()), 2 * N, n_tied)


# ------------------------------------------------------------------------------------------------ K5 / K6
@pytest.mark.parametrize("C,N,npoint", [(4, 1024, 128), (2, 717, 128), (3, 200, 32), (2, 2048, 64), (1, 4096, 16)])
def test_fps_identical_chains(ops, C, N, npoint):
    xyz = clouds(C, N, seed=21)
    g = torch.Generator().manual_seed(1)
    start = torch.randint(0, N, (3, C), generator=g)
    got = ops.fps(dev(xyz), npoint, dev(start.int())).cpu().long()
    for s in range(3):
        assert torch.equal(got[s], O.fps(xyz, npoint, start[s])), "random-start chain %d differs" % s
    got_c = ops.fps(dev(xyz), npoint, None).cpu().long()
    ref_c = O.fps(xyz, npoint, None)
    assert torch.equal(got_c, ref_c), "centre-start chain differs in %d clouds" % int((got_c != ref_c).any(-1).sum())


def test_gather_rows_with_cloud_swap(ops):
    C, N, D, S = 4, 50, 64, 7
    feats = torch.randn(C * N, D)
    ids = torch.randint(0, N, (C, S))
    swap = torch.tensor([2, 3, 0, 1])
    got = ops.gather_rows(dev(feats), D, C, N, D, dev(ids.int()), dev(swap.int())).cpu()
    ref = O.gather_rows(feats.view(C, N, D), ids)[swap]
    assert torch.equal(got, ref)


# ------------------------------------------------------------------------------------------------ GEMM engine
def _gemm_ref(A, B):
    return (A.double() @ B.double().t())


ENGINES = ["f32", "f16x3_frag"]          # (the row-major-planes engine "f16x3" is no model path: it moved to the tools-only libogmm_probe.so)


def _split(ops, B, engine):
    if engine == "f32":
        return None
    return ops.split_f16(dev(B), frag=engine == "f16x3_frag")


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("M,N,K", [(300, 70, 36), (128, 128, 32), (1000, 512, 512), (129, 65, 4), (2048, 1024, 1024), (257, 130, 516)])
def test_gemm_plain(ops, M, N, K, engine):
    torch.manual_seed(M + N + K)
    A, B = torch.randn(M, K), torch.randn(N, K) * 0.05
    out = torch.empty(M, N, device="cuda")
    ops.gemm_nt(dev(A), K, K, dev(B), K, M, N, C=out, ldc=N, split=_split(ops, B, engine))
    ref = _gemm_ref(A, B)
    # v_mfma_f32_32x32x2_f32 is a k-ordered fp32 fma chain: |err| <~ 1.5e-7 * sum_k |a_k b_k| (guide, section 3);
    # the fp16x3 engine adds <= 3 * 2^-22 per product on top of its own fp32 accumulation
    bound = 6e-7 * (A.double().abs() @ B.double().abs().t()) + 1e-7
    excess = ((out.cpu().double() - ref).abs() / bound).max().item()
    assert excess < 1.0, excess


@pytest.mark.parametrize("engine", ENGINES)
def test_gemm_detects_transposition(ops, engine):
    """A = identity, asymmetric B (guide rule: symmetric inputs hide a swapped C write)."""
    n = 96
    A = torch.eye(n)
    B = torch.arange(n * n, dtype=torch.float32).view(n, n) / 8.0
    out = torch.empty(n, n, device="cuda")
    ops.gemm_nt(dev(A), n, n, dev(B), n, n, n, C=out, ldc=n, split=_split(ops, B, engine))
    assert torch.equal(out.cpu(), B.t().contiguous())      # exact in both engines: every entry fits 22 bits


def test_gemm_f16x3_accuracy_is_fp32_class_and_flags_overflow(ops):
    torch.manual_seed(9)
    M, N, K = 512, 256, 1024
    A, B = torch.randn(M, K) * 3, torch.randn(N, K) * 0.03
    ref = _gemm_ref(A, B)
    o32, o16 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.gemm_nt(dev(A), K, K, dev(B), K, M, N, C=o32, ldc=N)
    ops.gemm_nt(dev(A), K, K, None, K, M, N, C=o16, ldc=N, split=ops.split_f16(dev(B), frag=True), overflow=flag)
    e32 = (o32.cpu().double() - ref).abs().max().item()
    e16 = (o16.cpu().double() - ref).abs().max().item()
    print("max abs error: exact-fp32 engine %.3e, fp16x3 engine %.3e (|C| max %.2f)" % (e32, e16, ref.abs().max().item()))
    assert e16 < 3 * e32 + 1e-6
    assert int(flag.item()) == 0
    A[3, 5] = 1e5
    ops.gemm_nt(dev(A), K, K, None, K, M, N, C=o16, ldc=N, split=ops.split_f16(dev(B), frag=True), overflow=flag)
    assert int(flag.item()) == 1
    with pytest.raises(Exception, match="libogmm_probe"):          # row-major split planes: not a product engine any more
        ops.gemm_nt(dev(A), K, K, None, K, M, N, C=o16, ldc=N, split=ops.split_f16(dev(B)), overflow=flag)


@pytest.mark.parametrize("engine", ENGINES)
def test_gemm_two_pieces_epilogue_residual(ops, engine):
    torch.manual_seed(0)
    M, N, K1, K2 = 333, 200, 64, 4
    A1, A2, B = torch.randn(M, K1), torch.randn(M, K2), torch.randn(N, K1 + K2)
    scale, shift, res = torch.rand(N) + 0.5, torch.randn(N), torch.randn(M, N)
    for act, fn in ((ops.ACT_RELU, torch.relu), (ops.ACT_LEAKY02, lambda v: torch.nn.functional.leaky_relu(v, 0.2)),
                    (ops.ACT_SIGMOID, torch.sigmoid), (ops.ACT_NONE, lambda v: v)):
        out = torch.empty(M, N, device="cuda")
        ops.gemm_nt(dev(A1), K1, K1, dev(B), K1 + K2, M, N, C=out, ldc=N, A2=dev(A2), lda2=K2, K2=K2, scale=dev(scale), shift=dev(shift),
                    alpha=0.5, act=act, res=dev(res), ldr=N, split=_split(ops, B, engine))
        ref = fn(0.5 * _gemm_ref(torch.cat([A1, A2], 1), B) * scale.double() + shift.double()) + res.double()
        assert (out.cpu().double() - ref).abs().max().item() < 2e-5


# ---- the LDS-DMA engine (gemm_f16x3_v8.hip) only takes shapes of >= 256 tiles of 256 x 256: these run it on purpose
LARGE = [  # M, N, K1, K2, residual, act, what
    (65536, 256, 64, 0, False, "none", "one K step pair, exact tiles"),
    (65536, 256, 32, 0, False, "relu", "a single K step"),
    (65536 + 77, 512, 128, 32, True, "leaky", "ragged last row tile, second A piece, residual"),
    (32768, 768, 128, 0, True, "sigmoid", "three column tiles, sigmoid"),
    (33000, 600, 64, 64, False, "relu", "ragged rows AND ragged columns (per-element epilogue), two pieces"),
]


@pytest.mark.parametrize("M,N,K1,K2,has_res,act,what", LARGE)
def test_gemm_lds_dma_engine(ops, M, N, K1, K2, has_res, act, what):
    torch.manual_seed(M + N + K1)
    A1 = torch.relu(torch.randn(M, K1, device="cuda")) * (torch.rand(M, 1, device="cuda") * 3)
    A2 = torch.randn(M, K2, device="cuda") if K2 else None
    W = torch.randn(N, K1 + K2, device="cuda") * 0.05
    scale, shift = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda") if has_res else None
    out = torch.full((M, N), float("nan"), device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    code = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY02, "sigmoid": ops.ACT_SIGMOID}[act]
    ops.gemm_nt(A1, K1, K1, None, K1 + K2, M, N, C=out, ldc=N, A2=A2, lda2=K2, K2=K2, scale=scale, shift=shift, act=code, res=res, ldr=N if has_res else 0,
                split=ops.split_f16(W, frag=True, k1=(K1 if K2 else None)), overflow=flag)
    rows = torch.cat([torch.arange(0, 300), torch.randint(0, M, (600,)), torch.arange(M - 300, M)]).cuda()
    Af = (A1[rows] if A2 is None else torch.cat([A1[rows], A2[rows]], 1)).double()
    acc = Af @ W.double().t()
    pre = acc * scale.double() + shift.double()
    ref = {"none": lambda v: v, "relu": torch.relu, "leaky": lambda v: torch.where(v > 0, v, 0.2 * v), "sigmoid": torch.sigmoid}[act](pre)
    if has_res:
        ref = ref + res[rows].double()
    bound = 6e-7 * ((Af.abs() @ W.double().abs().t()) * scale.double() + shift.double().abs() + (res[rows].double().abs() if has_res else 0)) + 1e-7
    got = out[rows].double()
    assert not torch.isnan(out).any(), what
    assert ((got - ref).abs() / bound).max().item() < 1.0, what
    assert int(flag.item()) == 0
    # the same shape on the register-staged engines must agree to the last bit (same products in the same order per accumulator)
    out2 = torch.empty_like(out)
    sp = ops.split_f16(W, frag=True, k1=(K1 if K2 else None)); sp["variant"] = 21            # 128 x 128 tiles (gemm_f16x3_v2.hip)
    ops.gemm_nt(A1, K1, K1, None, K1 + K2, M, N, C=out2, ldc=N, A2=A2, lda2=K2, K2=K2, scale=scale, shift=shift, act=code, res=res, ldr=N if has_res else 0, split=sp)
    assert torch.equal(out, out2), what


def test_gemm_lds_dma_engine_statistics_a_transform_batch_overflow(ops):
    """Column statistics in the epilogue (producer of an InstanceNorm), the normalisation applied to A (its consumer), outer batching with a
    per-batch weight image, and the binary16 overflow flag -- all on shapes the LDS-DMA engine takes."""
    torch.manual_seed(5)
    G, rows, N, K = 64, 1024, 256, 128
    M = G * rows
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    st = torch.zeros((G, N, 2), dtype=torch.float64, device="cuda")
    z = torch.empty(M, N, device="cuda")
    ops.gemm_nt(A, K, K, None, K, M, N, C=z, ldc=N, shift=b, split=ops.split_f16(W, frag=True), col_stats=st, group_rows=rows)
    zg = z.view(G, rows, N).double()
    assert (st[:, :, 0] - zg.sum(1)).abs().max().item() < 1e-6 * rows and ((st[:, :, 1] - (zg * zg).sum(1)).abs() / (zg * zg).sum(1)).max().item() < 1e-6
    # the same statistics dealt over 8 copies of the table (struct ogmm_gemm.col_stats_slot_mask), on every engine that has the epilogue
    for n_cols, variant in ((256, None), (512, None), (128, None), (64, None)):
        Wn = torch.randn(n_cols, K, device="cuda") * 0.05
        st1 = torch.zeros((G // 32, n_cols, 2), dtype=torch.float64, device="cuda")
        st8 = torch.zeros((8, G // 32, n_cols, 2), dtype=torch.float64, device="cuda")
        zn_ = torch.empty(M, n_cols, device="cuda")
        for st_ in (st1, st8):
            ops.gemm_nt(A, K, K, None, K, M, n_cols, C=zn_, ldc=n_cols, split=ops.split_f16(Wn, frag=True), col_stats=st_, group_rows=32 * rows)
        assert float(st8.abs().min(dim=0).values.max()) > 0                    # every copy took part
        assert ((st8.sum(0) - st1).abs() / st1.abs().clamp_min(1.0)).max().item() < 1e-9
    sc, sh = ops.instnorm_finalize(st, rows, 1e-5)
    W2 = torch.randn(N, N, device="cuda") * 0.05
    res = torch.randn(M, N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    ops.gemm_nt(z, N, N, None, N, M, N, C=y, ldc=N, split=ops.split_f16(W2, frag=True), a_affine=(sc, sh, True), group_rows=rows, res=res, ldr=N)
    zn = torch.relu((zg - zg.mean(1, keepdim=True)) / torch.sqrt(zg.var(1, unbiased=False, keepdim=True) + 1e-5)).view(M, N)
    ref = zn @ W2.double().t() + res.double()
    assert (y.double() - ref).abs().max().item() < 2e-5
    # batched: 4 problems of 16384 x 1024 x 64 with their own weight images (the similarity GEMM's form)
    Bn, Mb, Nb, Kb = 4, 16384, 1024, 64
    Ab = torch.randn(Bn, Mb, Kb, device="cuda"); Wb = torch.randn(Bn, Nb, Kb, device="cuda")
    imgs = [ops.split_f16(Wb[i], frag=True) for i in range(Bn)]
    img = {"W_hi": torch.stack([im["W_hi"] for im in imgs]).contiguous(), "W_lo": torch.stack([im["W_lo"] for im in imgs]).contiguous(), "inv_scale": imgs[0]["inv_scale"],
           "variant": imgs[0]["variant"], "ldb_h": imgs[0]["ldb_h"], "sB": imgs[0]["W_hi"].numel()}
    if all(im["inv_scale"] == imgs[0]["inv_scale"] for im in imgs):
        Cb = torch.empty(Bn, Mb, Nb, device="cuda")
        ops.gemm_nt(Ab, Kb, Kb, None, Kb, Mb, Nb, C=Cb, ldc=Nb, batch=(Bn, 1), sA=(Mb * Kb, 0), sC=(Mb * Nb, 0), split=img)
        refb = torch.einsum("bmk,bnk->bmn", Ab[:, :512].double(), Wb.double())
        assert (Cb[:, :512].double() - refb).abs().max().item() < 2e-5
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    A[777, 5] = 1e5
    ops.gemm_nt(A, K, K, None, K, M, N, C=z, ldc=N, split=ops.split_f16(W, frag=True), overflow=flag)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("head_act,with_map", [("none", False), ("sigmoid", False), ("sigmoid", True)])
def test_gemm_with_fused_cout1_head(ops, head_act, with_map):
    """A 256-channel layer followed by a Cout = 1 convolution (proj.0 + proj.3, overlap.3 + overlap.6 of models/gmmreg.py:30-47): the head runs in the
    layer's epilogue (struct ogmm_gemm.rd_*), the 256-wide map is written only on request."""
    M, N, K = 65536, 256, 512
    if ops._lib.load().ogmm_gemm_rowdot_fusable(M, N, K, 0) != 1:
        pytest.skip("the engine does not take the fused head for this shape")
    torch.manual_seed(17)
    A = torch.relu(torch.randn(M, K, device="cuda")) * (torch.rand(M, 1, device="cuda") * 2)
    W = torch.randn(N, K, device="cuda") * 0.05
    scale, shift = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
    w, b = torch.randn(N, device="cuda") * 0.2, torch.randn(1, device="cuda")
    out = torch.zeros(M, 3, device="cuda")
    ymap = torch.full((M, N), float("nan"), device="cuda") if with_map else None
    code = {"none": ops.ACT_NONE, "sigmoid": ops.ACT_SIGMOID}[head_act]
    ops.gemm_nt(A, K, K, None, K, M, N, C=ymap, ldc=N if with_map else 0, scale=scale, shift=shift, act=ops.ACT_RELU, split=ops.split_f16(W, frag=True),
                head=(w, b, code, out[:, 1], 3))
    y = torch.relu(A.double() @ W.double().t() * scale.double() + shift.double())
    ref = y @ w.double() + b.double()
    if head_act == "sigmoid":
        ref = torch.sigmoid(ref)
    tol = 2e-6 * ((y.abs() @ w.double().abs()) + 1.0) if head_act == "none" else 2e-6
    assert ((out[:, 1].double() - ref).abs() / tol).max().item() < 1.0
    assert float(out[:, 0].abs().max()) == 0.0 and float(out[:, 2].abs().max()) == 0.0
    if with_map:
        assert (ymap.double() - y).abs().max().item() < 2e-5
    # and the unfused pair of launches agrees with the fused one to rounding
    y32 = torch.empty(M, N, device="cuda")
    ops.gemm_nt(A, K, K, None, K, M, N, C=y32, ldc=N, scale=scale, shift=shift, act=ops.ACT_RELU, split=ops.split_f16(W, frag=True))
    out2 = torch.zeros(M, 3, device="cuda")
    ops.rowdot(y32, w, b, code, out2[:, 1], ldy=3)
    assert (out2[:, 1] - out[:, 1]).abs().max().item() < (2e-5 if head_act == "none" else 2e-6)


@pytest.mark.parametrize("with_map", [False, True])
def test_gemm_with_gathered_rows(ops, with_map):
    """index_points in front of a convolution (the anchors' K/V projection, models/gmmreg.py:54, 67-68): the GEMM's operand DMA gathers the rows
    (struct ogmm_gemm.a_gather_*); bit-identical to gather_rows + the same GEMM."""
    C, N, S, D, Cout = 128, 1024, 128, 512, 1024
    if ops._lib.load().ogmm_gemm_gather_fusable(C * S, Cout, D, C * N) != 1:
        pytest.skip("the engine does not take gathered rows for this shape")
    torch.manual_seed(23)
    feats = torch.randn(C * N, D, device="cuda")
    ids = torch.stack([torch.randperm(N)[:S] for _ in range(C)]).to(torch.int32).cuda()
    cmap = (torch.arange(C) ^ 1).to(torch.int32).cuda() if with_map else None
    W = torch.randn(Cout, D, device="cuda") * 0.05
    layer = {"W": W, "scale": None, "shift": torch.randn(Cout, device="cuda"), "split": ops.split_f16(W, frag=True)}
    got = ops.conv1x1_gathered(feats, C, N, ids, layer, cloud_map=cmap)
    a = ops.gather_rows(feats, D, C, N, D, ids, cloud_map=cmap).view(C * S, D)
    ref = ops.conv1x1(a, layer)
    assert torch.equal(got, ref)
    src = (cmap.long() if with_map else torch.arange(C, device="cuda"))
    rows = (src[:, None] * N + ids.long()[src]).reshape(-1)
    assert (got[:512].double() - (feats[rows[:512]].double() @ W.double().t() + layer["shift"].double())).abs().max().item() < 2e-5


def test_gemm_batched_strided_row_affine(ops):
    torch.manual_seed(1)
    Co, Hh, N, M, dh = 3, 4, 130, 32, 16
    D = Hh * dh
    q, kk = torch.randn(Co * N, D), torch.randn(Co * M, D)
    S = torch.empty(Co, Hh, N, M, device="cuda")
    ops.gemm_nt(dev(q), D, dh, dev(kk), D, N, M, C=S, ldc=M, alpha=0.25, batch=(Co, Hh), sA=(N * D, dh), sB=(M * D, dh), sC=(Hh * N * M, N * M))
    ref = 0.25 * torch.einsum("cnhd,cmhd->chnm", q.view(Co, N, Hh, dh).double(), kk.view(Co, M, Hh, dh).double())
    assert (S.cpu().double() - ref).abs().max().item() < 1e-5
    W, bias, anchors = torch.randn(D, D), torch.randn(D), torch.randn(Co, M, D)
    vT = torch.empty(Co, D, M, device="cuda")
    ops.gemm_nt(dev(W), D, D, dev(anchors), D, D, M, C=vT, ldc=M, shift=dev(bias), row_affine=True, batch=(Co, 1), sB=(M * D, 0), sC=(D * M, 0))
    ref = torch.einsum("ok,cmk->com", W.double(), anchors.double()) + bias.double()[None, :, None]
    assert (vT.cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("Bn,N,D", [(3, 256, 128), (2, 200, 64), (64, 1024, 512)])
def test_gemm_activation_times_activation_batched(ops, Bn, N, D):
    """models/gmmreg.py:75: S[b] = src_fn[b] tgt_fn[b]^T with the tgt side packed into split fragment images per pair."""
    torch.manual_seed(N)
    a, b = torch.randn(Bn * N, D), torch.randn(Bn * N, D)
    ad, bd = dev(a), dev(b)
    S = torch.empty(Bn, N, N, device="cuda")
    img = ops.pack_frag_batched(bd, Bn, N)
    ops.gemm_nt(ad, D, D, None, D, N, N, C=S, ldc=N, batch=(Bn, 1), sA=(N * D, 0), sC=(N * N, 0), split=img)
    nb = min(Bn, 3)
    ref = torch.einsum("bmd,bnd->bmn", a.view(Bn, N, D)[:nb].double(), b.view(Bn, N, D)[:nb].double())
    bound = 6e-7 * torch.einsum("bmd,bnd->bmn", a.view(Bn, N, D)[:nb].double().abs(), b.view(Bn, N, D)[:nb].double().abs()) + 1e-7
    assert ((S[:nb].cpu().double() - ref).abs() / bound).max().item() < 1.0
    if Bn > 3:
        S32 = torch.empty(Bn, N, N, device="cuda")
        ops.gemm_nt(ad, D, D, bd, D, N, N, C=S32, ldc=N, batch=(Bn, 1), sA=(N * D, 0), sB=(N * D, 0), sC=(N * N, 0))
        assert (S - S32).abs().max().item() < 3e-4          # |S| up to ~100 here: both engines are ~1e-6 relative


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("k,Cout", [(20, 64), (20, 128), (12, 256), (7, 64)])
def test_gemm_edge_pooling(ops, k, Cout, engine):
    torch.manual_seed(k)
    P, Cin = 37, 64
    h = torch.randn(P * k, Cin)
    W, s, t = torch.randn(Cout, Cin) / 8, torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1
    pool = torch.full((P, 512), -1.0, device="cuda")
    layer = {"W": dev(W), "scale": dev(s), "shift": dev(t), "split": _split(ops, W, engine)}
    out = ops.edgeconv_layer(dev(h), layer, k, pool[:, 64:64 + Cout], split=engine != "f32")
    ref = torch.relu(_gemm_ref(h, W) * s.double() + t.double())
    assert (out.cpu().double() - ref).abs().max().item() < 1e-5
    assert (pool[:, 64:64 + Cout].cpu().double() - ref.view(P, k, Cout).max(1)[0]).abs().max().item() < 1e-5
    assert bool((torch.cat([pool[:, :64], pool[:, 64 + Cout:]], 1) == -1.0).all()), "pooled write left its column slab"
    pool2 = torch.zeros((P, Cout), device="cuda")
    assert ops.edgeconv_layer(dev(h), layer, k, pool2, store=False, split=engine != "f32") is None
    assert torch.equal(pool2, pool[:, 64:64 + Cout].contiguous())


@pytest.mark.parametrize("C,N,k", [(4, 1024, 20), (3, 200, 12), (2, 717, 20), (1, 333, 32), (2, 64, 7)])
def test_edgeconv_fused_matches_oracle_dgcnn_front(ops, C, N, k):
    """models/dgcnn.py:135-150: xcat = cat(x1..x4) against the oracle's EdgeConv chain (same kNN graph)."""
    from ogmm_amd.gmmreg import pack_weights, state_spec
    torch.manual_seed(k)
    sd = {key: torch.zeros(shape, dtype=torch.int64 if key.endswith("num_batches_tracked") else torch.float32) for key, shape in state_spec(512)}
    synth.fill_state_dict(sd)
    L = pack_weights({key: v.cuda() for key, v in sd.items()}, 512, 4)
    xyz = clouds(C, N, seed=61)
    idx = O.knn_indices(xyz, k)
    emd = [L["emd1"], L["emd2"], L["emd3"], L["emd4"]]
    assert ops.edgeconv_fused_supported(k, emd)
    xcat = torch.full((C * N, 512), -1.0, device="cuda")
    ops.edgeconv_fused(dev(xyz), dev(idx.int()), emd, xcat)
    # oracle: the four pooled maps (fp64 evaluation of the same chain)
    Pd = {key: (v.double() if v.is_floating_point() else v) for key, v in sd.items()}
    h = O.edge_features(xyz.transpose(1, 2).double(), idx)
    pooled = []
    for l in (1, 2, 3, 4):
        h = torch.relu(O._bn(Pd, "emd.bn%d" % l, O._conv(Pd, "emd.conv%d" % l, h)))
        pooled.append(h.max(dim=-1)[0])
    ref = torch.cat(pooled, 1).transpose(1, 2).reshape(C * N, 512)
    err = (xcat.cpu().double() - ref).abs().max().item()
    assert err < 5e-6 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("C,N", [(128, 1024), (3, 717), (1, 41), (1, 24)])
def test_edgeconv_two_kernels_are_bit_identical(raw_ops, monkeypatch, C, N):
    """k = 20: the barrier-phased kernel (ops.EDGECONV_PC = False) and the producer / consumer pipeline (the default) write the same xcat, bit for bit -- whole
    chip (more tiles than workgroups), ragged point counts, a few tiles only -- on the sharp weight family."""
    from ogmm_amd.gmmreg import pack_weights, state_spec
    ops = raw_ops
    sd = {key: torch.zeros(shape, dtype=torch.int64 if key.endswith("num_batches_tracked") else torch.float32) for key, shape in state_spec(512)}
    synth.fill_state_dict(sd, profile="sharp")
    L = pack_weights({key: v.cuda() for key, v in sd.items()}, 512, 4)
    xyz = dev(clouds(C, N, seed=7)) if N >= 33 else dev(torch.randn(C, N, 3, generator=torch.Generator().manual_seed(N)))
    k = 20 if N >= 20 else N
    if k != 20:
        pytest.skip("k = 20 only")
    idx = ops.knn(xyz, k)
    emd = [L["emd1"], L["emd2"], L["emd3"], L["emd4"]]
    outs = []
    for mode in (False, True):
        monkeypatch.setattr(ops, "EDGECONV_PC", mode)
        xcat = torch.full((C * N, 512), float("nan"), device="cuda")
        st = torch.zeros(1, dtype=torch.int32, device="cuda")
        ops.edgeconv_fused(xyz, idx, emd, xcat, status=st)
        assert int(st.item()) == 0 and bool(torch.isfinite(xcat).all())
        outs.append(xcat)
    assert torch.equal(outs[0], outs[1]), (outs[0] - outs[1]).abs().max().item()


@pytest.mark.parametrize("C,N,Cmid,Cout", [(2, 256, 256, 128), (64, 1024, 1024, 512)])
def test_gemm_fused_instance_norm(ops, C, N, Cmid, Cout):
    """conv -> InstanceNorm1d -> ReLU -> conv (models/attn.py:17-27): statistics from the first GEMM's epilogue, the
    normalisation applied while the second GEMM stages its A operand.  Both the 128x128 and the 256x256 engines."""
    torch.manual_seed(C + N)
    Cin = 128
    x = torch.randn(C * N, Cin)
    W0, b0 = torch.randn(Cmid, Cin) / 11, torch.randn(Cmid) * 0.5
    W1, b1 = torch.randn(Cout, Cmid) / 16, torch.randn(Cout) * 0.1
    L0 = {"W": dev(W0), "shift": dev(b0), "split": ops.split_f16(dev(W0), frag=True)}
    L1 = {"W": dev(W1), "shift": dev(b1), "split": ops.split_f16(dev(W1), frag=True)}
    assert ops.instnorm_fusable(L0["split"], N)
    stats = torch.zeros((C, Cmid, 2), dtype=torch.float64, device="cuda")
    xd = dev(x)
    z = ops.conv1x1(xd, L0, col_stats=stats, group_rows=N, split=True)
    sc, sh = ops.instnorm_finalize(stats, N)
    y = ops.conv1x1(z, L1, a_affine=(sc, sh, True), group_rows=N, split=True)
    # unfused path of this library, and an fp64 reference on a few clouds
    z2 = ops.conv1x1(xd, L0, split=True)
    ops.instnorm_relu_(z2, C, N)
    y2 = ops.conv1x1(z2, L1, split=True)
    assert (y - y2).abs().max().item() < 2e-5 * max(1.0, y2.abs().max().item())
    nc = min(C, 2)
    zr = x[:nc * N].double() @ W0.double().t() + b0.double()
    zr = torch.relu(torch.nn.functional.instance_norm(zr.view(nc, N, Cmid).transpose(1, 2), eps=1e-5)).transpose(1, 2).reshape(nc * N, Cmid)
    yr = zr @ W1.double().t() + b1.double()
    assert (y[:nc * N].cpu().double() - yr).abs().max().item() < 2e-5 * max(1.0, yr.abs().max().item())


@pytest.mark.parametrize("C,N,M", [(3, 1024, 128), (2, 717, 128), (2, 200, 32), (1, 300, 64)])
def test_fused_attention(ops, C, N, M):
    """models/attn.py:78-82 with head-major channels (c = h*dh + d)."""
    torch.manual_seed(N + M)
    H, dh = 4, 128
    D = H * dh
    q, k, v = torch.randn(C * N, D), torch.randn(C * M, D), torch.randn(C * M, D)
    got = ops.attention(dev(q), dev(k), dev(v), C, N, M, H).cpu().double()
    qd, kd, vd = q.double().view(C, N, H, dh), k.double().view(C, M, H, dh), v.double().view(C, M, H, dh)
    prob = torch.softmax(torch.einsum("cnhd,cmhd->chnm", qd, kd) / dh ** .5, dim=-1)
    ref = torch.einsum("chnm,cmhd->cnhd", prob, vd).reshape(C * N, D)
    assert (got - ref).abs().max().item() < 2e-6
    # strided views (keys | values produced by one GEMM)
    kv = torch.cat([k, v], 1)
    kvd = dev(kv)
    got2 = ops.attention(dev(q), kvd[:, :D], kvd[:, D:], C, N, M, H).cpu().double()
    assert torch.equal(got2, got)
    # the variant that stages K / V per workgroup instead of reading the packed images
    got3 = ops.attention(dev(q), dev(k), dev(v), C, N, M, H, use_workspace=False).cpu().double()
    assert (got3 - ref).abs().max().item() < 2e-6


@pytest.mark.parametrize("split", [False, 1, 2])
@pytest.mark.parametrize("C,N", [(2, 1024), (3, 300), (1, 717), (2, 20)])
def test_attention_backward_kernel(ops, C, N, split):
    """Kernel T11 against fp64 autograd of models/attn.py:78-82: dq, dk, dv for whole and ragged query tiles, a large-scale dO (the
    trainer's 2^16 loss scale), strided q / k / v / dO views; every output element is written (the buffers start as NaN).
    split = 1 (round 5): the two products over the head dimension on the engines' fp16x3 arithmetic; split = 2: all five (P as P * 2^10, dS with a per-tile
    power of two) -- same bar, also with dO tiles spanning 2^-10 ... 2^4 (tiles whose dS differ by many binades: the running dK is re-scaled between them);
    an operand beyond binary16 sets the overflow word."""
    torch.manual_seed(C * 1000 + N)
    H, dh, M = 4, 128, 128
    D = H * dh
    assert ops.attention_bwd_supported(M, dh) and not ops.attention_bwd_supported(32, dh)
    q, k, v = torch.randn(C * N, D) * 1.5, torch.randn(C * M, D) * 1.5, torch.randn(C * M, D)
    g = torch.randn(C * N, D) * 300.0
    qd, kd, vd = (t_.double().requires_grad_(True) for t_ in (q, k, v))
    prob = torch.softmax(torch.einsum("cnhd,cmhd->chnm", qd.view(C, N, H, dh), kd.view(C, M, H, dh)) / dh ** .5, dim=-1)
    out = torch.einsum("chnm,cmhd->cnhd", prob, vd.view(C, M, H, dh)).reshape(C * N, D)
    rq, rk, rv = torch.autograd.grad(out, (qd, kd, vd), g.double(), retain_graph=True)
    ovf = torch.zeros(1, dtype=torch.int32, device="cuda")
    dq, dk, dv = ops.attention_bwd(dev(q), dev(k), dev(v), dev(g), C, N, M, H, split=split, overflow=ovf)
    assert int(ovf.item()) == 0
    for name, got, ref in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        got = got.cpu().double()
        assert torch.isfinite(got).all(), name
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-6, (name, err)
    # strided operands: q | k-padding in one buffer, keys | values produced by one GEMM, dO a column slice
    kv = dev(torch.cat([k, v], 1))
    qg = dev(torch.cat([q, g], 1))
    dq2, dk2, dv2 = ops.attention_bwd(qg[:, :D], kv[:, :D], kv[:, D:], qg[:, D:], C, N, M, H, split=split, overflow=ovf)
    assert torch.equal(dq2, dq) and torch.equal(dk2, dk) and torch.equal(dv2, dv)
    if split:
        # gradient TILES (32 queries) of very different magnitude, 2^-10 ... 2^4, so that consecutive tiles take different powers of two for dS.  (Within a
        # tile the rows share a scale: the binary16 split of dO has an absolute floor of 2^-25 per element, as for every operand of the engines, so rows far
        # below their map's maximum are not resolved relative to themselves by any of the forms -- the bar is relative to the tile.)
        ex = torch.randint(-10, 5, ((N + 31) // 32,)).float().repeat_interleave(32)[:N].repeat(C)
        gw = g * torch.exp2(ex)[:, None]
        rq, rk, rv = torch.autograd.grad(out, (qd, kd, vd), gw.double())
        dq, dk, dv = ops.attention_bwd(dev(q), dev(k), dev(v), dev(gw), C, N, M, H, split=split, overflow=ovf)
        assert int(ovf.item()) == 0
        for name, got, ref in (("dk", dk, rk), ("dv", dv, rv)):
            err = (got.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
            assert err < 2e-6, (name, err)
        rowmax = rq.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)          # dq rows inherit their query's scale: compare row by row
        err = ((dq.cpu().double() - rq).abs() / rowmax).max().item()
        assert err < 2e-5, ("dq rows", err)
        g2 = g.clone()
        g2[N // 2, 7] = 1.0e5                                  # beyond binary16
        ops.attention_bwd(dev(q), dev(k), dev(v), dev(g2), C, N, M, H, split=split, overflow=ovf)
        assert int(ovf.item()) & 1
        # a NaN operand (v_max_f32 drops it from the running maximum) must raise the word too -- in every one of the four operands
        for which in range(4):
            ovf.zero_()
            t4 = [q.clone(), k.clone(), v.clone(), g.clone()]
            t4[which][t4[which].shape[0] // 3, 5] = float("nan")
            ops.attention_bwd(dev(t4[0]), dev(t4[1]), dev(t4[2]), dev(t4[3]), C, N, M, H, split=split, overflow=ovf)
            assert int(ovf.item()) & 1, "NaN in operand %d not flagged" % which
        ovf.zero_()


def test_attention_backward_in_autograd(ops, monkeypatch):
    """train_ops._Attention: the kernel path and the library path (train_ops.FUSED_ATTENTION_BWD = False) give the same gradients"""
    from ogmm_amd import train_ops
    torch.manual_seed(5)
    C, N, M, H, D = 2, 256, 128, 4, 512
    q, k, v, g = (dev(torch.randn(r_, D)) for r_ in (C * N, C * M, C * M, C * N))
    grads = []
    for fused in (True, False):
        monkeypatch.setattr(train_ops, "FUSED_ATTENTION_BWD", fused)
        a, b, c_ = (t_.clone().requires_grad_(True) for t_ in (q, k, v))
        o = train_ops.TrainOps().attention(a, b, c_, C, N, M, H)
        grads.append(torch.autograd.grad(o, (a, b, c_), g))
    for x, y in zip(*grads):
        assert (x - y).abs().max().item() <= 2e-5 * y.abs().max().item()


@pytest.mark.parametrize("batch,rows", [(3, 256), (2, 300), (1, 1024)])
def test_l2norm_pack_frag_equals_two_steps(ops, batch, rows):
    """The fused normalise + split-image kernel against pack_frag_batched(l2norm_rows(x)): bit-identical images (same sums, same division)."""
    torch.manual_seed(rows)
    x = dev(torch.randn(batch * rows, 512) * 3.0)
    two = ops.pack_frag_batched(ops.l2norm_rows(x), batch, rows)
    one = ops.l2norm_pack_frag_batched(x, batch, rows)
    assert torch.equal(one["W_hi"], two["W_hi"]) and torch.equal(one["W_lo"], two["W_lo"])
    assert one["sB"] == two["sB"] and one["ldb_h"] == two["ldb_h"]


# ------------------------------------------------------------------------------------------------ row / column kernels
def test_softmax_instnorm_l2norm_rowdot(ops):
    torch.manual_seed(2)
    x = torch.randn(1000, 128) * 3
    got = ops.softmax_rows_(dev(x)).cpu()
    assert (got - torch.softmax(x, -1)).abs().max().item() < 2e-7
    C, N, D = 3, 717, 192
    z = torch.randn(C * N, D) * 2 + 0.3
    got = ops.instnorm_relu_(dev(z), C, N).cpu()
    ref = torch.relu(torch.nn.functional.instance_norm(z.view(C, N, D).transpose(1, 2), eps=1e-5)).transpose(1, 2).reshape(C * N, D)
    assert (got - ref).abs().max().item() < 2e-6
    y = torch.randn(500, 512)
    assert (ops.l2norm_rows(dev(y)).cpu() - torch.nn.functional.normalize(y, dim=1)).abs().max().item() < 2e-7
    w, b = torch.randn(512) / 20, torch.randn(1)
    out = torch.empty(500, device="cuda")
    ops.rowdot(dev(y), dev(w), dev(b), ops.ACT_SIGMOID, out)
    assert (out.cpu() - torch.sigmoid(y @ w + b)).abs().max().item() < 1e-6


@pytest.mark.parametrize("B,N,two_pass", [(2, 300, False), (2, 300, True), (1, 1024, False), (2, 1100, False), (1, 2048, False)])
def test_overlap_cross(ops, B, N, two_pass):
    """One-pass tiles + merge (row blocks of 64, column panels of 1024: N = 1100 / 2048 need two panels) and the older two-kernel form."""
    torch.manual_seed(3)
    S = torch.rand(B, N, N) * 2 - 1
    o = torch.randn(2 * B * N, 4)
    out = torch.zeros(2 * B * N, 4, device="cuda")
    od = dev(o)
    ops.overlap_cross(dev(S), od[:B * N, 1], od[B * N:, 1], 4, out[:B * N, 0], out[B * N:, 0], 4, two_pass=two_pass)
    o_src, o_tgt = o[:B * N, 1].view(B, 1, N), o[B * N:, 1].view(B, 1, N)
    wo_s = torch.einsum("bmn,bdn->bdm", torch.softmax(S.double(), -1), o_src.double()).reshape(-1)      # models/gmmreg.py:79
    wo_t = torch.einsum("bmn,bdm->bdn", torch.softmax(S.double(), 1), o_tgt.double()).reshape(-1)       # models/gmmreg.py:80
    assert (out[:B * N, 0].cpu().double() - wo_s).abs().max().item() < 1e-6
    assert (out[B * N:, 0].cpu().double() - wo_t).abs().max().item() < 1e-6
    assert float(out[:, 1:].abs().max()) == 0.0


@pytest.mark.parametrize("B,N", [(16, 1024), (4, 2048)])
def test_overlap_block_fused_into_the_similarity_gemm(ops, B, N):
    """models/gmmreg.py:75-80 with S living only in the GEMM's accumulators (struct ogmm_gemm.ovl_rowpart) against fp64 and against the unfused path
    (N = 2048: eight partial softmax-dots per row and column, the shape of BASELINE configs[2] / [3])."""
    D = 512
    if not ops.overlap_fusable(B, N, D):
        pytest.skip("the engine does not take the fused form for this shape")
    torch.manual_seed(11)
    f = torch.randn(2 * B * N, D) * (torch.rand(2 * B * N, 1) * 4 + 0.1)
    f[5] = 0.0                                                     # a zero row: F.normalize's eps path
    o = torch.randn(2 * B * N, 4)
    fd, od = dev(f), dev(o)
    out = torch.zeros(2 * B * N, 4, device="cuda")
    tgt_img = ops.l2norm_pack_frag_batched(fd[B * N:], B, N)
    ops.overlap_fused(fd[:B * N], tgt_img, B, N, D, od[:B * N, 1], od[B * N:, 1], 4, out[:B * N, 0], out[B * N:, 0], 4)
    # the unfused path of the forward
    out2 = torch.zeros(2 * B * N, 4, device="cuda")
    S = torch.empty(B, N, N, device="cuda")
    ops.gemm_nt(ops.l2norm_rows(fd[:B * N]), D, D, None, D, N, N, C=S, ldc=N, batch=(B, 1), sA=(N * D, 0), sC=(N * N, 0), split=tgt_img)
    ops.overlap_cross(S, od[:B * N, 1], od[B * N:, 1], 4, out2[:B * N, 0], out2[B * N:, 0], 4)
    fn = torch.nn.functional.normalize(f.double(), dim=1)
    Sd = torch.einsum("bmd,bnd->bmn", fn[:B * N].view(B, N, D), fn[B * N:].view(B, N, D))
    o_src, o_tgt = o[:B * N, 1].view(B, 1, N).double(), o[B * N:, 1].view(B, 1, N).double()
    wo_s = torch.einsum("bmn,bdn->bdm", torch.softmax(Sd, -1), o_src).reshape(-1)      # models/gmmreg.py:79
    wo_t = torch.einsum("bmn,bdm->bdn", torch.softmax(Sd, 1), o_tgt).reshape(-1)       # models/gmmreg.py:80
    e_fused = max((out[:B * N, 0].cpu().double() - wo_s).abs().max().item(), (out[B * N:, 0].cpu().double() - wo_t).abs().max().item())
    e_plain = max((out2[:B * N, 0].cpu().double() - wo_s).abs().max().item(), (out2[B * N:, 0].cpu().double() - wo_t).abs().max().item())
    print("OVERLAP fused vs fp64 %.2e   unfused vs fp64 %.2e" % (e_fused, e_plain))
    assert e_fused < 2e-6 and e_plain < 2e-6
    assert float(out[:, 1:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ GMM head
@pytest.mark.parametrize("C,N,J", [(3, 2048, 64), (2, 717, 64), (5, 300, 20), (40, 1500, 32)])
def test_gmm_em_resident_kernel_equals_the_launch_sequence(ops, monkeypatch, C, N, J):
    """The resident E/M kernel (one launch, per-cloud barriers in global memory, tickets) against the sequence of grid-wide launches: same
    arithmetic per entry, so gamma, pi, mu agree to the last bit -- also when the grid exceeds what is resident at once (40 clouds x 6 chunks
    with three workgroups per CU still fits; the forced mode below runs it regardless of the size gate)."""
    torch.manual_seed(C * 100 + J)
    xyz = dev(torch.randn(C, N, 3) * 0.5)
    o = dev(torch.rand(C, N))
    ids = ops.fps(xyz, J, None)
    monkeypatch.setenv("OGMM_EM_RESIDENT", "0")
    ref = ops.gmm_em(xyz, o, ids, engine="multi")
    monkeypatch.setenv("OGMM_EM_RESIDENT", "1")
    for _ in range(3):
        got = ops.gmm_em(xyz, o, ids, engine="multi")
        for a, b in zip(got, ref):
            assert torch.equal(a, b)


def test_gmm_em_reports_the_sinkhorn_residual(ops):
    """The quantity the reference's early exit tests (lib/utils.py:99-102: sum|du| + sum|dv| per cloud and sweep) as a diagnostic output, on every kernel
    family; reporting it changes nothing, and with the exit off (thresh = 0) all sweeps run."""
    torch.manual_seed(9)
    for C, N, J, engine in ((4, 1024, 16, None), (2, 2048, 64, None), (2, 717, 128, None), (3, 300, 20, "chip"), (2, 2048, 64, "chip")):
        xyz = clouds(C, N, seed=33)
        o = torch.sigmoid(torch.randn(C, N))
        res = []
        O.weighted_em(xyz, torch.zeros(C, N, 4), o, J, resid=res)
        ref = torch.stack(res).view(10, 10, C).permute(2, 0, 1)                      # [C, iters, sweeps]
        ids = ops.fps(dev(xyz), J, None)
        out = ops.gmm_em(dev(xyz), dev(o), ids, engine=engine, return_resid=True, return_sweeps=True)
        got = out[3].cpu()
        assert got.shape == (C, 10, 10) and not torch.isnan(got).any(), (C, N, J, engine)
        assert ((got - ref).abs() / ref.abs().clamp_min(1e-6)).max().item() < 2e-3, ((got - ref).abs() / ref.abs()).max().item()
        assert ref.mean(0).min().item() > 1e-2 and bool((out[4] == 10).all())        # the oracle ran all 10 sweeps on this input, like the kernel
        for kw in (dict(), dict(thresh=0.0)):                                        # without the outputs / with the exit off nothing changes
            plain = ops.gmm_em(dev(xyz), dev(o), ids, engine=engine, **kw)
            assert torch.equal(plain[0], out[0]) and torch.equal(plain[2], out[2])


EXIT_CASES = [  # C, N, J, group, engine, OGMM_EM_RESIDENT, scale, what
    (4, 1024, 16, 2, None, None, 0.04, "on-chip kernel, J = 16 register form"),
    (6, 300, 20, 3, "chip", None, 0.05, "on-chip kernel, generic cached form"),
    (4, 2048, 64, 2, "chip", None, 0.04, "on-chip kernel, recomputing form"),
    (4, 2048, 64, 2, "multi", "1", 0.04, "resident kernel"),
    (4, 2048, 64, 2, "multi", "0", 0.04, "fused launch sequence"),
    (6, 1500, 32, 3, "multi", "0", 0.04, "fused launch sequence, J = 32, ragged chunks"),
    (6, 1500, 32, 6, "multi", "1", 0.04, "resident kernel, one group"),
    (2, 717, 128, 1, None, None, 0.04, "two-launch sweeps (J > 64), every cloud its own call"),
    (4, 717, 128, 2, None, None, 0.04, "two-launch sweeps (J > 64)"),
]


@pytest.mark.parametrize("C,N,J,G,engine,resident,scale,what", EXIT_CASES)
def test_gmm_em_early_exit_matches_the_reference_semantics(ops, monkeypatch, C, N, J, G, engine, resident, scale, what):
    """lib/utils.py:99-102: an E-step's sweeps end after the first sweep whose residual, averaged over the clouds of ONE wkeans_plus call (a group),
    is below 1e-2.  Scaled-down clouds make that happen after 2-9 sweeps; every kernel family must run exactly the oracle's number of sweeps in
    every E-step of every group and end with its gamma / pi / mu."""
    if resident is not None:
        monkeypatch.setenv("OGMM_EM_RESIDENT", resident)
    torch.manual_seed(N + J + C)
    xyz = clouds(C, N, seed=41) * scale
    o = torch.sigmoid(torch.randn(C, N))
    want, pis, mus, gammas, errs = [], [], [], [], []
    for g in range(C // G):          # the oracle, one call per group -- as the reference calls wkeans_plus once per cloud set
        h = slice(g * G, (g + 1) * G)
        st, rs = [], []
        gamma, pi, mu, _, _ = O.weighted_em(xyz[h], torch.zeros(G, N, 1), o[h], J, stats=st, resid=rs)
        means = torch.stack(rs).mean(1)
        assert ((means - 1e-2).abs() / 1e-2).min().item() > 0.01, "test input sits on the decision's knife edge"
        st64 = []
        _, pid, mud, _, _ = O.weighted_em(xyz[h].double(), torch.zeros(G, N, 1).double(), o[h].double(), J, stats=st64)
        errs.append(max((pi - pid).abs().max().item(), (mu - mud).abs().max().item()) if st64 == st else 0.0)
        want.append(st); pis.append(pi); mus.append(mu); gammas.append(gamma)
    want = torch.tensor(want, dtype=torch.int32)
    assert int(want.min()) < 10, "the exit never fired in the oracle: nothing tested"
    ids = ops.fps(dev(xyz), J, None)
    for rep in range(2):             # twice: the workspace counters start from zero every call
        g_gamma, g_pi, g_mu, g_sweeps = ops.gmm_em(dev(xyz), dev(o), ids, engine=engine, thresh=1e-2, group_size=G, return_sweeps=True)
        assert torch.equal(g_sweeps.cpu(), want), (what, g_sweeps.cpu(), want)
        tol = max(4 * max(errs), 5e-7)
        pi, mu, gamma = torch.cat(pis), torch.cat(mus), torch.cat(gammas)
        assert not torch.isnan(g_mu).any()
        assert (g_pi.cpu() - pi).abs().max().item() < tol and (g_mu.cpu() - mu).abs().max().item() < tol, (what, tol)
        assert (g_gamma.cpu() - gamma).abs().max().item() < 2e-5
    # and the same clouds as ONE call give different sweep counts unless the groups happen to agree: the grouping is really per call
    if C // G > 1 and not bool((want[0] == want[1:]).all()):
        one = ops.gmm_em(dev(xyz), dev(o), ids, engine=engine, thresh=1e-2, group_size=None, return_sweeps=True)[3].cpu()
        assert one.shape == (1, 10)


@pytest.mark.parametrize("N,J,resident,what", [(1024, 16, None, "on-chip, J = 16"), (500, 24, None, "on-chip, generic J"), (2048, 64, "1", "resident kernel")])
def test_gmm_em_exit_protocol_timeout_is_reported_not_hung(ops, monkeypatch, N, J, resident, what):
    """The clouds of a call group wait for each other's residuals with BOUNDED polls.  Debug knobs force the failure the bound exists for: cloud 1 never
    publishes (OGMM_EM_DEBUG_LOSE_CLOUD: what a lost workgroup looks like to its group) and the poll limit is lowered (OGMM_EM_POLL_LIMIT) so that the test
    does not take the production limit's second.  The kernels must come back, pi / mu must be NaN-poisoned, and the status word handed to the call
    must carry STATUS_EM_EXIT_PROTOCOL; with the knobs gone the same call is clean again."""
    from ogmm_amd import _lib
    if resident is not None:
        monkeypatch.setenv("OGMM_EM_RESIDENT", resident)
    C = 4
    xyz = dev(clouds(C, N, seed=7) * 0.03)          # scaled down: the exit is live (decisions are being taken)
    o = dev(torch.sigmoid(torch.randn(C, N)))
    ids = ops.fps(xyz, J, None)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    monkeypatch.setenv("OGMM_EM_DEBUG_LOSE_CLOUD", "1")
    monkeypatch.setenv("OGMM_EM_POLL_LIMIT", "2000")
    _, pi, mu = ops.gmm_em(xyz, o, ids, thresh=1e-2, group_size=C, status=status)[:3]
    torch.cuda.synchronize()
    assert int(status.item()) & _lib.STATUS_EM_EXIT_PROTOCOL, what
    assert bool(torch.isnan(pi).any()) and bool(torch.isnan(mu).any()), what
    monkeypatch.delenv("OGMM_EM_DEBUG_LOSE_CLOUD")
    monkeypatch.delenv("OGMM_EM_POLL_LIMIT")
    status.zero_()
    _, pi, mu = ops.gmm_em(xyz, o, ids, thresh=1e-2, group_size=C, status=status)[:3]
    assert int(status.item()) == 0 and not bool(torch.isnan(pi).any()) and not bool(torch.isnan(mu).any())


def test_edgeconv_pipeline_protocol_timeout_is_reported_and_the_model_raises(ops, monkeypatch):
    """The producer / consumer EdgeConv kernel hands blocks over through LDS sequence counters with bounded waits.  OGMM_EDGECONV_POLL_LIMIT=1 makes every
    ordinary wait "time out": the kernel must come back, OR STATUS_EDGECONV_PROTOCOL into the status word (not rely on a poisoned output element that
    another workgroup's store could overwrite), and a model forward must raise instead of returning results built on it."""
    from argparse import Namespace
    from ogmm_amd import _lib, synth
    from ogmm_amd.gmmreg import GMMReg
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=16)
    model = GMMReg(512, 16, cfg)
    synth.fill_state_dict(model.state_dict())
    model = model.cuda().eval()
    model.overflow_policy = "sync"
    src, tgt, _, _ = synth.make_batch(0, 2, 512, "partial")
    starts = synth.fps_starts_for(0, 2, 512)
    with torch.no_grad():
        model(src.cuda(), tgt.cuda(), fps_starts=starts)          # clean
        monkeypatch.setenv("OGMM_EDGECONV_POLL_LIMIT", "1")
        with pytest.raises(_lib.OgmmError, match="protocol error"):
            model(src.cuda(), tgt.cuda(), fps_starts=starts)
        monkeypatch.delenv("OGMM_EDGECONV_POLL_LIMIT")
        out = model(src.cuda(), tgt.cuda(), fps_starts=starts)          # and clean again: the status word was consumed
    assert bool(torch.isfinite(out[0]).all())


@pytest.mark.parametrize("C,N,J,D", [(3, 2048, 64, 512), (2, 717, 40, 512), (2, 300, 17, 96), (1, 1025, 64, 260)])
def test_feat_mean_matrix_core_form(ops, C, N, J, D):
    """K16 for 16 < J <= 64 (v_mfma_f32_32x32x2_f32) against fp64: whole and ragged row counts, partial cluster and channel blocks, a row pitch wider
    than D (the features are a column view)."""
    torch.manual_seed(N + J)
    g = torch.softmax(torch.randn(C, N, J), -1)
    pi = g.mean(1)
    wide = torch.randn(C * N, D + 32)
    fd = dev(wide)[:, 16:16 + D]
    got = ops.gmm_feat_mean(dev(g), dev(pi), fd, C, N).cpu().double()
    f64 = wide[:, 16:16 + D].double().view(C, N, D)
    ref = torch.einsum("cnj,cnd->cjd", g.double(), f64) / (pi.double() * N + 1e-5)[:, :, None]
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 5e-6 * ref.abs().max().item()


@pytest.mark.parametrize("engine", [None, "chip", "multi"])
@pytest.mark.parametrize("C,N,J", [(4, 1024, 16), (2, 717, 128), (2, 2048, 64), (3, 200, 8)])
def test_gmm_em_and_feat_mean(ops, C, N, J, engine):
    torch.manual_seed(N + J)
    xyz = clouds(C, N, seed=31)
    o = torch.sigmoid(torch.randn(C, N))
    feats = torch.randn(C, N, 64)
    st = []
    gamma, pi, mu, muf, ids = O.weighted_em(xyz, feats, o, J, stats=st)
    assert set(st) == {10}
    ids_g = ops.fps(dev(xyz), J, None)
    assert torch.equal(ids_g.cpu().long(), ids)
    g_gamma, g_pi, g_mu = ops.gmm_em(dev(xyz), dev(o), ids_g, engine=engine)      # on-chip loop / grid-wide kernel sequence / automatic choice
    # fp64 run of the same algorithm = the yardstick both fp32 paths are measured against
    gd, pid, mud, _, _ = O.weighted_em(xyz.double(), feats.double(), o.double(), J)
    ref_err = max((pi - pid).abs().max().item(), (mu - mud).abs().max().item())
    got_err = max((g_pi.cpu() - pid).abs().max().item(), (g_mu.cpu() - mud).abs().max().item())
    print("EM-ACCURACY engine=%s C=%d N=%d J=%d: max |pi,mu - fp64| hip %.2e, torch fp32 %.2e" % (engine, C, N, J, got_err, ref_err))
    assert got_err < max(4 * ref_err, 2e-6), (got_err, ref_err)
    assert (g_gamma.cpu() - gd).abs().max().item() < max(4 * (gamma - gd).abs().max().item(), 2e-5)
    g_muf = ops.gmm_feat_mean(g_gamma, g_pi, dev(feats.view(C * N, 64)), C, N)
    mufd = O.gmm_moments(g_gamma.cpu().double(), feats.double())[1]
    assert (g_muf.cpu() - mufd).abs().max().item() < 2e-5


def test_kabsch_matches_lapack_and_fixes_reflections(ops):
    torch.manual_seed(4)
    B, J = 64, 16
    src = torch.randn(B, 3, J)
    Rg = torch.linalg.qr(torch.randn(B, 3, 3))[0]
    Rg = Rg * torch.sign(torch.det(Rg))[:, None, None]
    corr = Rg @ src + torch.randn(B, 3, 1) + 0.05 * torch.randn(B, 3, J)
    corr[:8] = corr[:8] * torch.tensor([1.0, 1.0, -1.0])[None, :, None]         # mirrored targets: det(V U^T) < 0 branch
    src[8:12, 2] = 0.0                                                          # coplanar sources
    w = torch.rand(B, 1, J)
    R, t = ops.kabsch(dev(src), dev(corr), dev(w))
    Ro, to = O.kabsch(src.double(), corr.double(), w.double())
    assert O.rotation_error_rad(R.cpu(), Ro).max().item() < 2e-6
    assert (t.cpu().double() - to).abs().max().item() < 2e-6
    assert (torch.det(R.cpu().double()) - 1).abs().max().item() < 1e-6


@pytest.mark.parametrize("B,J,D", [(4, 16, 512), (2, 128, 512), (3, 8, 64)])
def test_match_kabsch(ops, B, J, D):
    torch.manual_seed(J)
    mu_s, mu_t = torch.randn(B, J, 3), torch.randn(B, J, 3)
    f_s = torch.randn(B, J, D)
    f_t = f_s[:, torch.randperm(J)] + 0.3 * torch.randn(B, J, D)
    R, t, sc = ops.match_kabsch(dev(mu_s), dev(mu_t), dev(f_s), dev(f_t), 0.05, want_scores=True)
    Ro, to, sco = O.match_and_solve(mu_s.double(), mu_t.double(), f_s.double(), f_t.double())
    assert (sc.cpu().double() - sco).abs().max().item() < 2e-5
    assert O.rotation_error_rad(R.cpu(), Ro).max().item() < 1e-5
    assert (t.cpu().double() - to).abs().max().item() < 1e-5


@pytest.mark.parametrize("C,N,J,D", [(4, 1024, 16, 512), (2, 300, 64, 128)])
def test_clu_infonce(ops, C, N, J, D):
    torch.manual_seed(7)
    xyz = clouds(C, N, seed=41)
    feats = torch.randn(C, N, D)
    mu = xyz[:, torch.randperm(N)[:J]] + 0.01 * torch.randn(C, J, 3)
    muf = torch.randn(C, J, D)
    row_loss, near = ops.clu_infonce(dev(xyz), dev(mu), dev(feats.view(C * N, D)), dev(muf), C, N, 0.1)
    anchors, near_o = O.nearest_feats(xyz, mu, feats)
    assert torch.equal(near.cpu().long(), near_o)
    ref = O.info_nce(anchors.double(), muf.double(), 0.1)
    assert abs(row_loss.mean().item() - ref.item()) < 1e-5
