"""Parity of every HIP kernel (through the C-ABI) against the CPU oracle.  GPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def nat():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from dp_timeautodiff_amd import _native
    _native.load()
    return _native


@pytest.fixture(autouse=True)
def _k1_defaults():
    """dpk_linear_configure is process-wide: every test starts from, and leaves, the defaults."""
    yield
    if torch.cuda.is_available():
        from dp_timeautodiff_amd import _native
        _native.linear_configure(arithmetic=None, role_split=None)


def _arena(B, width, pad_front=64, pad_back=64):
    stride = pad_front + width + pad_back
    stride = (stride + 3) // 4 * 4
    base = torch.zeros(B, stride, device="cuda", dtype=torch.float32)
    return base, pad_front


LINEAR_SHAPES = [
    # (B, T, in, out)
    (3, 24, 400, 200), (8, 24, 208, 208), (5, 24, 13, 200), (4, 24, 200, 7), (7, 24, 7, 200),
    (2, 24, 128, 13), (9, 48, 1, 125), (3, 48, 1000, 250), (2, 48, 500, 7), (4, 5, 8, 32),
    (1, 1, 33, 65), (6, 7, 80, 80), (2, 70, 96, 40), (3, 256, 160, 96), (16, 3, 64, 64),
]


@pytest.mark.parametrize("B,T,in_f,out_f", LINEAR_SHAPES)
@pytest.mark.parametrize("bias", [True, False])
def test_linear_grad_sample(nat, B, T, in_f, out_f, bias):
    from oracle import opacus_ref as O
    g = torch.Generator().manual_seed(B * 1000 + T * 7 + in_f + out_f)
    A = torch.randn(B, T, in_f, generator=g)
    G = torch.randn(B, T, out_f, generator=g) * 0.01
    scale = float(B)
    gw_ref, gb_ref = O.linear_grad_sample(A, G * scale, bias)
    nw = in_f * out_f
    base, off = _arena(B, nw + out_f)
    slots = nat.linear_slots(in_f, out_f)
    part = torch.full((B, slots + 3), -1.0, device="cuda")
    vw = nat.FlatView(base, off, nw)
    vb = nat.FlatView(base, off + nw, out_f) if bias else None
    nat.linear_grad_sample(A.cuda(), G.cuda(), vw, vb, part, 2, scale)
    torch.cuda.synchronize()
    out = base.cpu()
    gw = out[:, off:off + nw].reshape(B, out_f, in_f)
    assert rel_err(gw.numpy(), gw_ref.numpy()) < TOL
    ref_sq = (gw_ref.double() ** 2).sum((1, 2))
    if bias:
        gb = out[:, off + nw:off + nw + out_f]
        assert rel_err(gb.numpy(), gb_ref.numpy()) < TOL
        ref_sq = ref_sq + (gb_ref.double() ** 2).sum(1)
    else:
        assert (out[:, off + nw:] == 0).all()
    # pads untouched
    assert (out[:, :off] == 0).all() and (out[:, off + nw + out_f:] == 0).all()
    p = part.cpu()
    assert (p[:, :2] == -1).all() and (p[:, 2 + slots:] == -1).all()
    assert rel_err(p[:, 2:2 + slots].double().sum(1).numpy(), ref_sq.numpy()) < TOL


@pytest.mark.parametrize("B,T,in_f,out_f", [(5, 24, 400, 200), (3, 48, 250, 1000), (4, 24, 208, 208), (2, 48, 500, 70)])
def test_linear_grad_sample_padded_rows(nat, B, T, in_f, out_f):
    """Row stride ld > in (arena pads awkward widths to whole 128-byte lines): pads are written as zeros."""
    from oracle import opacus_ref as O
    g = torch.Generator().manual_seed(in_f)
    A = torch.randn(B, T, in_f, generator=g)
    G = torch.randn(B, T, out_f, generator=g)
    ld = (in_f + 31) // 32 * 32
    gw_ref, gb_ref = O.linear_grad_sample(A, G * float(B), True)
    nw = ld * out_f
    base = torch.full((B, 64 + nw + out_f + 64), 7.0, device="cuda")
    off = 64
    part = torch.zeros(B, nat.linear_slots(in_f, out_f), device="cuda")
    nat.linear_grad_sample(A.cuda(), G.cuda(), nat.FlatView(base, off, nw), nat.FlatView(base, off + nw, out_f), part,
                           0, float(B), ld_w=ld)
    out = base.cpu()
    w = out[:, off:off + nw].reshape(B, out_f, ld)
    assert rel_err(w[:, :, :in_f].numpy(), gw_ref.numpy()) < TOL
    assert (w[:, :, in_f:] == 0).all()
    assert rel_err(out[:, off + nw:off + nw + out_f].numpy(), gb_ref.numpy()) < TOL
    assert (out[:, :off] == 7).all() and (out[:, off + nw + out_f:] == 7).all()
    ref_sq = (gw_ref.double() ** 2).sum((1, 2)) + (gb_ref.double() ** 2).sum(1)
    assert rel_err(part.cpu().double().sum(1).numpy(), ref_sq.numpy()) < TOL


def test_linear_grad_sample_is_deterministic(nat):
    g = torch.Generator().manual_seed(5)
    A = torch.randn(16, 24, 400, generator=g).cuda()
    G = torch.randn(16, 24, 200, generator=g).cuda()
    outs = []
    for _ in range(3):
        base, off = _arena(16, 400 * 200 + 200)
        part = torch.zeros(16, nat.linear_slots(400, 200), device="cuda")
        nat.linear_grad_sample(A, G, nat.FlatView(base, off, 80000), nat.FlatView(base, off + 80000, 200), part, 0, 16.0)
        outs.append((base.cpu(), part.cpu()))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])


@pytest.mark.parametrize("B,T,in_f,out_f", [(150, 24, 400, 200), (100, 48, 300, 260), (41, 31, 250, 1000)])
def test_linear_grad_sample_persistent_multi_item(nat, B, T, in_f, out_f):
    """More (sample, tile) items than the persistent grid has workgroups: every workgroup walks several items
    (register prefetch of the next item, LDS-only barriers, partial-slot double buffer)."""
    from oracle import opacus_ref as O
    g = torch.Generator().manual_seed(B + T)
    A = torch.randn(B, T, in_f, generator=g)
    G = torch.randn(B, T, out_f, generator=g) * 0.01
    gw_ref, gb_ref = O.linear_grad_sample(A, G * float(B), True)
    ld = (in_f + 31) // 32 * 32
    nw = ld * out_f
    base = torch.full((B, 64 + nw + out_f + 64), 3.0, device="cuda")
    slots = nat.linear_slots(in_f, out_f)
    part = torch.zeros(B, slots, device="cuda")
    nat.linear_grad_sample(A.cuda(), G.cuda(), nat.FlatView(base, 64, nw), nat.FlatView(base, 64 + nw, out_f), part, 0,
                           float(B), ld_w=ld)
    out = base.cpu()
    w = out[:, 64:64 + nw].reshape(B, out_f, ld)
    assert rel_err(w[:, :, :in_f].numpy(), gw_ref.numpy()) < TOL
    assert (w[:, :, in_f:] == 0).all()
    assert rel_err(out[:, 64 + nw:64 + nw + out_f].numpy(), gb_ref.numpy()) < TOL
    assert (out[:, :64] == 3).all() and (out[:, 64 + nw + out_f:] == 3).all()
    ref_sq = (gw_ref.double() ** 2).sum((1, 2)) + (gb_ref.double() ** 2).sum(1)
    assert rel_err(part.cpu().double().sum(1).numpy(), ref_sq.numpy()) < TOL


@pytest.mark.parametrize("B,T,in_f,out_f,n", [(64, 24, 256, 128, 6), (9, 24, 400, 200, 3), (5, 24, 32, 128, 4),
                                               (3, 70, 300, 140, 2), (130, 48, 256, 250, 8)])
def test_linear_grad_sample_group_equals_single_launches(nat, B, T, in_f, out_f, n):
    """dpk_linear_grad_sample_group == n dpk_linear_grad_sample calls, bit for bit (also on the plans and sequence
    lengths where the library falls back to one launch per job)."""
    g = torch.Generator().manual_seed(n)
    nw = in_f * out_f
    width = n * (nw + out_f + 64)
    slots = nat.linear_slots(in_f, out_f)
    As = [torch.randn(B, T, in_f, generator=g).cuda() for _ in range(n)]
    Gs = [torch.randn(B, T, out_f, generator=g).cuda() for _ in range(n)]
    outs = []
    for grouped in (False, True):
        base, off = _arena(B, width)
        part = torch.full((B, n * slots + 1), -1.0, device="cuda")
        jobs = []
        for j in range(n):
            o = off + j * (nw + out_f + 64)
            vb = nat.FlatView(base, o + nw, out_f) if j % 2 == 0 else None  # mixed bias / no bias
            jobs.append((As[j], Gs[j], nat.FlatView(base, o, nw), vb, 1 + j * slots))
        if grouped:
            nat.linear_grad_sample_group(jobs, part, float(B))
        else:
            for A, G, vw, vb, s0 in jobs:
                nat.linear_grad_sample(A, G, vw, vb, part, s0, float(B))
        outs.append((base.cpu(), part.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert (outs[1][1][:, 0] == -1).all() and (outs[1][1][:, 1:] >= 0).all()


@pytest.mark.parametrize("B,T,in_f,out_f", [(20, 24, 256, 128), (6, 24, 16, 40), (5, 60, 300, 136)])
def test_linear_grad_sample_reads_column_blocks_of_a_wider_backprop_buffer(nat, B, T, in_f, out_f):
    """G operands that are column blocks of one [B,T,3*out] buffer (ld_g = 3*out, as the fused GRU backward writes
    them) give bit-identical results to contiguous copies."""
    g = torch.Generator().manual_seed(7)
    A = torch.randn(B, T, in_f, generator=g).cuda()
    G3 = torch.randn(B, T, 3 * out_f, generator=g).cuda()
    nw = in_f * out_f
    slots = nat.linear_slots(in_f, out_f)
    outs = []
    for strided in (False, True):
        base, off = _arena(B, 3 * (nw + out_f))
        part = torch.zeros(B, 3 * slots, device="cuda")
        jobs = []
        for k in range(3):
            Gk = G3[..., k * out_f:(k + 1) * out_f]
            o = off + k * (nw + out_f)
            jobs.append((A, Gk if strided else Gk.contiguous(), nat.FlatView(base, o, nw),
                         nat.FlatView(base, o + nw, out_f), k * slots))
        if strided:
            nat.linear_grad_sample_group(jobs[:2], part, float(B))
            nat.linear_grad_sample(*jobs[2][:4], part, jobs[2][4], float(B))  # single-call form of a strided G
        else:
            for A_, G_, vw, vb, s0 in jobs:
                nat.linear_grad_sample(A_, G_, vw, vb, part, s0, float(B))
        outs.append((base.cpu(), part.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("case", ["vae_small_mixed.npz", "vae_small_nums.npz", "denoiser_small.npz"])
def test_stacked_calls_match_reference_autograd(nat, golden, case):
    """T calls of one Linear stacked into one K=T contraction == the reference's per-sample grads."""
    g = golden(case)
    B = g.meta["B"]
    n = 0
    for lname, ncalls in g.layers.items():
        if not g.has(f"gs/{lname}.weight") or g[f"A/{lname}/0"].dtype.kind == "i" or "layer_norm" in lname:
            continue
        As = [torch.from_numpy(g[f"A/{lname}/{k}"]) for k in range(ncalls)]
        Gs = [torch.from_numpy(g[f"G/{lname}/{k}"]) for k in range(ncalls)]
        if As[0].dim() == 2:
            A = torch.stack(As, 1)
            G = torch.stack(Gs, 1)
        else:
            A = torch.cat(As, 1)
            G = torch.cat(Gs, 1)
        in_f, out_f = A.shape[-1], G.shape[-1]
        base, off = _arena(B, in_f * out_f + out_f)
        part = torch.zeros(B, nat.linear_slots(in_f, out_f), device="cuda")
        nat.linear_grad_sample(A.contiguous().cuda(), G.contiguous().cuda(), nat.FlatView(base, off, in_f * out_f),
                               nat.FlatView(base, off + in_f * out_f, out_f), part, 0, float(B))
        out = base.cpu().numpy()
        assert rel_err(out[:, off:off + in_f * out_f].reshape(B, out_f, in_f), g[f"gs/{lname}.weight"]) < TOL, lname
        assert rel_err(out[:, off + in_f * out_f:off + in_f * out_f + out_f], g[f"gs/{lname}.bias"]) < TOL, lname
        n += 1
    assert n >= 7


@pytest.mark.parametrize("B,T,V,E", [(8, 6, 5, 8), (4, 24, 48, 128), (3, 24, 3, 128), (5, 1, 2, 7), (2, 300, 18, 33)])
def test_embedding_grad_sample(nat, B, T, V, E):
    from oracle import opacus_ref as O
    g = torch.Generator().manual_seed(B + T + V + E)
    idx = torch.randint(0, V, (B, T), generator=g)
    G = torch.randn(B, T, E, generator=g)
    ref = O.embedding_grad_sample(idx, G * float(B), V)
    base, off = _arena(B, V * E)
    part = torch.zeros(B, 1, device="cuda")
    nat.embedding_grad_sample(idx.cuda(), G.cuda(), nat.FlatView(base, off, V * E), V, part, 0, float(B))
    out = base.cpu()
    assert rel_err(out[:, off:off + V * E].reshape(B, V, E).numpy(), ref.numpy()) < TOL
    assert (out[:, :off] == 0).all() and (out[:, off + V * E:] == 0).all()
    assert rel_err(part.cpu()[:, 0].numpy(), (ref.double() ** 2).sum((1, 2)).numpy()) < TOL


def test_embedding_golden(nat, golden):
    g = golden("vae_small_mixed.npz")
    B = g.meta["B"]
    for k in range(3):
        lname = f"Emb.embeddings_list.{k}"
        idx = torch.from_numpy(g[f"A/{lname}/0"])
        G = torch.from_numpy(g[f"G/{lname}/0"])
        ref = g[f"gs/{lname}.weight"]
        V, E = ref.shape[1:]
        base, off = _arena(B, V * E)
        part = torch.zeros(B, 1, device="cuda")
        nat.embedding_grad_sample(idx.cuda(), G.cuda(), nat.FlatView(base, off, V * E), V, part, 0, float(B))
        assert rel_err(base.cpu()[:, off:off + V * E].reshape(B, V, E).numpy(), ref) < TOL


@pytest.mark.parametrize("B,T,C", [(6, 5, 16), (4, 48, 500), (3, 300, 70), (2, 1, 1000)])
def test_layernorm_grad_sample(nat, B, T, C):
    from oracle import opacus_ref as O
    g = torch.Generator().manual_seed(B + T + C)
    X = torch.randn(B, T, C, generator=g) * 2 + 0.5
    G = torch.randn(B, T, C, generator=g)
    gw_ref, gb_ref = O.layernorm_grad_sample(X, G * float(B), (C,), 1e-5)
    base, off = _arena(B, 2 * C)
    part = torch.zeros(B, 1, device="cuda")
    nat.layernorm_grad_sample(X.cuda(), G.cuda(), nat.FlatView(base, off, C), nat.FlatView(base, off + C, C), part, 0,
                              1e-5, float(B))
    out = base.cpu()
    assert rel_err(out[:, off:off + C].numpy(), gw_ref.numpy()) < TOL
    assert rel_err(out[:, off + C:off + 2 * C].numpy(), gb_ref.numpy()) < TOL
    ref_sq = (gw_ref.double() ** 2).sum(1) + (gb_ref.double() ** 2).sum(1)
    assert rel_err(part.cpu()[:, 0].numpy(), ref_sq.numpy()) < TOL


def test_layernorm_golden(nat, golden):
    g = golden("denoiser_small.npz")
    B = g.meta["B"]
    X = torch.from_numpy(g["A/layer_norm/0"])
    G = torch.from_numpy(g["G/layer_norm/0"])
    C = X.shape[-1]
    base, off = _arena(B, 2 * C)
    part = torch.zeros(B, 1, device="cuda")
    nat.layernorm_grad_sample(X.cuda(), G.cuda(), nat.FlatView(base, off, C), nat.FlatView(base, off + C, C), part, 0,
                              1e-5, float(B))
    out = base.cpu().numpy()
    assert rel_err(out[:, off:off + C], g["gs/layer_norm.weight"]) < TOL
    assert rel_err(out[:, off + C:off + 2 * C], g["gs/layer_norm.bias"]) < TOL


@pytest.mark.parametrize("B,P", [(5, 100), (3, 8192), (4, 20001), (2, 649400)])
def test_normsq_partials_and_clip_factors(nat, B, P):
    g = torch.Generator().manual_seed(P)
    GS = torch.randn(B, P, generator=g) * 0.01
    Pp = (P + 3) // 4 * 4
    base = torch.zeros(B, Pp + 8, device="cuda")
    base[:, 4:4 + P] = GS.cuda()
    slots = nat.normsq_slots(P)
    part = torch.zeros(B, slots, device="cuda")
    nat.normsq_partials(nat.FlatView(base, 4, P), B, part, 0)
    clip = torch.zeros(B, device="cuda")
    norms = torch.zeros(B, device="cuda")
    ref_n = GS.double().norm(2, dim=1)
    C = float(ref_n.median())
    nat.clip_factors(part, slots, clip, norms, B, C)
    assert rel_err(norms.cpu().numpy(), ref_n.numpy()) < TOL
    ref_c = (C / (ref_n + 1e-6)).clamp(max=1.0)
    assert rel_err(clip.cpu().numpy(), ref_c.numpy()) < TOL


@pytest.mark.parametrize("B,P", [(8, 64), (64, 4096), (50, 649400), (200, 10000), (1000, 2048), (1, 12), (33, 100000)])
@pytest.mark.parametrize("fused", [True, False])
def test_clip_reduce_noise(nat, B, P, fused):
    from oracle import opacus_ref as O
    g = torch.Generator().manual_seed(B * 31 + P)
    GS = torch.randn(B, P, generator=g)
    clip = torch.rand(B, generator=g)
    summed_in = torch.randn(P, generator=g)
    noise = torch.randn(P, generator=g) * 0.3
    stride = P + 16
    base = torch.zeros(B, stride, device="cuda")
    base[:, 8:8 + P] = GS.cuda()
    ws_bytes = nat.clip_reduce_workspace(B, P)
    ws = torch.empty(max(ws_bytes // 4, 4), device="cuda") if ws_bytes else None
    view = nat.FlatView(base, 8, P)
    ebs = 49.0
    ref_sum = torch.einsum("i,ij->j", clip.double(), GS.double()) + summed_in.double()
    ref_grad = (ref_sum + noise.double()) / ebs
    summed = torch.zeros(P, device="cuda")
    grad = torch.zeros(P, device="cuda")
    if fused:
        nat.clip_reduce_noise(view, clip.cuda(), summed_in.cuda(), noise.cuda(), 0, 0, 0.0, 1.0 / ebs, summed, grad, B,
                              ws)
    else:
        nat.clip_reduce(view, clip.cuda(), summed_in.cuda(), summed, B, ws)
        nat.noise_scale(summed, noise.cuda(), 0, 0, 0.0, 1.0 / ebs, grad, P)
    scale = np.abs(GS.numpy()).max() * np.sqrt(B)
    assert np.abs(summed.cpu().double().numpy() - ref_sum.numpy()).max() / max(np.abs(ref_sum.numpy()).max(), 1e-30) < TOL
    assert rel_err(grad.cpu().numpy(), ref_grad.numpy()) < TOL


def test_clip_reduce_in_place_accumulate_matches_two_chunks(nat):
    """BatchMemoryManager protocol: summed_grad accumulated over physical chunks == one pass."""
    g = torch.Generator().manual_seed(3)
    B, P = 96, 20000
    GS = torch.randn(B, P, generator=g).cuda()
    clip = torch.rand(B, generator=g).cuda()
    ws = torch.empty(max(nat.clip_reduce_workspace(B, P) // 4, 4), device="cuda")
    one = torch.zeros(P, device="cuda")
    nat.clip_reduce(nat.FlatView(GS, 0, P), clip, None, one, B, ws)
    two = torch.zeros(P, device="cuda")
    h = 40
    nat.clip_reduce(nat.FlatView(GS[:h].contiguous(), 0, P), clip[:h].contiguous(), None, two, h, ws)
    nat.clip_reduce(nat.FlatView(GS[h:].contiguous(), 0, P), clip[h:].contiguous(), two, two, B - h, ws)
    assert rel_err(two.cpu().numpy(), one.cpu().numpy()) < TOL


def test_philox_stream_matches_oracle(nat):
    from oracle import opacus_ref as O
    n = 100003
    out = torch.zeros(n, device="cuda")
    nat.philox_normal(out, seed=0x1234567890ABCDEF, offset=77, std=1.5)
    ref = O.philox_normal(n, 0x1234567890ABCDEF, 77, 1.5)
    got = out.cpu().numpy()
    # integer pipeline is bit-exact; float transform (log/sincos) may differ by a few ulp
    assert np.abs(got - ref).max() < 2e-5
    assert abs(got.mean()) < 0.02 and abs(got.std() - 1.5) < 0.02


def test_noise_is_identical_for_same_seed_and_offset(nat):
    a = torch.zeros(4096, device="cuda")
    b = torch.zeros(4096, device="cuda")
    nat.philox_normal(a, 9, 5, 1.0)
    nat.philox_normal(b, 9, 5, 1.0)
    assert torch.equal(a, b)
    nat.philox_normal(b, 9, 6, 1.0)
    assert torch.equal(a[4:], b[:-4])


def test_argument_errors_launch_nothing(nat):
    base = torch.zeros(2, 64, device="cuda")
    part = torch.zeros(2, 1, device="cuda")
    A = torch.zeros(2, 3, 4, device="cuda")
    G = torch.zeros(2, 3, 5, device="cuda")
    with pytest.raises(nat.DpkError):
        nat.linear_grad_sample(A, G, nat.FlatView(base, 0, 19), None, part, 0, 1.0)  # wrong window
    with pytest.raises(nat.DpkError):
        nat.linear_grad_sample(A.cpu(), G, nat.FlatView(base, 0, 20), None, part, 0, 1.0)  # CPU tensor
    with pytest.raises(nat.DpkError):
        nat.linear_grad_sample(A, G, nat.FlatView(base, 0, 20), None, part, 1, 1.0)  # slot overflow


# ---- K1 on the ARENA layout (what GradSampleArena hands it): windows on 256-byte boundaries, row stride a multiple of
# 32 floats, rows of awkward width padded — the whole-line path (persistent / software-pipelined kernels).  The shapes
# above use odd strides on purpose and land on the generic kernels; without these the fast path would only be covered
# through the engine tests.
ARENA_SHAPES = [
    # (B, T, in, out, n_jobs)
    (9, 48, 250, 1000, 1), (5, 48, 500, 1000, 1), (6, 48, 1000, 250, 1), (11, 48, 250, 250, 1),   # cfg3 layers
    (70, 24, 512, 512, 1), (33, 24, 256, 128, 6), (17, 24, 400, 200, 6), (40, 24, 208, 208, 1),   # cfg2 / cfg1 layers
    (3, 1, 130, 33, 1), (4, 31, 300, 260, 2), (2, 47, 129, 129, 3), (300, 12, 160, 40, 1), (1, 48, 512, 97, 1),
    (13, 32, 384, 70, 1), (5, 33, 640, 64, 1),
    # beyond 32 steps the large layers run as 32-step chunks on the split-operand kernel (cfg5: T = 256)
    (3, 256, 512, 512, 1), (4, 70, 300, 260, 2), (2, 100, 129, 129, 3), (5, 49, 250, 250, 1), (2, 33, 640, 140, 1),
    (3, 64, 256, 128, 1), (2, 40, 1024, 40, 1), (9, 36, 130, 33, 1),
]


def _product_ld(in_f):
    """Row stride the product's arena gives a Linear weight (grad_sample._row_stride): >= 64 columns -> multiple of 32,
    narrower -> multiple of 4 (13 -> 16, 7 -> 8, 1 -> 4), so that every row can leave as 16-byte stores."""
    from dp_timeautodiff_amd.grad_sample import _row_stride
    return _row_stride(torch.nn.Parameter(torch.empty(1, in_f)))


def _arena_case(nat, B, T, in_f, out_f, nj, bias, seed, ld=None):
    from oracle import opacus_ref as O
    g = torch.Generator().manual_seed(seed)
    if ld is None:
        ld = (in_f + 31) // 32 * 32 if (in_f >= 64 and in_f % 32) else in_f
    win = (ld * out_f + 63) // 64 * 64
    bwin = (out_f + 63) // 64 * 64
    row = 64 + nj * (win + bwin) + 64
    assert row % 32 == 0
    base = torch.full((B, row), 5.0, device="cuda")
    assert base.data_ptr() % 256 == 0
    slots = nat.linear_slots(in_f, out_f)
    part = torch.full((B, nj * slots + 3), -1.0, device="cuda")
    As = [torch.randn(B, T, in_f, generator=g) for _ in range(nj)]
    Gs = [torch.randn(B, T, out_f, generator=g) * 0.01 for _ in range(nj)]
    jobs = []
    for j in range(nj):
        off = 64 + j * (win + bwin)
        jobs.append((As[j].cuda(), Gs[j].cuda(), nat.FlatView(base, off, ld * out_f),
                     nat.FlatView(base, off + win, out_f) if bias else None, 1 + j * slots))
    if nj == 1:
        a, gg, vw, vb, s0 = jobs[0]
        nat.linear_grad_sample(a, gg, vw, vb, part, s0, float(B), ld_w=ld)
    else:
        nat.linear_grad_sample_group(jobs, part, float(B), ld_w=ld)
    out, p = base.cpu(), part.cpu()
    touched = torch.zeros(row, dtype=torch.bool)
    for j in range(nj):
        off = 64 + j * (win + bwin)
        gw_ref, gb_ref = O.linear_grad_sample(As[j], Gs[j] * float(B), True)
        w = out[:, off:off + ld * out_f].reshape(B, out_f, ld)
        assert rel_err(w[:, :, :in_f].numpy(), gw_ref.numpy()) < TOL, (j, "weight")
        assert (w[:, :, in_f:] == 0).all(), "row padding must be exact zeros"
        # row-wise too: a sample with a small gradient must not hide behind one with a large gradient
        num = (w[:, :, :in_f] - gw_ref).abs().amax((1, 2))
        assert float((num / gw_ref.abs().amax((1, 2)).clamp_min(1e-30)).max()) < TOL
        ref_sq = (gw_ref.double() ** 2).sum((1, 2))
        touched[off:off + ld * out_f] = True
        if bias:
            assert rel_err(out[:, off + win:off + win + out_f].numpy(), gb_ref.numpy()) < TOL, (j, "bias")
            ref_sq = ref_sq + (gb_ref.double() ** 2).sum(1)
            touched[off + win:off + win + out_f] = True
        got_sq = p[:, 1 + j * slots:1 + (j + 1) * slots].double().sum(1)
        assert rel_err(got_sq.numpy(), ref_sq.numpy()) < TOL, (j, "normsq")
    assert (out[:, ~touched] == 5.0).all(), "wrote outside the layer's windows"
    assert (p[:, 0] == -1).all() and (p[:, 1 + nj * slots:] == -1).all()


@pytest.mark.parametrize("B,T,in_f,out_f,nj", ARENA_SHAPES)
@pytest.mark.parametrize("bias", [True, False])
def test_linear_grad_sample_arena_layout(nat, B, T, in_f, out_f, nj, bias):
    _arena_case(nat, B, T, in_f, out_f, nj, bias, seed=B * 1000 + T)


# narrow layers at the row stride the PRODUCT's arena gives them (the cases above keep in < 64 unpadded, the
# arena pads to 4 floats and relies on exact zeros in [in, ld_w): K7 and the all-reduce read those columns)
NARROW_SHAPES = [
    # (B, T, in, out, n_jobs)
    (64, 24, 13, 200, 2), (50, 24, 7, 200, 1), (9, 48, 1, 125, 2), (12, 24, 33, 130, 1), (5, 48, 8, 250, 1),
    (4096, 48, 7, 250, 1), (33, 256, 13, 512, 1), (7, 24, 62, 7, 1), (20, 40, 5, 33, 3),
]


@pytest.mark.parametrize("B,T,in_f,out_f,nj", NARROW_SHAPES)
@pytest.mark.parametrize("bias", [True, False])
def test_linear_grad_sample_narrow_rows_at_the_arena_stride(nat, B, T, in_f, out_f, nj, bias):
    ld = _product_ld(in_f)
    assert ld % 4 == 0 and in_f <= ld < in_f + 4
    _arena_case(nat, B, T, in_f, out_f, nj, bias, seed=B + 7 * T + in_f, ld=ld)  # asserts zero padding, untouched neighbours


def test_linear_grad_sample_arena_layout_many_items_is_deterministic(nat):
    """Far more (sample, tile) items than workgroups: every workgroup's pipeline runs for many stages; twice, bit for bit."""
    B, T, in_f, out_f = 700, 48, 250, 500
    ld, win = 256, 256 * 500
    g = torch.Generator().manual_seed(1)
    A = torch.randn(B, T, in_f, generator=g).cuda()
    G = torch.randn(B, T, out_f, generator=g).cuda()
    outs = []
    for _ in range(2):
        base = torch.zeros(B, win + 512, device="cuda")
        part = torch.zeros(B, nat.linear_slots(in_f, out_f), device="cuda")
        nat.linear_grad_sample(A, G, nat.FlatView(base, 0, win), nat.FlatView(base, win, out_f), part, 0, float(B), ld_w=ld)
        outs.append((base.cpu(), part.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = torch.einsum("bto,bti->boi", G.double() * B, A.double()).cpu()
    w = outs[0][0][:, :win].reshape(B, out_f, ld)[:, :, :in_f].double()
    err = (w - ref).abs().amax((1, 2)) / ref.abs().amax((1, 2))
    assert float(err.max()) < TOL


@pytest.mark.parametrize("B,T,in_f,out_f", [(6, 48, 300, 260), (4, 24, 512, 512), (3, 256, 512, 384), (5, 70, 250, 1000)])
def test_linear_grad_sample_arithmetic_modes_are_fp32_accurate(nat, monkeypatch, B, T, in_f, out_f):
    """The large-layer kernels compute fp32 products either on the fp32 matrix pipe (DPK_K1_MFMA=f32) or as 6 / 8 / 9
    of the 9 exact partial products of a three-way bf16 split (split6 / split8 / split9; k1_linear_gs.hip).  Against a
    float64 contraction of the same fp32 inputs every mode must be an fp32-grade result: error below 1e-6 of the
    sample's largest entry (the path's tolerance is 1e-5), and dropping partial products (split6) must cost no more
    than the accumulation rounding that all modes share."""
    g = torch.Generator().manual_seed(T + in_f)
    A = (torch.randn(B, T, in_f, generator=g) * torch.logspace(-3, 3, in_f)).cuda()   # 6 decades across columns
    G = (torch.randn(B, T, out_f, generator=g) * 0.01).cuda()
    ld = (in_f + 31) // 32 * 32
    ref = torch.einsum("bto,bti->boi", G.double() * B, A.double())
    ref_b = (G.double() * B).sum(1)
    errs = {}
    for mode in ("f32", "split6", "split8", "split9", None):
        nat.linear_configure(arithmetic=mode)
        base = torch.zeros(B, ld * out_f + 64 * ((out_f + 63) // 64), device="cuda")
        part = torch.zeros(B, nat.linear_slots(in_f, out_f), device="cuda")
        nat.linear_grad_sample(A, G, nat.FlatView(base, 0, ld * out_f), nat.FlatView(base, ld * out_f, out_f), part, 0,
                               float(B), ld_w=ld)
        w = base[:, :ld * out_f].reshape(B, out_f, ld)[:, :, :in_f].double()
        # column by column: every column judged against its own scale (the columns span six decades)
        err = ((w - ref).abs().amax(1) / ref.abs().amax(1)).max()
        errs[mode] = float(err)
        assert float(err) < 1e-6, (mode, float(err))
        assert float(((base[:, ld * out_f:ld * out_f + out_f].double() - ref_b).abs() / ref_b.abs().amax()).max()) < 1e-6
        sq = (ref ** 2).sum((1, 2)) + (ref_b ** 2).sum(1)
        assert float((part.double().sum(1) / sq - 1).abs().max()) < 1e-6, mode
    assert errs["split6"] <= 2.0 * errs["split9"] + 2e-8, errs
    assert errs["split8"] <= 1.25 * errs["split9"] + 1e-8, errs


MULTI_LAYERS = [
    # (T, in, out, bias) — the cfg1 VAE's layers with 16-byte-storable rows, plus odd ones
    (24, 400, 200, True), (24, 400, 200, True), (24, 208, 208, True), (24, 208, 128, True), (24, 200, 200, True),
    (24, 128, 13, True), (24, 200, 7, True), (24, 200, 13, False), (1, 130, 33, True), (31, 300, 260, True),
    (48, 512, 97, True), (24, 32, 128, True), (7, 64, 300, False), (24, 400, 200, True), (24, 400, 200, True),
    (24, 400, 200, True), (24, 400, 200, True), (12, 160, 40, True),
    # narrow inputs at the arena's padded stride (13 -> 16, 7 -> 8, 1 -> 4, 33 -> 36): dead column groups, zero padding
    (24, 13, 200, True), (24, 7, 200, True), (24, 1, 125, False), (24, 33, 40, True),
]


@pytest.mark.parametrize("B,max_jobs", [(50, None), (7, None), (130, None), (50, 5)])
def test_linear_grad_sample_multi_equals_single_launches(nat, monkeypatch, B, max_jobs):
    """dpk_linear_grad_sample_multi (layers of different shapes in one launch: the small-batch path) against one
    dpk_linear_grad_sample call per layer on the fp32 matrix pipe: weights bit for bit for every layer; bias and
    squared-norm partials bit for bit for layers of the large plan (thinner layers sum the bias in another order)."""
    nat.linear_configure(arithmetic="f32")
    if max_jobs:  # the host wrapper cuts longer job lists into several launches
        monkeypatch.setattr(nat, "K1_MULTI_MAX_JOBS", max_jobs)
    g = torch.Generator().manual_seed(B)
    lds = [_product_ld(i) for _, i, _, _ in MULTI_LAYERS]
    wins = [(ld * o + 63) // 64 * 64 for ld, (_, _, o, _) in zip(lds, MULTI_LAYERS)]
    bw = [(o + 63) // 64 * 64 for _, _, o, _ in MULTI_LAYERS]
    offs, off = [], 0
    for w, b in zip(wins, bw):
        offs.append(off)
        off += w + b
    row = off
    slots = [nat.linear_slots(i, o) for _, i, o, _ in MULTI_LAYERS]
    s0 = [sum(slots[:k]) for k in range(len(slots))]
    As = [torch.randn(B, T, i, generator=g).cuda() for T, i, _, _ in MULTI_LAYERS]
    Gs = [(torch.randn(B, T, o, generator=g) * 0.01).cuda() for T, _, o, _ in MULTI_LAYERS]

    def views(base, k):
        _, _, o, bias = MULTI_LAYERS[k]
        return (nat.FlatView(base, offs[k], lds[k] * o), nat.FlatView(base, offs[k] + wins[k], o) if bias else None)

    one = torch.full((B, row), 3.0, device="cuda")
    p_one = torch.full((B, sum(slots) + 2), -1.0, device="cuda")
    for k in range(len(MULTI_LAYERS)):
        vw, vb = views(one, k)
        nat.linear_grad_sample(As[k], Gs[k], vw, vb, p_one, s0[k], float(B), ld_w=lds[k])
    many = torch.full((B, row), 3.0, device="cuda")
    p_many = torch.full((B, sum(slots) + 2), -1.0, device="cuda")
    assert all(nat.multi_ok(As[k], Gs[k], lds[k]) for k in range(len(MULTI_LAYERS)))
    nat.linear_grad_sample_multi([(As[k], Gs[k], *views(many, k), s0[k], lds[k]) for k in range(len(MULTI_LAYERS))],
                                 p_many, float(B))
    torch.cuda.synchronize()
    for k, (T, i, o, bias) in enumerate(MULTI_LAYERS):
        w1 = one[:, offs[k]:offs[k] + lds[k] * o]
        w2 = many[:, offs[k]:offs[k] + lds[k] * o]
        assert torch.equal(w1, w2), (k, "weight")
        large = i > 128 and o > 32
        if bias:
            b1 = one[:, offs[k] + wins[k]:offs[k] + wins[k] + o]
            b2 = many[:, offs[k] + wins[k]:offs[k] + wins[k] + o]
            assert torch.equal(b1, b2) if large else rel_err(b2.cpu().numpy(), b1.cpu().numpy()) < 1e-6, (k, "bias")
        q1 = p_one[:, s0[k]:s0[k] + slots[k]]
        q2 = p_many[:, s0[k]:s0[k] + slots[k]]
        if large:
            assert torch.equal(q1, q2), (k, "normsq")
        else:
            assert rel_err(q2.double().sum(1).cpu().numpy(), q1.double().sum(1).cpu().numpy()) < 1e-6, (k, "normsq")
    assert (p_many[:, sum(slots):] == -1).all()
    # nothing outside the windows was touched
    mask = torch.ones(row, dtype=torch.bool)
    for k, (T, i, o, bias) in enumerate(MULTI_LAYERS):
        mask[offs[k]:offs[k] + lds[k] * o] = False
        if bias:
            mask[offs[k] + wins[k]:offs[k] + wins[k] + o] = False
    assert (many[:, mask.cuda()] == 3.0).all()
    for k, (T, i, o, bias) in enumerate(MULTI_LAYERS):   # row padding [in, ld) is exact zeros in both forms
        for buf in (one, many):
            assert not buf[:, offs[k]:offs[k] + lds[k] * o].reshape(B, o, lds[k])[:, :, i:].any(), (k, "padding")


def test_linear_grad_sample_multi_refuses_what_it_cannot_store(nat):
    A = torch.randn(4, 24, 13, device="cuda")
    G = torch.randn(4, 24, 200, device="cuda")
    base = torch.zeros(4, 13 * 200 + 256, device="cuda")
    part = torch.zeros(4, 8, device="cuda")
    assert not nat.multi_ok(A, G, 13)
    with pytest.raises(nat.DpkError):
        nat.linear_grad_sample_multi([(A, G, nat.FlatView(base, 0, 13 * 200), None, 0, 13)], part, 4.0)


def test_linear_grad_sample_random_shapes(nat):
    """Seeded sweep over shapes no hand-written list would contain: every kernel choice (thin plans, wide-unit, split
    with and without the resident A slice, chunked long sequences), ragged tiles, padded and unpadded rows, groups."""
    rs = np.random.RandomState(20261111)
    for case in range(48):
        T = int(rs.choice([1, 2, 3, 7, 16, 17, 24, 31, 32, 33, 40, 47, 48, 49, 63, 64, 65, 90, 130]))
        in_f = int(rs.choice([1, 5, 13, 31, 64, 100, 128, 129, 200, 250, 257, 300, 512, 640]))
        out_f = int(rs.choice([1, 7, 32, 33, 64, 100, 127, 128, 129, 200, 260]))
        B = int(rs.randint(1, 20))
        nj = int(rs.choice([1, 1, 1, 2, 3]))
        bias = bool(rs.randint(0, 2))
        _arena_case(nat, B, T, in_f, out_f, nj, bias, seed=1000 + case)


ROLE_SPLIT_CASES = [
    # (B, T, in, out, n_jobs, ld_g_extra)  — 32 < T <= 48: linear_gs_ws_kernel
    (37, 48, 250, 1000, 1, 0),     # the denoiser's ih/hh shape; B not a multiple of 8; ragged last row / column tile
    (16, 48, 500, 1000, 1, 0),
    (9, 33, 300, 385, 1, 0),       # T just above 32: steps 33..47 of the k-step padding are zeros; 4 row tiles, last of 1 row
    (21, 41, 129, 512, 3, 1024),   # three jobs whose G are column blocks of one [B, T, 3*512] tensor (K8/K10 hand-over)
    (300, 48, 256, 384, 2, 0),     # more (sample, column tile) groups than workgroups: every workgroup walks several
    (19, 48, 1000, 250, 1, 0),     # fewer than 3 row tiles: the G-resident form (tile group = column tiles of a row tile)
    (70, 40, 400, 130, 2, 0),      # the same with a one-row-tile-and-a-bit layer, two jobs, T < 48
    (300, 48, 640, 128, 1, 0),     # G-resident, ONE row tile, more groups than workgroups
    (23, 48, 250, 250, 1, 0),      # 2 x 2 tiles: groups of two, a group change after every second tile
    # T > 48: linear_gs_wsl_kernel (256 x 128 super-tiles, one 16-step k-step per phase) against MODE 1
    (9, 256, 256, 256, 1, 0),      # the long-sequence VAE's T; one super-tile of two live halves per column tile
    (11, 80, 200, 300, 1, 0),      # 3 row tiles: the last super-tile has one live half; ragged rows and columns
    (5, 49, 130, 129, 2, 0),       # T = 3 k-steps + 1 step; two jobs sharing one G tensor; 2nd row tile of 1 row
    (300, 50, 128, 100, 1, 0),     # one row tile; more super-tiles than workgroups: tile changes inside a workgroup
    (3, 100, 1000, 40, 1, 7),      # many column tiles per sample, G rows wider than out_f
]


@pytest.mark.parametrize("B,T,in_f,out_f,nj,ldx", ROLE_SPLIT_CASES)
def test_role_split_kernel_equals_split_kernel_bit_for_bit(nat, B, T, in_f, out_f, nj, ldx):
    """linear_gs_ws_kernel (loader waves + two contractor groups, round 3) against linear_gs_split_kernel MODE 2 (every
    wave does everything, round 2): same arithmetic per output element, so weights, bias, row padding and squared-norm
    partials must agree bit for bit — and both against the float64 contraction at the fp32 bar."""
    g = torch.Generator().manual_seed(B * 1000 + T)
    ld_g = out_f + ldx if nj == 1 else out_f * nj
    Gall = (torch.randn(B, T, ld_g if nj > 1 else out_f + ldx, generator=g) * 0.05).cuda()
    As = [torch.randn(B, T, in_f, generator=g).cuda() for _ in range(nj)]
    ld = (in_f + 31) // 32 * 32 if in_f % 32 else in_f
    win = (ld * out_f + 63) // 64 * 64
    bw = (out_f + 63) // 64 * 64
    row = nj * (win + bw) + 64
    slots = nat.linear_slots(in_f, out_f)
    outs = {}
    for split in (False, True):
        nat.linear_configure(arithmetic="split6", role_split=split)
        base = torch.full((B, row), 7.0, device="cuda")
        part = torch.full((B, nj * slots + 3), -1.0, device="cuda")
        jobs = []
        for j in range(nj):
            Gj = Gall[:, :, j * out_f:(j + 1) * out_f] if nj > 1 else Gall[:, :, :out_f]
            jobs.append((As[j], Gj, nat.FlatView(base, j * (win + bw), ld * out_f),
                         nat.FlatView(base, j * (win + bw) + win, out_f), j * slots))
        nat.linear_grad_sample_group(jobs, part, float(B), ld_w=ld)
        torch.cuda.synchronize()
        outs[split] = (base, part)
    assert torch.equal(outs[True][0], outs[False][0])
    assert torch.equal(outs[True][1], outs[False][1])
    base, part = outs[True]
    assert float(base[:, -64:].min()) == 7.0 and float(part[:, -3:].max()) == -1.0      # nothing outside the windows
    for j in range(nj):
        Gj = (Gall[:, :, j * out_f:(j + 1) * out_f] if nj > 1 else Gall[:, :, :out_f]).double() * B
        ref = torch.einsum("bto,bti->boi", Gj, As[j].double())
        w = base[:, j * (win + bw):j * (win + bw) + ld * out_f].reshape(B, out_f, ld)
        assert not w[:, :, in_f:].any()                                                # row padding: exact zeros
        assert float(((w[:, :, :in_f].double() - ref).abs().amax((1, 2)) / ref.abs().amax((1, 2))).max()) < 1e-6
        rb = Gj.sum(1)
        b = base[:, j * (win + bw) + win:j * (win + bw) + win + out_f].double()
        assert float(((b - rb).abs().amax(1) / rb.abs().amax(1)).max()) < 1e-6
        sq = (ref ** 2).sum((1, 2)) + (rb ** 2).sum(1)
        assert float((part[:, j * slots:(j + 1) * slots].double().sum(1) / sq - 1).abs().max()) < 1e-6


def test_round3_kernels_equal_the_split_kernel_on_random_shapes(nat):
    """A seeded sweep over sequence lengths on both sides of 48, ragged feature counts, odd tile counts and batch sizes
    that do not fill the XCDs: linear_gs_ws_kernel / linear_gs_wsl_kernel against linear_gs_split_kernel, bit for bit
    (weights, bias, row padding, squared-norm slots, nothing outside the windows)."""
    rng = np.random.default_rng(20240611)
    for case in range(24):
        T = int(rng.choice([33, 40, 48, 49, 64, 65, 96, 130, 200]))
        in_f = int(rng.choice([129, 200, 250, 257, 384, 500, 513]))
        out_f = int(rng.choice([33, 100, 128, 129, 250, 256, 300, 385]))
        B = int(rng.integers(1, 40))
        g = torch.Generator().manual_seed(case)
        A = torch.randn(B, T, in_f, generator=g).cuda()
        G = (torch.randn(B, T, out_f, generator=g) * 0.1).cuda()
        ld = (in_f + 31) // 32 * 32 if in_f % 32 else in_f
        win = (ld * out_f + 63) // 64 * 64
        slots = nat.linear_slots(in_f, out_f)
        outs = []
        for split in (False, True):
            nat.linear_configure(arithmetic="split6", role_split=split)
            base = torch.full((B, win + (out_f + 63) // 64 * 64 + 64), 7.0, device="cuda")
            part = torch.full((B, slots + 2), -1.0, device="cuda")
            nat.linear_grad_sample(A, G, nat.FlatView(base, 0, ld * out_f), nat.FlatView(base, win, out_f), part, 0, 1.0, ld_w=ld)
            torch.cuda.synchronize()
            outs.append((base, part))
        tag = f"case {case}: B={B} T={T} {in_f}->{out_f}"
        assert torch.equal(outs[0][0], outs[1][0]), tag
        assert torch.equal(outs[0][1], outs[1][1]), tag
        assert float(outs[1][0][:, -64:].min()) == 7.0 and float(outs[1][1][:, -2:].max()) == -1.0, tag
        ref = torch.einsum("bto,bti->boi", G.double(), A.double())
        w = outs[1][0][:, :ld * out_f].reshape(B, out_f, ld)
        assert not w[:, :, in_f:].any(), tag
        assert float(((w[:, :, :in_f].double() - ref).abs().amax((1, 2)) / ref.abs().amax((1, 2))).max()) < 1e-6, tag


# ---- special values through the bf16-split arithmetic ------------------------------------------------
def _k1_run(nat, A, G, mode, role_split, g_scale=1.0):
    B, T, in_f = A.shape
    out_f = G.shape[2]
    ld = (in_f + 31) // 32 * 32 if in_f % 32 else in_f
    nat.linear_configure(arithmetic=mode, role_split=role_split)
    base = torch.zeros(B, ld * out_f + 64 * ((out_f + 63) // 64), device="cuda")
    part = torch.zeros(B, nat.linear_slots(in_f, out_f), device="cuda")
    nat.linear_grad_sample(A, G, nat.FlatView(base, 0, ld * out_f), nat.FlatView(base, ld * out_f, out_f), part, 0,
                           g_scale, ld_w=ld)
    torch.cuda.synchronize()
    return base[:, :ld * out_f].reshape(B, out_f, ld)[:, :, :in_f].clone(), base[:, ld * out_f:ld * out_f + out_f].clone(), part


# (T, in, out): role-split kernel / split kernel MODE 2 at T = 48, the T > 48 split kernel (MODE 1) at T = 80
SPECIAL_SHAPES = [(48, 256, 384), (80, 256, 256)]


def _special_modes(T):
    modes = [("split6", True), ("split6", False), ("split9", False)]
    if T <= 48:
        modes.append(("f32", False))
    return modes


@pytest.mark.parametrize("T,in_f,out_f", SPECIAL_SHAPES)
def test_split_arithmetic_near_flt_max(nat, T, in_f, out_f):
    """Operands between the largest finite bf16 (3.3895e38) and FLT_MAX (3.4028e38): rounding the first piece to nearest
    would give bf16 infinity and NaN products; the clamped first piece keeps the split exact.  One large term per
    output element (the rest of the sequence is zero) so that the exact result is finite: 1.7e38."""
    B = 8
    big = 3.4e38
    g = torch.Generator().manual_seed(1)
    G = torch.zeros(B, T, out_f)
    A = torch.zeros(B, T, in_f)
    t0 = torch.randint(0, T, (B,), generator=g)
    sign = torch.where(torch.rand(B, out_f, generator=g) < 0.5, -1.0, 1.0)
    for b in range(B):
        G[b, t0[b]] = big * sign[b]
        A[b, t0[b]] = 0.5 * (1 + torch.arange(in_f) % 2 * -0.25)   # 0.5 / 0.375
    ref = torch.einsum("bto,bti->boi", G.double(), A.double())
    assert float(ref.abs().max()) < 3.4e38
    for mode, rs in _special_modes(T):
        w, b, part = _k1_run(nat, A.cuda(), G.cuda(), mode, rs)
        assert torch.isfinite(w).all(), (mode, rs)
        assert float(((w.double().cpu() - ref).abs() / ref.abs().clamp_min(1e30)).max()) < 1e-6, (mode, rs)
        assert float((b.double().cpu() - G.double().sum(1)).abs().max()) == 0.0   # one term per row: exact


@pytest.mark.parametrize("T,in_f,out_f", SPECIAL_SHAPES)
def test_split_arithmetic_inf_and_nan_stay_where_they_are(nat, T, in_f, out_f):
    """An infinite backprop entry poisons its own row of its own sample (non-finite: the split forms NaN where the fp32
    instruction forms +-inf), a NaN activation its own column; every other output element equals the clean run bit for
    bit in every mode."""
    B = 8
    g = torch.Generator().manual_seed(2)
    A = torch.randn(B, T, in_f, generator=g)
    G = torch.randn(B, T, out_f, generator=g) * 0.1
    Ab, Gb = A.clone(), G.clone()
    Gb[3, T // 2, 200] = float("inf")
    Gb[5, 1, 7] = float("-inf")
    Ab[6, T - 1, 129] = float("nan")
    for mode, rs in _special_modes(T):
        w0, b0, p0 = _k1_run(nat, A.cuda(), G.cuda(), mode, rs, g_scale=float(B))
        w1, b1, p1 = _k1_run(nat, Ab.cuda(), Gb.cuda(), mode, rs, g_scale=float(B))
        bad = torch.zeros_like(w0, dtype=torch.bool)
        bad[3, 200, :] = True
        bad[5, 7, :] = True
        bad[6, :, 129] = True
        assert not torch.isfinite(w1[bad]).any(), (mode, rs)
        assert torch.equal(w1[~bad], w0[~bad]), (mode, rs)
        assert not torch.isfinite(b1[3, 200]) and not torch.isfinite(b1[5, 7])
        clean = torch.ones_like(b0, dtype=torch.bool)
        clean[3, 200] = clean[5, 7] = False
        assert torch.equal(b1[clean], b0[clean])
        # the poisoned samples' squared norms are non-finite, the others' untouched
        s0, s1 = p0.sum(1), p1.sum(1)
        assert not torch.isfinite(s1[[3, 5, 6]]).any() and torch.equal(s1[[0, 1, 2, 4, 7]], s0[[0, 1, 2, 4, 7]])


@pytest.mark.parametrize("T,in_f,out_f", SPECIAL_SHAPES)
def test_split_arithmetic_zeros_subnormals_and_dynamic_range(nat, T, in_f, out_f):
    B = 8
    g = torch.Generator().manual_seed(3)
    A = torch.randn(B, T, in_f, generator=g)
    G = torch.randn(B, T, out_f, generator=g)
    # sample 0: all-zero backprops; sample 1: negative zeros; sample 2: rows 10..19 zero; sample 3: scaled so that
    # g_scale * G is SUBNORMAL (1e-39 .. 1e-41); samples 4..7: twelve decades across t within every column
    G[0] = 0.0
    G[1] = -0.0
    G[2, :, 10:20] = 0.0
    G[3] *= 1e-40
    decades = 10.0 ** torch.linspace(-6, 6, T)
    G[4:] *= decades[None, :, None]
    ref = torch.einsum("bto,bti->boi", G.double(), A.double())
    tiny = 2.0 ** -126
    for mode, rs in _special_modes(T):
        w, b, part = _k1_run(nat, A.cuda(), G.cuda(), mode, rs)
        w = w.double().cpu()
        assert torch.isfinite(w).all()
        assert not w[0].any() and not w[1].any() and not w[2, 10:20].any(), (mode, rs)
        assert float(part[0].abs().max()) == 0.0 and float(part[1].abs().max()) == 0.0
        # subnormal operands may be flushed by the matrix pipes (either of them): what can be lost is at most one
        # smallest-normal-sized product per step
        assert float((w[3] - ref[3]).abs().max()) <= T * tiny * float(A[3].abs().max()), (mode, rs)
        # dynamic range: every column on its own scale
        err = ((w[4:] - ref[4:]).abs().amax(1) / ref[4:].abs().amax(1)).max()
        assert float(err) < 1e-6, (mode, rs, float(err))
        err2 = ((w[2] - ref[2]).abs().amax(0) / ref[2].abs().amax(0)).max()
        assert float(err2) < 1e-6
