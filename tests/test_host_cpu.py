"""Host-side logic and the C-ABI surface — no GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, T, golden, manifest, maxdiff


@pytest.fixture(scope="module")
def built_library():
    """The shared library is a build artefact (git-ignored): build it with hipcc when it is not there yet
    (cross-compiles for gfx950 without a GPU, ~2 min), exactly what __graft_entry__.build() does."""
    from str2str_amd import build, ops

    if not os.path.exists(ops.LIB_PATH):
        build.build(verbose=False)
    return ops.LIB_PATH


def test_library_exports_every_declared_symbol(built_library):
    from str2str_amd import ops

    hdr = open(os.path.join(ROOT, "include", "str2str_hip.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|long long)\s+(s2s_\w+)\s*\(", hdr, flags=re.M)))
    assert declared and set(declared) == set(ops.EXPORTS), (declared, ops.EXPORTS)
    lib = ops.load_library()  # dlopen + symbol lookup for every entry point; raises if one is missing
    for name in declared:
        assert isinstance(getattr(lib, name), ctypes._CFuncPtr)
    assert lib.s2s_abi_version() == ops.ABI_VERSION


def test_missing_library_fails_loudly(tmp_path):
    from str2str_amd import ops

    with pytest.raises(ops.HipLibraryError, match="no CPU fallback"):
        ops.load_library(str(tmp_path / "nope.so"))


def test_no_cpu_fallback_in_product_path():
    from str2str_amd import ops
    from str2str_amd.factory import build_net
    from str2str_amd.synth import synth_chain

    net = build_net()  # parameters on the host
    feats = synth_chain(8)
    batch = {k: v for k, v in feats.items() if isinstance(v, torch.Tensor)}
    batch.update(t=torch.ones(1) * 0.5, sc_ca_t=torch.zeros(1, 8, 3), rigids_t=torch.zeros(1, 8, 7))
    with pytest.raises(ops.HipLibraryError):
        net(batch)
    with pytest.raises(ops.HipLibraryError):
        ops.edge_transition(torch.zeros(1, 2, 2, 128), torch.zeros(1, 2, 768), torch.zeros(1, 2, 128), *[torch.zeros(1)] * 8)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "str2str_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(d, f)
                assert "/root/reference" not in src, os.path.join(d, f)


def test_state_dict_contract():
    from str2str_amd.factory import build_net

    sd = build_net().state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == manifest()
    assert len(sd) == 274 and sum(v.numel() for v in sd.values()) == 17446106


def test_pack_weight_layout():
    from str2str_amd.ops import pack_weight

    w = torch.arange(40 * 16, dtype=torch.float32).reshape(40, 16)
    p = pack_weight(w).reshape(2, 2, 64, 4)  # [s4, t, lane, q]  (S4 = 2, T = 2 after padding 40 -> 64)
    for s4, t, lane, q in [(0, 0, 0, 0), (1, 0, 33, 2), (0, 1, 7, 3), (1, 1, 63, 1)]:
        row, col = 32 * t + (lane & 31), 8 * s4 + 4 * (lane >> 5) + q
        want = float(w[row, col]) if row < 40 else 0.0
        assert float(p[s4, t, lane, q]) == want


def test_f16x3_weight_split_and_stream_layout():
    """Host side of the split-f16 kernels (csrc/pair_mlp_f16.hip, node_gemm.hip): fragments sit where the kernels' lanes read them,
    the weight pair (W_h, W_l) of 2^5 w recovers w to fp32 rounding over the weights' range (the factor keeps W_l out of f16's
    subnormals for |w| >= 2^-14), a weight beyond the packing's range is refused, the stream follows the slot schedule documented
    in csrc/pair_mlp_f16.hip (A_0 A_1 | B_0 A_2 | ... | B_10 B_11 | F), the fp32 node packing follows pack_weight's lane order, and
    the gather tables' column blocking is a pure permutation."""
    import pytest

    from str2str_amd import ops

    # chain order: element j of lane (row m, k-group g) in k-step ks is W[32t + m][32t' + (r&3) + 8(r>>2) + 4g], t' = ks>>1, r = 8(ks&1)+j
    wk = torch.arange(64 * 64, dtype=torch.float32).reshape(64, 64)
    fo = ops.fragment_order(wk, "chain")  # [KS=4, T=2, 64, 8]
    assert fo.shape == (4, 2, 64, 8)
    for ks, t, lane, j in [(0, 0, 0, 0), (1, 1, 37, 5), (3, 0, 63, 7), (2, 1, 31, 3)]:
        r = 8 * (ks & 1) + j
        col = 32 * (ks >> 1) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        assert float(fo[ks, t, lane, j]) == float(wk[32 * t + (lane & 31), col]), (ks, t, lane, j)
    assert float(ops.fragment_order(wk, "row")[1, 0, 40, 2]) == float(wk[8, 16 + 8 + 2])
    with pytest.raises(ops.WeightRangeError):
        ops.pack_f16x2_layer(torch.full((32, 32), 2048.0))
    # fp32 node packing [cb][s4][t][lane][q] = W[32 (cb TG + t) + (lane & 31)][8 s4 + 4 (lane >> 5) + q]
    w32 = torch.arange(128 * 16, dtype=torch.float32).reshape(128, 16)
    p32 = ops.pack_node_weight_f32(w32, 2).reshape(2, 2, 2, 64, 4)
    for cb, s4, t, lane, q in [(0, 0, 0, 0, 0), (1, 1, 0, 33, 2), (0, 1, 1, 7, 3), (1, 0, 1, 63, 1)]:
        assert float(p32[cb, s4, t, lane, q]) == float(w32[32 * (cb * 2 + t) + (lane & 31), 8 * s4 + 4 * (lane >> 5) + q])

    g = torch.Generator().manual_seed(1)
    w = torch.randn(96, 64, generator=g) * torch.logspace(-4, 1, 64)        # |w| from 1e-4 to ~30: beyond any trained layer
    pk = ops.pack_f16x2_layer(w, "chain")                                  # [KS=4, T=3, 2, 64, 8]
    assert pk.shape == (4, 3, 2, 64, 8) and pk.dtype == torch.float16
    ref = ops.fragment_order(w, "chain")                                   # the same fragments in fp32
    rec = (pk[:, :, 0].double() + pk[:, :, 1].double()) / 32.0
    err = (rec - ref.double()).abs()
    assert (err <= 2.0 ** -23 * ref.double().abs() + 2.0 ** -30).all(), float((err / ref.abs().clamp_min(1e-30)).max())
    assert (pk[:, :, 1].float().abs() <= pk[:, :, 0].float().abs() * 2.0 ** -10 + 2.0 ** -24).all()   # W_l is the residue of W_h
    assert torch.isfinite(pk.float()).all() and float(pk[:, :, 0].float().abs().max()) < 65504

    w1, w2, wf = torch.randn(384, 128, generator=g), torch.randn(384, 384, generator=g), torch.randn(128, 384, generator=g)
    st = ops.pack_f16x3_stream(w1, w2, wf).view(torch.float16).reshape(240, 4, 64, 8)
    l1, l2, lf = ops.pack_f16x2_layer(w1), ops.pack_f16x2_layer(w2), ops.pack_f16x2_layer(wf)
    assert torch.equal(st[0], l1[0:2, 0].reshape(4, 64, 8))             # A_0 slot 0: k-steps 0, 1 of tile 0, [k-step][plane]
    assert torch.equal(st[4 + 3], l1[6:8, 1].reshape(4, 64, 8))         # A_1 slot 3
    assert torch.equal(st[8 + 7], l2[1, 2:4].reshape(4, 64, 8))         # B_0 slot (u = 1, pair 1): k-step 1, tiles 2, 3
    assert torch.equal(st[8 + 12 + 1], l1[2:4, 2].reshape(4, 64, 8))    # A_2 slot 1 follows B_0
    assert torch.equal(st[168 + 7], l2[21, 2:4].reshape(4, 64, 8))       # B_10 (u = 1, pair 1): k-step major like every B_t but the last
    assert torch.equal(st[180 + 1], l2[23, 0:2].reshape(4, 64, 8))       # B_11 is tile-pair major: slot 2 b + u = (pair b, k-step 22 + u)
    assert torch.equal(st[180 + 6], l2[22, 6:8].reshape(4, 64, 8))
    assert torch.equal(st[192 + 2 * 5 + 1], lf[5, 2:4].reshape(4, 64, 8))  # final layer k-step 5, tiles 2, 3
    assert ops.pack_f16x3_embed_stream(torch.randn(128, 128, generator=g), torch.randn(128, 128, generator=g)).numel() * 2 == 4 * 32 * 1024
    assert ops.pack_node_weight(torch.randn(256, 320, generator=g), 8).numel() == 256 * 320 * 2

    t = torch.randn(3, 7, 128, generator=g)
    cb = ops.column_blocked(t)                                             # [3, 32, 7, 4]
    assert cb.shape == (3, 32, 7, 4) and torch.equal(cb.permute(0, 2, 1, 3).reshape(3, 7, 128), t)


def test_rotation_and_rigid_host_types():
    from str2str_amd.common import rotation3d as R3
    from str2str_amd.common.rigid_utils import Rigid, Rotation, quat_multiply, quat_to_rot

    g = golden("prims.npz")
    q, R, aa = T(g["q"]), T(g["R"]), T(g["aa"])
    assert maxdiff(quat_to_rot(q), R) < 1e-6
    assert maxdiff(R3.matrix_to_quaternion(R), g["m2q"]) < 1e-6
    assert maxdiff(R3.quaternion_to_axis_angle(q), g["q2aa"]) < 2e-6
    assert maxdiff(R3.axis_angle_to_matrix(aa), g["aa2m"]) < 1e-6
    assert maxdiff(R3.matrix_to_axis_angle(R), g["m2aa"]) < 2e-6
    assert maxdiff(quat_multiply(q, T(g["q2"])), g["qmul"]) < 1e-6
    rig = Rigid(Rotation(quats=q, normalize_quats=False), T(g["t"]))
    comp = rig.compose_q_update_vec(T(g["upd"]), T(g["msk"]))  # host tensors: plain torch value-type path
    assert maxdiff(comp.to_tensor_7(), g["comp7"]) < 1e-6
    back = Rigid.from_tensor_4x4(rig.to_tensor_4x4())
    assert maxdiff(back.get_rots().get_rot_mats(), R) < 1e-6 and back.shape == rig.shape
    assert rig[3:5].shape == (2,) and maxdiff(rig[3:5].get_trans(), g["t"][3:5]) == 0
    pts = torch.randn(q.shape[0], 3)
    assert maxdiff(rig.invert_apply(rig.apply(pts)), pts) < 1e-5
    p = T(g["p3"])
    f3 = Rigid.from_3_points(p[:, 0], p[:, 1], p[:, 2])
    assert maxdiff(f3.get_rots().get_rot_mats(), g["f3_rot"]) < 1e-6


def test_forward_marginal_and_prior_match_reference_noise(tmp_path):
    from str2str_amd.common.rigid_utils import Rigid
    from str2str_amd.factory import build_diffuser

    g = golden("forward_marginal.npz")
    d = build_diffuser(str(tmp_path))
    torch.manual_seed(int(g["seed_fm"]))
    fm = d.forward_marginal(Rigid.from_tensor_4x4(T(g["gt4"])), float(g["t_delta"]) * torch.ones(3), T(g["mask"]))["rigids_t"]
    assert maxdiff(fm, g["rigids_t"]) < 5e-6
    torch.manual_seed(int(g["seed_prior"]))
    pr = d.sample_prior(shape=torch.Size([3, 10]), device="cpu", as_tensor_7=True)["rigids_t"]
    assert maxdiff(pr, g["prior"]) < 5e-6


def test_start_frames_do_not_depend_on_the_host_thread_count(tmp_path):
    """sampler._start_frames draws and assembles a chunk's frames on ONE intra-op thread (S2S_HOST_FM_THREADS): the same bits as with
    the whole pool, at a size above torch's parallel grain (64 x 256 residues)."""
    from str2str_amd.common.rigid_utils import Rigid
    from str2str_amd.factory import build_diffuser

    d = build_diffuser(str(tmp_path))
    B, N = 64, 256
    q = torch.randn(N, 4, generator=torch.Generator().manual_seed(3))
    r0 = Rigid.from_tensor_7(torch.cat([q / q.norm(dim=-1, keepdim=True), 10 * torch.randn(N, 3, generator=torch.Generator().manual_seed(4))], -1)[None].repeat(B, 1, 1))
    keep, out = torch.get_num_threads(), []
    try:
        for nt in (1, max(2, keep)):
            torch.set_num_threads(nt)
            torch.manual_seed(11)
            out.append(d.forward_marginal(rigids_0=r0, t=0.4 * torch.ones(B), diffuse_mask=torch.ones(B, N))["rigids_t"])
            out.append(d.sample_prior(shape=r0.shape, device="cpu", as_tensor_7=True)["rigids_t"])
    finally:
        torch.set_num_threads(keep)
    assert torch.equal(out[0], out[2]) and torch.equal(out[1], out[3])


def test_schedule_and_step_params(tmp_path):
    from str2str_amd.factory import build_diffuser
    from str2str_amd.sampler import schedule, shard_range

    g = golden("schedule.npz")
    d = build_diffuser(str(tmp_path))
    for i, (nt, Tt) in enumerate([(20, 1.0), (100, 1.0), (1000, 0.25), (1000, 0.7)]):
        T_, n, dt, ts = schedule(Tt, nt, 0.01)
        assert (ts == g[f"ts_{i}"]).all() and dt == float(g[f"dt_{i}"])
        p8 = d.step_params(torch.as_tensor(ts.copy()).float())
        assert maxdiff(p8[:, 0], g[f"sigma_{i}"]) == 0
        assert maxdiff(p8[:, 6], g[f"g_rot_{i}"]) == 0 and maxdiff(p8[:, 1], T(g[f"g_rot_{i}"]) ** 2) == 0
        assert maxdiff(p8[:, 2], torch.exp(-0.5 * T(g[f"mb_t_{i}"]))) == 0
        assert maxdiff(p8[:, 3], g[f"cond_var_{i}"]) == 0 and maxdiff(p8[:, 4], g[f"b_t_{i}"]) == 0
    assert schedule(-1.0, 10, 0.01)[1] == 10
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    sd = golden("so3_score.npz")
    for row, idx in zip(sd["cdf_rows"], sd["cdf_row_idx"]):
        assert np.abs(d.rot_diffuser.cdf_row(int(idx)) - row).max() < 1e-12


def test_mixed_length_plan_covers_every_replica_once_and_balances():
    """cfg5 work distribution: 32 chains U[64,384] (seed 5) x 256 replicas over 1, 8 and 3 ranks, plus awkward
    cases (fewer replicas than ranks): every (chain, replica) exactly once, FLOP load balanced, batches inside the memory cap."""
    import numpy as np

    from str2str_amd.sampler import forward_flops, plan_mixed_work

    lens = [int(x) for x in np.random.default_rng(5).integers(64, 385, size=32)]
    for replicas, world in [(256, 1), (256, 8), (256, 3), (3, 8), (1, 4)]:
        plan = plan_mixed_work(lens, replicas, world)
        seen = {}
        loads = []
        for r in range(world):
            load = 0.0
            for b in plan[r]:
                assert b["n_pad"] == max(lens[k] for k, _, _ in b["items"])
                assert sum(hi - lo for _, lo, hi in b["items"]) * b["n_pad"] ** 2 <= 24 << 20 or len(b["items"]) == 1
                for k, lo, hi in b["items"]:
                    for q in range(lo, hi):
                        assert (k, q) not in seen
                        seen[(k, q)] = r
                    load += forward_flops(lens[k]) * (hi - lo)
            loads.append(load)
        assert len(seen) == len(lens) * replicas
        if replicas >= world:
            assert max(loads) <= 1.12 * sum(loads) / world, (replicas, world, loads)


def test_mixed_length_batches_never_share_a_host_noise_stream():
    """Every rank of a run starts from the same seed (eval.py), so the host generators are re-seeded per padded batch from (run seed,
    t_delta, first work item) -- sampler.mixed_batch_seed.  The advisor's case: lens [10, 20, 28, 35, 80] x 100 replicas over 4 ranks,
    where every rank gets the same SHAPES of batches: no two batches of the run (on any ranks, at any t_delta) may get the same seed."""
    from str2str_amd.sampler import mixed_batch_seed, plan_mixed_work

    lens = [10, 20, 28, 35, 80]
    for world in (1, 2, 4, 8):
        plan = plan_mixed_work(lens, 100, world)
        seeds = [mixed_batch_seed(3, t, *b["items"][0][:2]) for t in (0.1, 0.5, 1.0) for r in range(world) for b in plan[r]]
        assert len(set(seeds)) == len(seeds) and all(0 <= s < 2 ** 63 for s in seeds)
    assert mixed_batch_seed(3, 0.5, 1, 25) != mixed_batch_seed(4, 0.5, 1, 25)


def test_tica_fit_recovers_the_slow_mode_of_a_two_state_process():
    """metrics.tica_fit (the estimator behind js_tica when deeptime is not installed; reference src/metrics/metrics.py:169-200) on a
    process with a known answer: a two-state jump process s_t = +-1 (switching probability p per frame, autocorrelation (1 - 2p)^lag)
    seen through 6 features x_t = s_t a + fast noise + two constant-zero-variance directions (rank-deficient C00, as pairwise
    distances are).  The slowest component must be the optimal linear read-out of s_t (correlation sqrt(snr / (1 + snr)) with it), its
    autocorrelation at the lag must match the analytic value, the second component must carry no slow signal, and the absolute eigenvalue cut-off must drop the
    degenerate directions instead of amplifying them."""
    import numpy as np

    from str2str_amd.metrics.metrics import tica_fit

    rng = np.random.default_rng(0)
    T, p, lag = 20000, 0.01, 20
    flips = rng.random(T) < p
    s = np.where(np.cumsum(flips) % 2 == 0, 1.0, -1.0)
    a = np.array([2.0, -1.0, 0.5, 3.0])
    x = s[:, None] * a[None, :] + rng.normal(size=(T, 4)) * np.array([1.0, 2.0, 0.5, 4.0])
    x = np.concatenate([x, np.full((T, 1), 7.0), x[:, :1] * 1e-5], axis=1)    # a constant feature and a copy scaled below the cut-off
    mean, proj = tica_fit(x, lag, dim=2)
    assert proj.shape == (6, 2) and np.isfinite(proj).all()
    y = (x - mean) @ proj
    snr = np.sum((a / np.array([1.0, 2.0, 0.5, 4.0])) ** 2)             # signal variance in the noise-whitened direction
    c1 = np.corrcoef(y[:, 0], s)[0, 1]
    assert abs(abs(c1) - np.sqrt(snr / (1 + snr))) < 0.01, c1
    assert abs(np.corrcoef(y[:, 1], s)[0, 1]) < 0.05
    # whitened components have unit variance; the autocorrelation of the first at the lag = signal fraction x (1 - 2p)^lag
    y0 = y[:, 0]
    auto = np.mean(y0[:-lag] * y0[lag:]) / np.mean(y0 * y0)
    want = (snr / (1 + snr)) * (1 - 2 * p) ** lag
    assert abs(auto - want) < 0.03, (auto, want)
    assert np.abs(proj[4:]).max() < 1e3          # neither the constant nor the sub-cut-off copy is blown up by the whitening


def test_tica_restatement_and_product_estimator_agree_with_the_reference_driven_fixture():
    """`js_tica` (reference src/metrics/metrics.py:166-200).  The estimator is the third-party deeptime.decomposition.TICA
    (deeptime==0.4.4), absent here: oracle/tica.py restates its published algorithm -- NOT a run of deeptime -- and
    tests/golden/tica.npz holds what the REFERENCE's own js_tica returned when it drove that restatement
    (tests/golden/make_golden_tica.py).  Here, on CPU: (1) the oracle's stand-alone js_tica (its own histogram / weights tail)
    reproduces the reference-driven values, so oracle.tica.js_tica == reference js_tica around the same estimator; (2) the product's
    numpy estimator (metrics.tica_fit: what runs when deeptime is not installed) gives the restatement's projections -- magnitude
    ordering with a NEGATIVE second eigenvalue (case a), canonical signs and kinetic-map scale included."""
    import numpy as np

    from conftest import golden
    from oracle import tica as OT
    from str2str_amd.metrics.metrics import tica_fit

    g = golden("tica.npz")

    def pwd(x, k=1):   # reference pairwise_distance_ca (metrics.py:38-50), float32
        d = np.sqrt(np.sum((x[..., None, :, :] - x[..., None, :]) ** 2, axis=-1))
        r, c = np.triu_indices(x.shape[-2], k=k)
        return d[..., r, c]

    for tag in ("a", "b"):
        lag = int(g[f"{tag}_lag"])
        feats = {"target": pwd(g[f"{tag}_target"]), "pred": pwd(g[f"{tag}_pred"])}
        res, tics = OT.js_tica(feats, lagtime=lag)
        assert np.around(res["pred"], 4) == float(g[f"{tag}_js_tica"])
        assert np.abs(tics["pred"] - g[f"{tag}_tic_pred"]).max() < 1e-9 and np.abs(tics["target"] - g[f"{tag}_tic_target"]).max() < 1e-9
        res_w, _ = OT.js_tica(feats, lagtime=lag, weights={"pred": g[f"{tag}_weights"]})
        assert np.around(res_w["pred"], 4) == float(g[f"{tag}_js_tica_w"]) and res_w["pred"] != res["pred"]
        est = OT.TICA(dim=2, lagtime=lag).fit(feats["target"])
        assert np.allclose(est.eigenvalues[:4], g[f"{tag}_tica_eigenvalues"], rtol=1e-9)
        mean, proj = tica_fit(feats["target"], lag, dim=2)
        scale = np.abs(est.coefficients).max()
        assert np.abs(mean - est.mean).max() < 1e-12 and np.abs(proj - est.coefficients).max() < 1e-8 * scale
        assert np.abs((feats["pred"].astype(np.float64) - mean) @ proj - g[f"{tag}_tic_pred"]).max() < 1e-8 * np.abs(g[f"{tag}_tic_pred"]).max()
    assert g["a_tica_eigenvalues"][1] < 0          # the edge case the magnitude ordering exists for


def test_pair_tiled_layout_matches_the_header():
    """ops.pair_tiled / pair_untiled against the formula of include/str2str_hip.h ("Pair-tensor layouts"): channel 8 g + 4 h + q of pair
    32 b + n at float offset 4096 b + 256 g + 128 h + 4 n + q; whole blocks, zero padding; round trip."""
    import torch

    from str2str_amd import ops

    for B, N in ((1, 8), (2, 5), (1, 3)):
        M = B * N * N
        z = torch.arange(M * 128, dtype=torch.float32).reshape(B, N, N, 128)
        t = ops.pair_tiled(z)
        assert t.buf.numel() == -(-M // 32) * 32 * 128 and t.shape == (B, N, N, 128)
        flat = z.reshape(M, 128)
        for p in {0, 1, 31, 32, M // 2, M - 1} & set(range(M)):
            for c in (0, 3, 4, 7, 8, 77, 127):
                b, n, g, h, q = p // 32, p % 32, c // 8, (c % 8) // 4, c % 4
                assert t.buf[4096 * b + 256 * g + 128 * h + 4 * n + q] == flat[p, c]
        assert torch.equal(ops.pair_untiled(t), z)
        if M % 32:
            pad = t.buf.view(-1, 16, 2, 32, 4)[-1, :, :, M % 32:, :]
            assert float(pad.abs().max()) == 0.0


def test_every_global_name_the_package_uses_resolves():
    """A function that went missing shows up only when its caller runs on a GPU box: check on the CPU that every global name the
    package's functions and methods (and the torch-op implementations, lambdas included) load exists in its module or the builtins."""
    import builtins
    import dis
    import importlib
    import pkgutil
    import types

    import str2str_amd

    def codes(c):
        yield c
        for k in c.co_consts:
            if isinstance(k, types.CodeType):
                yield from codes(k)

    missing = set()
    for info in pkgutil.walk_packages(str2str_amd.__path__, "str2str_amd."):
        if info.name.rsplit(".", 1)[-1].startswith("lib"):         # the built shared library, not a Python module
            continue
        mod = importlib.import_module(info.name)
        fns = [v for v in vars(mod).values() if isinstance(v, types.FunctionType) and v.__module__ == mod.__name__]
        for cls in [v for v in vars(mod).values() if isinstance(v, type) and v.__module__ == mod.__name__]:
            fns += [getattr(v, "__func__", v) for v in vars(cls).values()
                    if isinstance(getattr(v, "__func__", v), types.FunctionType)]
        if info.name == "str2str_amd.ops":
            fns += [f for f in mod._TORCH_OPS.values() if isinstance(f, types.FunctionType)]
        for f in fns:
            for c in codes(f.__code__):
                for ins in dis.get_instructions(c):
                    if ins.opname == "LOAD_GLOBAL" and ins.argval not in f.__globals__ and not hasattr(builtins, ins.argval):
                        missing.add((info.name, f.__name__, ins.argval))
    assert not missing, sorted(missing)


def test_edge_prescale_ladder():
    """The sampler's answer to an edge-transition range flag (sampler._raise_edge_prescale): the block exponent of EVERY f16x3
    EdgeTransition goes 0 -> 5 -> 10 -> 15 and then reports that nothing is left (the family is demoted to fp32); modules on the exact
    arithmetic are not touched."""
    import torch.nn as nn

    from str2str_amd.sampler import _raise_edge_prescale

    class ET(nn.Module):
        def __init__(self, arith):
            super().__init__()
            self.arith, self.prescale_exp = arith, 0

    net = nn.ModuleList([ET("f16x3"), ET("f16x3"), ET("f32")])
    seen = []
    while True:
        e = _raise_edge_prescale(net)
        if not e:
            break
        seen.append(e)
        assert [m.prescale_exp for m in net] == [e, e, 0]
    assert seen == [5, 10, 15]
    assert _raise_edge_prescale(nn.ModuleList([ET("f32")])) == 0


def test_chunk_merging_groups(monkeypatch):
    """sampler.merge_chunk_groups: consecutive replica chunks are sampled as one trajectory while this rank's replicas in the group
    stay below the pair budget; order and membership are untouched; the switch and the SDE-with-host-noise case keep single chunks."""
    from str2str_amd.sampler import merge_chunk_groups, rank_chunk_slices

    monkeypatch.delenv("S2S_MERGE_CHUNKS", raising=False)
    ref_default = rank_chunk_slices(100, 64, 0, 1)
    assert ref_default == [(64, 0, 64), (36, 0, 36)]
    assert merge_chunk_groups(ref_default, 35) == [ref_default] and merge_chunk_groups(ref_default, 80) == [ref_default]
    assert merge_chunk_groups(ref_default, 512) == [[c] for c in ref_default]              # 64 x 512^2 alone exceeds the budget
    assert merge_chunk_groups(rank_chunk_slices(256, 128, 0, 1), 256) == [[(128, 0, 128)], [(128, 0, 128)]]   # cfg2-sized chunks stay
    for n, rpb, world, N in [(1000, 64, 8, 80), (100, 64, 3, 35), (9, 4, 8, 20), (130, 64, 4, 300)]:
        for r in range(world):
            chunks = rank_chunk_slices(n, rpb, r, world)
            groups = merge_chunk_groups(chunks, N)
            assert [c for g in groups for c in g] == chunks
            for g in groups:
                assert len(g) == 1 or sum(hi - lo for _, lo, hi in g) * N * N <= 8 << 20
    assert merge_chunk_groups(ref_default, 35, mergeable=False) == [[c] for c in ref_default]
    monkeypatch.setenv("S2S_MERGE_CHUNKS", "0")
    assert merge_chunk_groups(ref_default, 35) == [[c] for c in ref_default]


def test_host_generator_fast_forward_equals_the_discarded_draws(built_library, monkeypatch):
    """Parity mode consumes the reference's per-step float64 draws that the probability-flow ODE never uses (so3.py:360, r3.py:109 there).
    ``sampler._burn_step_draws`` fast-forwards torch's CPU generator over them (s2s_mt19937_discard behind
    ops.host_rng_discard_float64_normals): the generator state -- and everything drawn afterwards -- must be exactly what the real
    draws leave, for chunk sizes with and without ATen's re-drawn tail block, from any engine position, across many twists; tensors
    below 16 elements (ATen's scalar path) and S2S_HOST_RNG_FAST=0 take the real draws."""
    import torch

    from str2str_amd import ops
    from str2str_amd.sampler import _burn_step_draws

    monkeypatch.delenv("S2S_HOST_RNG_FAST", raising=False)
    assert ops.host_rng_fast_forward_ok()
    assert ops.float64_normal_outputs(48) == 96 and ops.float64_normal_outputs(50) == 132

    def real(B, N, steps):
        for _ in range(steps):
            torch.randn(B, N, 3, dtype=torch.float64)
            torch.randn(B, N, 3, dtype=torch.float64)

    for seed, pre, (B, N, steps) in [(0, 0, (64, 80, 40)), (1, 5, (36, 35, 57)), (2, 623, (3, 7, 9)), (3, 1, (1, 6, 11)), (4, 300, (100, 10, 700)),
                                     (5, 2, (1, 5, 4))]:          # (1 x 5 x 3 = 15 elements: the scalar path, drawn for real)
        torch.manual_seed(seed)
        torch.rand(pre)
        start = torch.get_rng_state()
        real(B, N, steps)
        want, after = torch.get_rng_state(), torch.randn(7)
        torch.set_rng_state(start)
        _burn_step_draws(B, N, steps)
        assert torch.equal(torch.get_rng_state(), want) and torch.equal(torch.randn(7), after), (seed, B, N, steps)
    assert not ops.host_rng_discard_float64_normals(15, 4)          # below ATen's block path: the caller draws
    monkeypatch.setenv("S2S_HOST_RNG_FAST", "0")
    assert not ops.host_rng_discard_float64_normals(4800, 4)
    torch.manual_seed(9)
    start = torch.get_rng_state()
    real(4, 20, 3)
    want = torch.get_rng_state()
    torch.set_rng_state(start)
    _burn_step_draws(4, 20, 3)
    assert torch.equal(torch.get_rng_state(), want)


def test_t_delta_merging_groups(monkeypatch):
    """sampler.merge_delta_groups: consecutive t_deltas of a target are sampled as one growing batch while their replicas fit the pair
    budget at the batch's end (the reference's default block: ten t_deltas x 100 replicas on chains of 35 .. 80 residues -> one batch);
    larger targets fall back to fewer t_deltas per batch, down to one (cfg2-sized chunks); the switch keeps one t_delta per batch."""
    from str2str_amd.sampler import merge_delta_groups, schedule

    monkeypatch.delenv("S2S_MERGE_DELTAS", raising=False)
    deltas = [round(0.25 + 0.05 * k, 2) for k in range(10)]
    steps = [schedule(d, 1000, 0.01)[1] for d in deltas]
    assert steps == [250, 300, 350, 400, 450, 500, 550, 600, 650, 700]
    assert schedule(0.25, 1000, 0.01)[2] == 1.0 / 250 and schedule(-1.0, 1000, 0.01)[1] == 1000      # dt = 1 / int(num_timesteps T); prior: T = 1
    assert merge_delta_groups(steps, 100, 35) == [list(range(10))] and merge_delta_groups(steps, 100, 80) == [list(range(10))]
    assert merge_delta_groups(steps, 100, 160) == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9]]             # 3 x 100 x 160^2 <= 8 Mi pairs < 4 x ...
    assert merge_delta_groups(steps, 128, 256) == [[i] for i in range(10)]                           # a cfg2-sized chunk per t_delta
    assert merge_delta_groups(steps, 0, 80) == [list(range(10))]                                     # (a rank without replicas: nothing to hold)
    for b, n in [(100, 35), (64, 200), (1000, 20), (7, 512)]:
        groups = merge_delta_groups(steps, b, n)
        assert [i for g in groups for i in g] == list(range(10))
        assert all(len(g) == 1 or len(g) * b * n * n <= 8 << 20 for g in groups)
    monkeypatch.setenv("S2S_MERGE_DELTAS", "0")
    assert merge_delta_groups(steps, 100, 35) == [[i] for i in range(10)]
