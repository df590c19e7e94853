"""v2 program compiler (engine/program2.py: tiles / K slices / item tables / record packing / slot plan / epilogue partition)
proven on CPU: the lane-level model of csrc/cdx_unet2.hip (oracle/lane_sim2.py) interprets the compiled program and must
reproduce the reference's first forward (``pred0`` of the fixtures, produced by the real reference).  Tolerance 2e-5: same
fp32 math, different summation order."""
import numpy as np
import pytest
import torch

from cleandiffuser_amd.engine import program2 as P2
from oracle import cases
from oracle.lane_sim2 import LaneSim2, emb_table
from cleandiffuser_amd.utils import load_synth
from conftest import golden_path


def _first_forward_inputs(name, agent):
    c = cases.CASES[name]
    inp = cases.make_inputs(name)
    temp = c["sample"].get("temperature", 1.0)
    xt0 = inp["noise"][0] * np.float32(temp)
    if inp["fix_mask"] is not None:
        xt0 = xt0 * (1 - inp["fix_mask"][None]) + inp["prior"] * inp["fix_mask"][None]
    return inp, xt0.astype(np.float32)


def _first_t(agent, c):
    from cleandiffuser_amd.utils import SUPPORTED_SAMPLING_STEP_SCHEDULE as SS
    S = c["sample"]["sample_steps"]
    if c["solver"][0] == "DiscreteDiffusionSDE":
        sched = SS[c["sample"].get("sample_step_schedule", "uniform")](agent.diffusion_steps, S)
        return torch.tensor([int(sched[S])], dtype=torch.long)
    sched = SS[c["sample"].get("sample_step_schedule", "uniform_continuous")](agent.t_diffusion, S)
    return torch.tensor([float(sched[S])], dtype=torch.float32)


@pytest.mark.parametrize("nw", [4, 8])
@pytest.mark.parametrize("name", ["janner_cfg2_ddim", "janner_h4_ddpm", "janner_tiny_disc_ddim", "janner_tiny_cont_ddim", "janner_h64_single"])
def test_lane_sim2_reproduces_reference_forward(name, nw, amd_lib):
    gold = np.load(golden_path(name))
    agent, _ = cases.build(amd_lib, name)
    c = cases.CASES[name]
    net = agent.model_ema["diffusion"]
    prog = P2.compile_janner2(net, c["horizon"], nw=nw)
    assert prog.lds_bytes(1) <= 160 * 1024 and prog.nw == nw and prog.ops.shape[1] == P2.op_words(nw)
    inp, xt0 = _first_forward_inputs(name, agent)
    with torch.no_grad():
        temb = net.map_noise(_first_t(agent, c)).numpy()
    row = emb_table(prog, temb)[0]
    for b in range(2 if name == "janner_cfg2_ddim" else c["batch"]):
        sim = LaneSim2(prog)
        sim.load_x(xt0[b])
        np.testing.assert_allclose(sim.run_forward(row), gold["pred0"][b], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("nw", [4, 8])
@pytest.mark.parametrize("shape", [(32, 69, 64, [1, 2, 2, 2]), (16, 6, 32, [1, 2, 4])])
def test_lane_sim2_wide_nets_against_module_forward(shape, nw, amd_lib):
    """Shapes that leave the one-item-per-wave regime: the shipped kitchen Diffuser net (64 channels x 32 positions = 8 tiles, so
    waves loop over items of the tail table; concat layers with K slices cut at the source boundary; 2 float4 items per lane in the
    epilogue; C_out = 69 padded to 128 lanes) and a 3-level net with a x4 channel step.  Against the module's own forward (which
    is bit-identical to the reference's, tests/test_module_mirrors.py)."""
    from cleandiffuser_amd.utils import load_synth
    H, D, md, dm = shape
    net = load_synth(amd_lib.JannerUNet1d(D, model_dim=md, emb_dim=32, dim_mult=dm, kernel_size=5), 9).eval()
    prog = P2.compile_janner2(net, H, nw=nw)
    assert prog.lds_bytes(1) <= 160 * 1024
    assert max(int(op[P2.W2_NITEMS]) for op in prog.ops) > nw or md < 64 or nw == 8
    g = torch.Generator().manual_seed(2)
    x, t = torch.randn(1, H, D, generator=g), torch.tensor([11])
    with torch.no_grad():
        ref = net._forward_torch(x, t, None)[0].numpy()
        row = emb_table(prog, net.map_noise(t).numpy())[0]
    sim = LaneSim2(prog)
    sim.load_x(x[0].numpy())
    np.testing.assert_allclose(sim.run_forward(row), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("nw", [4, 8])
def test_program2_accounting_and_budget(nw, amd_lib):
    """North-star config: 19.67 M MAC per forward (embedding MLP included), 40 ops, and TWO trajectories fit
    one workgroup's 160 KiB."""
    _, net = cases.build(amd_lib, "janner_cfg2_ddim")
    prog = P2.compile_janner2(net, 32, nw=nw)
    # 16 blocks x 2 convs + 3 down + 3 up + 2 head = 40; six of the seven 1x1 skip convs ride in their block's second conv op, the
    # seventh (128 -> 256 channels at L = 4, next to a 320-record, stream-bound main conv) keeps an op of its own at 8 waves
    assert len(prog.ops) == (41 if nw == 8 else 43)          # (4 waves: three long main convs keep their skip conv separate)
    assert abs(prog.macs_per_forward - 19.67e6) / 19.67e6 < 0.01
    assert prog.lds_bytes(2) <= 160 * 1024
    ops = prog.ops
    assert (ops[:, P2.W2_NITEMS] <= 8).all() and (ops[:, P2.W2_NITEMS] >= 4).all()
    if nw == 8:
        assert (ops[:, P2.W2_NITEMS] <= nw).all(), "at most one work item per wave in this net"
    assert (ops[:, P2.W2_KPOST] > 0).sum() == (6 if nw == 8 else 4), "blocks that change the channel count: fused 1x1 skip convs"
    ring = P2.ring_depth(nw)
    for op in ops:
        items = np.stack([P2.op_item(prog.ops_buffer, op, j) for j in range(op[P2.W2_NITEMS])])
        # K slices of one tile exactly tile its record stream: same record count per tile, slices of a tile are contiguous
        per_tile = items[:, P2.I2_NQ].sum() * op[P2.W2_KSPLIT] // len(items)
        assert per_tile * len(items) == items[:, P2.I2_NQ].sum() * op[P2.W2_KSPLIT]
        assert (items[:, P2.I2_NQ] >= 1).all() and (items[:, P2.I2_CCN] >= 1).all()
        assert ((items[:, P2.I2_SRCSTR] & 0xffff) < prog.traj_floats).all()
        # long K slices start on a ring-aligned chunk of their tap (the kernel's immediate-offset steady loop)
        long = items[:, P2.I2_NQ] >= 2 * ring
        assert (((items[long, P2.I2_TAPCC] >> 8) % ring == 0) | (items[long, P2.I2_CCN] % ring != 0)).all()


def test_v2_refuses_what_it_cannot_run(amd_lib):
    """Nets outside the v2 epilogue partition report a reason: the runtime sends them to the implicit-GEMM executor."""
    from cleandiffuser_amd.engine import runtime2
    net = amd_lib.JannerUNet1d(5, model_dim=24, emb_dim=16, dim_mult=[1, 2], kernel_size=3)      # 24 channels: groups of 4 != 32/8
    why = runtime2.supported(net, 8)
    assert why is not None and "GroupNorm" in why
    agent, _ = cases.build(amd_lib, "janner_cfg2_ddim")
    assert runtime2.supported(agent.model_ema["diffusion"], 32) is None
    assert runtime2.supported(agent.model_ema["diffusion"], 36) is not None


@pytest.mark.parametrize("two", [False, True])
@pytest.mark.parametrize("shape", [(16, 6, [1, 2], 32), (32, 23, [1, 2, 2, 2], 32), (32, 69, [1, 2, 2, 2], 64)])
def test_lane_sim2_guided_program_gradient_matches_autograd(shape, two, amd_lib):
    """Guided program (engine/program2.py:compile_guided2): the denoiser's ops followed by the HalfJannerUNet1d classifier's forward
    and backward-data ops (saved x_hat / rstd, tap-flipped transposed weights, the stride-2 scatter as two parity convs, the
    GroupNorm -> Mish backward epilogue, the head op).  The lane-level twin of the kernel must reproduce the denoiser forward AND
    torch.autograd's d classifier(x, t).sum() / d x of the module (bit-identical to the reference's, tests/test_module_mirrors.py)."""
    from cleandiffuser_amd.utils import load_synth
    H, D, dm, md = shape
    net = load_synth(amd_lib.JannerUNet1d(D, model_dim=md, emb_dim=md, dim_mult=dm, kernel_size=5), 0).eval()
    clf = load_synth(amd_lib.HalfJannerUNet1d(H, D, out_dim=1, model_dim=md, emb_dim=md, dim_mult=tuple(dm), kernel_size=3), 1).eval()
    # `two`: the variant for two trajectories per workgroup -- saved tensors in the global workspace, capped staging area
    # model_dim 64 (the shipped kitchen Diffuser: H = 32, D = 69) fits ONE trajectory per workgroup, and only with the saved tensors
    # in the global workspace
    if md == 64:
        if two:
            with pytest.raises(ValueError):
                P2.compile_guided2(net, clf, H, save_global=True, max_stage=2304, max_lds_bytes=80 * 1024)
            return
        with pytest.raises(ValueError):
            P2.compile_guided2(net, clf, H)
        prog = P2.compile_guided2(net, clf, H, save_global=True)
    else:
        prog = P2.compile_guided2(net, clf, H, **(dict(save_global=True, max_stage=2304) if two else {}))
    assert prog.nw == 8 and prog.lds_bytes(2 if two else 1) <= 160 * 1024 and prog.grad_off > 0 and len(prog.embtabs) == 2
    assert (prog.ws_floats > 0) == (two or md == 64)
    n_den = prog.meta["n_den"]
    assert all(int(op[P2.W2_FLAGS]) & (P2.F2_SAVE | P2.F2_GNBWD | P2.F2_DUAL) == 0 for op in prog.ops[:n_den])
    assert sum(int(op[P2.W2_KIND]) == P2.KIND2_HEAD for op in prog.ops) == 1
    g = torch.Generator().manual_seed(3)
    x, t = torch.randn(1, H, D, generator=g), torch.tensor([7])
    xr = x.clone().requires_grad_()
    clf._forward_torch(xr, t, None).sum().backward()
    with torch.no_grad():
        ref_pred = net._forward_torch(x, t, None)[0].numpy()
        row = emb_table(prog, None, [net.map_noise(t).numpy(), clf.map_noise(t).numpy()])[0]
    sim = LaneSim2(prog)
    sim.load_x(x[0].numpy())
    np.testing.assert_allclose(sim.run_forward(row), ref_pred, rtol=2e-5, atol=2e-5)
    ref_grad = xr.grad[0].numpy()
    np.testing.assert_allclose(sim.grad(), ref_grad, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(ref_grad).max())))


@pytest.mark.parametrize("shape", [(16, 6, [1, 2], 32), (32, 23, [1, 2, 2, 2], 32), (64, 37, [1, 2, 2, 2], 64)])
def test_lane_sim2_compact_guided_program(shape, amd_lib):
    """The largest shipped Diffuser net (antmaze: model_dim 64 over H = 64, D = 37) fits one workgroup only as a COMPACT guided
    program: state and multistep memory in global memory, in-place residual outputs, saved tensors in the workspace behind the
    multistep memory, and the classifier reads its own copy of x_t back from global memory (a load op) so that the denoiser's
    copy need not survive the denoiser's LDS peak.  Prediction and gradient against the module / torch.autograd."""
    from cleandiffuser_amd.utils import load_synth
    H, D, dm, md = shape
    net = load_synth(amd_lib.JannerUNet1d(D, model_dim=md, emb_dim=md, dim_mult=dm, kernel_size=5), 0).eval()
    clf = load_synth(amd_lib.HalfJannerUNet1d(H, D, out_dim=1, model_dim=md, emb_dim=md, dim_mult=tuple(dm), kernel_size=3), 1).eval()
    if md == 64:
        with pytest.raises(ValueError):
            P2.compile_guided2(net, clf, H, save_global=True)
    if (H, D) == (32, 23):
        # config 2: with the staging area capped as well, THREE trajectories fit one workgroup (49.6 KB each) -- what guided batches
        # above 512 run on (runtime2.guided_sample2)
        prog = P2.compile_guided2(net, clf, H, save_global=True, compact=True, max_stage=2304)
        assert prog.lds_bytes(3) <= 160 * 1024
    else:
        prog = P2.compile_guided2(net, clf, H, save_global=True, compact=True)
    assert prog.compact and prog.lds_bytes(1) <= 160 * 1024 and prog.prev_off < 0
    assert sum(int(op[P2.W2_KIND]) == P2.KIND2_LOADX for op in prog.ops) == 1
    hd4 = (H * D + 3) // 4 * 4
    assert prog.ws_floats > hd4, "multistep memory first, saved tensors behind it"
    assert all(int(op[P2.W2_SAVE]) >= hd4 for op in prog.ops if int(op[P2.W2_FLAGS]) & P2.F2_SAVE)
    g = torch.Generator().manual_seed(3)
    x, t = torch.randn(1, H, D, generator=g), torch.tensor([7])
    xr = x.clone().requires_grad_()
    clf._forward_torch(xr, t, None).sum().backward()
    with torch.no_grad():
        ref_pred = net._forward_torch(x, t, None)[0].numpy()
        row = emb_table(prog, None, [net.map_noise(t).numpy(), clf.map_noise(t).numpy()])[0]
    sim = LaneSim2(prog)
    for _ in range(1 if md == 64 else 2):                   # (second pass: state slot and arena reused)
        sim.poison_arena()
        sim.load_x(x[0].numpy())
        np.testing.assert_allclose(sim.run_forward(row), ref_pred, rtol=2e-5, atol=2e-5)
        ref_grad = xr.grad[0].numpy()
        np.testing.assert_allclose(sim.grad(), ref_grad, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(ref_grad).max())))


def test_lane_sim2_compact_program_for_three_trajectories(amd_lib):
    """The compact variant of the config-2 program (state and multistep memory outside LDS, block outputs written in place over their
    identity-residual input, capped staging area): three trajectories fit one workgroup's 160 KiB, and the lane-level twin still
    reproduces the reference's first forward."""
    name = "janner_cfg2_ddim"
    gold = np.load(golden_path(name))
    agent, _ = cases.build(amd_lib, name)
    c = cases.CASES[name]
    net = agent.model_ema["diffusion"]
    prog = P2.compile_janner2(net, c["horizon"], nw=8, compact=True, max_stage=2304)
    assert prog.compact and prog.lds_bytes(3) <= 160 * 1024 and prog.prev_off < 0 and prog.ws_floats >= 32 * 23
    # in-place outputs: some op writes the slot it reads its residual from
    assert any(int(op[P2.W2_FLAGS]) & P2.F2_RES and int(op[P2.W2_RES]) == int(op[P2.W2_DST]) and int(op[P2.W2_KPOST]) == 0 for op in prog.ops)
    inp, xt0 = _first_forward_inputs(name, agent)
    with torch.no_grad():
        temb = net.map_noise(_first_t(agent, c)).numpy()
    row = emb_table(prog, temb)[0]
    for b in range(2):
        sim = LaneSim2(prog)
        sim.load_x(xt0[b])
        np.testing.assert_allclose(sim.run_forward(row), gold["pred0"][b], rtol=2e-5, atol=2e-5)
        # a later forward: the state slot is arena memory that other tensors used meanwhile -- the solver step's rewrite of its
        # halo rows and pad channels is what makes it a valid conv source again
        sim.poison_arena()
        sim.load_x(xt0[b])
        np.testing.assert_allclose(sim.run_forward(row), gold["pred0"][b], rtol=2e-5, atol=2e-5)


def test_lane_sim2_compact_one_trajectory_program_long_horizon(amd_lib):
    """H = 128 (maze2d-style plans): the default LDS plan needs 171 KB, the compact one 144 KB -- one trajectory per workgroup.  The
    state slot sits high in the arena here and IS reused during the forward (the case that needs the halo rewrite)."""
    from cleandiffuser_amd.utils import load_synth
    H, D = 128, 6
    net = load_synth(amd_lib.JannerUNet1d(D, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5), 11).eval()
    with pytest.raises(ValueError):
        P2.compile_janner2(net, H, nw=8)
    prog = P2.compile_janner2(net, H, nw=8, compact=True)
    assert prog.compact and prog.lds_bytes(1) <= 160 * 1024 < prog.lds_bytes(2)
    g = torch.Generator().manual_seed(2)
    x, t = torch.randn(1, H, D, generator=g), torch.tensor([11])
    with torch.no_grad():
        ref = net._forward_torch(x, t, None)[0].numpy()
        row = emb_table(prog, net.map_noise(t).numpy())[0]
    sim = LaneSim2(prog)
    for _ in range(2):
        sim.poison_arena()
        sim.load_x(x[0].numpy())
        np.testing.assert_allclose(sim.run_forward(row), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("nw", [4, 8])
@pytest.mark.parametrize("scale", [True, False])
def test_lane_sim2_chiunet_against_module_forward(scale, nw, amd_lib):
    """ChiUNet1d with a global condition on the v2 program format: FiLM rows [scale | bias] per (step, trajectory), the 1x1 skip convs
    riding in their block's second conv, stride-2 down / transposed up convs -- against the module's own forward (bit-identical to the
    reference's, tests/test_module_mirrors.py)."""
    from cleandiffuser_amd.utils import load_synth
    from oracle.lane_sim2 import chi_film_rows
    H, act, obs, To = 16, 2, 5, 2
    net = load_synth(amd_lib.ChiUNet1d(act, obs, To, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2], kernel_size=5,
                                       cond_predict_scale=scale, obs_as_global_cond=True), 5).eval()
    prog = P2.compile_chiunet2(net, H, nw=nw)
    assert prog.lds_bytes(1) <= 160 * 1024 and prog.meta["cond_dim"] == To * obs
    g = torch.Generator().manual_seed(4)
    x, t, cond = torch.randn(2, H, act, generator=g), torch.tensor([7, 31]), torch.randn(2, To, obs, generator=g)
    with torch.no_grad():
        want = net(x, t, cond).numpy()
    rows = chi_film_rows(prog, net, t, cond)
    for b in range(2):
        sim = LaneSim2(prog)
        sim.load_x(x[b].numpy())
        np.testing.assert_allclose(sim.run_forward(rows[b]), want[b], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("shape", [(16, 6, [1, 2], 32), (32, 23, [1, 2, 2, 2], 32), (32, 69, [1, 2, 2, 2], 64)])
def test_lane_sim2_classifier_program_reproduces_log_p(shape, amd_lib):
    """The classifier's own program (engine/program2.py:compile_classifier2): forward ops + head, nothing saved, no backward ops;
    the head's forward value is the module's output (what CumRewClassifier.logp returns)."""
    from cleandiffuser_amd.utils import load_synth
    H, D, dm, md = shape
    clf = load_synth(amd_lib.HalfJannerUNet1d(H, D, out_dim=1, model_dim=md, emb_dim=md, dim_mult=tuple(dm), kernel_size=3), 1).eval()
    prog = P2.compile_classifier2(clf, H)
    assert prog.nw == 8 and prog.lds_bytes(1) <= 160 * 1024 and prog.ws_floats == 0 and len(prog.embtabs) == 1
    assert all(int(op[P2.W2_FLAGS]) & (P2.F2_SAVE | P2.F2_GNBWD | P2.F2_DUAL) == 0 for op in prog.ops)
    assert int(prog.ops[prog.meta["head_op"]][P2.W2_KIND]) == P2.KIND2_HEAD and prog.meta["head_op"] == len(prog.ops) - 1
    g = torch.Generator().manual_seed(5)
    x, t = torch.randn(2, H, D, generator=g), torch.tensor([7, 0])
    with torch.no_grad():
        want = clf._forward_torch(x, t, None).numpy()
        rows = emb_table(prog, clf.map_noise(t).numpy())
    for b in range(2):
        sim = LaneSim2(prog)
        sim.load_x(x[b].numpy())
        sim.run_forward(rows[b])
        np.testing.assert_allclose(sim.logp, want[b, 0], rtol=2e-5, atol=2e-5 * max(1.0, abs(float(want[b, 0]))))


def _mlp_nets(amd_lib):
    from cleandiffuser_amd.utils import load_synth
    import torch.nn as nn
    return {
        "pearce": (load_synth(amd_lib.PearceMlp(6, To=2, emb_dim=32, hidden_dim=64), 1), 6, 2 * 32, P2.compile_pearce_mlp2),
        "pearce256": (load_synth(amd_lib.PearceMlp(6, To=1, emb_dim=64, hidden_dim=256), 2), 6, 64, P2.compile_pearce_mlp2),
        # hidden 192: GroupNorm groups of 24 channels -> padded hidden layout (8 x 32), pad channels out of the variance
        "pearce192": (load_synth(amd_lib.PearceMlp(6, To=1, emb_dim=64, hidden_dim=192), 7), 6, 64, P2.compile_pearce_mlp2),
        "dql": (load_synth(amd_lib.DQLMlp(11, 3, emb_dim=16), 3), 3, 11, P2.compile_dql_mlp2),
        "dvinv": (load_synth(amd_lib.DVInvMlp(5, 3, emb_dim=16, hidden_dim=128), 4), 3, 10, P2.compile_dql_mlp2),
        "mlpnn": (load_synth(amd_lib.MlpNNDiffusion(5, emb_dim=16, hidden_dims=[64, 128], activation=nn.SiLU()), 5), 5, 16, P2.compile_mlp_nn2),
        "sfbc": (load_synth(amd_lib.SfBCUNet(4, emb_dim=32, hidden_dims=[128, 64, 64]), 6), 4, 32, P2.compile_sfbc_unet2),
    }


@pytest.mark.parametrize("tile", [4, 16])
@pytest.mark.parametrize("kind", ["pearce", "pearce256", "pearce192", "dql", "dvinv", "mlpnn", "sfbc"])
def test_lane_sim2_tile_mlp_programs_against_module_forward(kind, tile, amd_lib):
    """Batch-tiled MLP denoisers on the v2 program format (a tile of samples = the position axis, Linears = 1-tap convs, the
    time-dependent inputs folded into per-step bias rows, the condition in a context slot): conditional and zero-condition forwards
    against the modules' own forward (bit-identical to the reference's, tests/test_module_mirrors.py)."""
    from oracle.lane_sim2 import mlp_rows
    net, d, n_cond, compiler = _mlp_nets(amd_lib)[kind]
    net = net.eval()
    prog = compiler(net, tile)
    assert prog.nw == 8 and prog.lds_bytes(1) <= 160 * 1024 and prog.meta["mlp"]["cond_dim"] == n_cond
    assert int(prog.ops[0][P2.W2_KIND]) == P2.KIND2_LOADC
    g = torch.Generator().manual_seed(6)
    x, cond = torch.randn(tile, d, generator=g), torch.randn(tile, n_cond, generator=g)
    t = torch.full((tile,), 7)
    row = mlp_rows(prog, net, t[:1])[0]
    cshape = (tile, 2, 32) if kind == "pearce" else ((tile, 1, 64) if kind.startswith("pearce") else (tile, n_cond))
    with torch.no_grad():
        want_c = net(x, t, cond.reshape(cshape)).numpy()
        want_u = net(x, t, None).numpy() if kind != "dvinv" else None
    sim = LaneSim2(prog)
    sim.load_x(x.numpy())
    np.testing.assert_allclose(sim.run_forward(row, cond.numpy()), want_c, rtol=2e-5, atol=2e-5)
    if want_u is not None:
        sim.poison_arena()
        np.testing.assert_allclose(sim.run_forward(row, None), want_u, rtol=2e-5, atol=2e-5)


def test_program_accounting_matches_survey(amd_lib):
    """19.67 M MAC / sample / forward for the north-star config: 41 ops (the 1x1 skips ride in their block's second
    conv) and, for BASELINE config 3 (68.9 M parameters), 298.4 M MAC per forward."""
    agent, net = cases.build(amd_lib, "janner_cfg2_ddim")
    prog = P2.compile_janner2(net, 32, nw=8)
    assert prog.n_conv == 41
    assert abs(prog.macs_per_forward - 19.67e6) / 19.67e6 < 0.01
    chi = amd_lib.ChiUNet1d(2, 20, 2, model_dim=256, emb_dim=256, dim_mult=[1, 2, 2])
    try:
        big = P2.compile_chiunet2(chi, 16, nw=8)
    except ValueError:
        big = P2.compile_chiunet2(chi, 16, nw=8, compact=True)
    assert abs(big.macs_per_forward - 298.4e6) / 298.4e6 < 0.01


def test_lane_sim2_classifier_program_reproduces_reference_log_p(amd_lib):
    """The classifier's own program against the reference's log_p fixture (CumRewClassifier over the final trajectories, t = 0)."""
    name = "janner_cfg2_diffuser_logp"
    gold = np.load(golden_path(name))
    agent, _ = cases.build(amd_lib, name)
    clf = agent.classifier.model_ema
    prog = P2.compile_classifier2(clf, 32)
    with torch.no_grad():
        row = emb_table(prog, clf.map_noise(torch.zeros(1, dtype=torch.long)).numpy())[0]
    for b in range(3):
        sim = LaneSim2(prog)
        sim.load_x(gold["x_out"][b])
        sim.run_forward(row)
        np.testing.assert_allclose(sim.logp, gold["log_p"][b], rtol=2e-5, atol=2e-5)


def _twin_network(prog, module, kind):
    """net(x, t, cond) for oracle/step_sim.py that evaluates the program's lane-level twin: FiLM / bias rows from the step's timestep,
    one twin instance per trajectory (U-Nets) or per tile of samples (MLP programs)."""
    from oracle.lane_sim2 import mlp_rows

    def net(x, t, cond):
        out = np.zeros(tuple(x.shape), np.float32)
        if kind == "mlp":
            tile = prog.meta["mlp"]["tile"]
            row = mlp_rows(prog, module, t[:1])[0]
            xs = np.zeros((-(-x.shape[0] // tile) * tile, x.shape[1]), np.float32)
            xs[:x.shape[0]] = x.numpy()
            cs = None
            if cond is not None:
                cs = np.zeros((xs.shape[0], prog.meta["mlp"]["cond_dim"]), np.float32)
                cs[:x.shape[0]] = torch.flatten(cond, 1).numpy()
            for i in range(0, xs.shape[0], tile):
                sim = LaneSim2(prog)
                sim.load_x(xs[i:i + tile])
                y = sim.run_forward(row, None if cs is None else cs[i:i + tile])
                out[i:i + tile] = y[:min(tile, x.shape[0] - i)]
        elif kind == "chi":
            from oracle.lane_sim2 import chi_film_rows
            rows = chi_film_rows(prog, module, t, cond)        # one row per (step, trajectory): the condition is part of it
            for b in range(x.shape[0]):
                sim = LaneSim2(prog)
                sim.load_x(x[b].numpy())
                out[b] = sim.run_forward(rows[b])
        else:
            with torch.no_grad():
                row = emb_table(prog, module.map_noise(t[:1]).numpy())[0]
            for b in range(x.shape[0]):
                sim = LaneSim2(prog)
                sim.load_x(x[b].numpy())
                out[b] = sim.run_forward(row)
        return torch.from_numpy(out)
    return net


@pytest.mark.parametrize("name", ["janner_tiny_disc_ddim", "janner_tiny_disc_ddpm", "dqlmlp_ddpm", "pearce_cfg1_ddpm", "chiunet_cfg_w18_ddim",
                                  "chiunet_nofilmscale_sde"])
def test_whole_sampling_loop_through_the_program_twin(name, amd_lib, monkeypatch):
    """The complete device loop on the CPU: the step records the solver class hands to the kernel (captured at the dispatch hook)
    interpreted by oracle/step_sim.py, with the compiled program's lane-level twin as the network -- program, tables, step records and
    their sequencing together land on the real reference's samples (MLP programs in tiles of 4 with a ragged last tile; ChiUNet1d with
    the classifier-free-guidance pair against the zero condition)."""
    from cleandiffuser_amd.engine import dispatch
    from oracle import step_sim
    gold = np.load(golden_path(name))
    c = cases.CASES[name]
    agent, module = cases.build(amd_lib, name)
    inp = cases.make_inputs(name)
    seen = {}

    def capture(solver, model, plan, xt, prior, cond_vec, w_cfg, w_cg, requires_grad, feed):
        seen.update(plan=plan, xt=xt.clone(), cond=cond_vec, w_cfg=w_cfg)
        return None
    monkeypatch.setattr(dispatch, "try_fused_sample", capture)
    kw = cases.sample_kwargs(name, inp)
    n_draws = int(gold["n_draws"])
    agent.sample(torch.from_numpy(inp["prior"]), noise=list(inp["noise"][:n_draws]), **kw)
    mlp = c["net"][0] in ("DQLMlp", "PearceMlp")
    kind = "mlp" if mlp else ("chi" if c["net"][0] == "ChiUNet1d" else "unet")
    if mlp:
        compiler = P2.compile_dql_mlp2 if c["net"][0] == "DQLMlp" else P2.compile_pearce_mlp2
        prog = compiler(module, 4)
    elif kind == "chi":
        prog = P2.compile_chiunet2(module, cases.x_shape_of(c)[0], nw=8)
    else:
        prog = P2.compile_janner2(module, c["horizon"], nw=8)
    fm = torch.from_numpy(inp["fix_mask"])[None] if inp["fix_mask"] is not None else None
    x = step_sim.run_plan(seen["plan"], _twin_network(prog, module, kind), seen["xt"],
                          predict_noise=bool(agent.predict_noise), prior=torch.from_numpy(inp["prior"]), fix_mask=fm,
                          noise=[torch.from_numpy(v) for v in inp["noise"][1:n_draws]], cond=seen["cond"], w_cfg=seen["w_cfg"],
                          x_min=getattr(agent, "x_min", None), x_max=getattr(agent, "x_max", None))
    if getattr(agent, "clip_pred", False):
        x = x.clip(agent.x_min, agent.x_max)
    np.testing.assert_allclose(x.numpy(), gold["x_out"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("k", [2, 4])
@pytest.mark.parametrize("name", ["janner_cfg2_ddim", "janner_tiny_disc_ddim"])
def test_lane_sim2_split_program_reproduces_reference_forward(name, k, amd_lib, monkeypatch):
    """One trajectory over k workgroups (small-batch mode): every member runs the whole op list on its own copy of the activations but
    computes only its share of the row tiles / GroupNorm groups of the ops that can be cut; the twin steps the k member views in
    lockstep and performs the all-gathers.  Must land on the reference's first forward like the unsplit program."""
    from oracle.lane_sim2 import run_forward_split
    gold = np.load(golden_path(name))
    agent, _ = cases.build(amd_lib, name)
    c = cases.CASES[name]
    net = agent.model_ema["diffusion"]
    if name == "janner_cfg2_ddim":                       # default threshold: only the stream-bound layers are cut (an exchange costs ~3 k cycles)
        assert 8 <= sum(1 for op in P2.compile_janner2_split(net, c["horizon"], k).ops if op[P2.W2_XG]) <= 24
    monkeypatch.setattr(P2, "SPLIT_MIN_RECORDS", 0)      # here: cut everything that can be cut
    prog = P2.compile_janner2_split(net, c["horizon"], k)
    assert prog.lds_bytes(1) <= 160 * 1024 and prog.meta["split_k"] == k and len(prog.meta["member_ops"]) == k
    assert prog.ops_buffer[:prog.meta["member_ops"][0].size].reshape(prog.meta["member_ops"][0].shape).tolist() == prog.meta["member_ops"][0].tolist()
    n_split = sum(1 for op in prog.ops if op[P2.W2_XG])
    assert n_split >= (len(prog.ops) // 2 if name == "janner_cfg2_ddim" else 1)          # (16-channel layers are one row tile: not cut)
    inp, xt0 = _first_forward_inputs(name, agent)
    with torch.no_grad():
        temb = net.map_noise(_first_t(agent, c)).numpy()
    row = emb_table(prog, temb)[0]
    for b in range(2):
        sims = [LaneSim2(prog, member=m) for m in range(k)]
        for s in sims:
            s.load_x(xt0[b])
        np.testing.assert_allclose(run_forward_split(sims, row), gold["pred0"][b], rtol=2e-5, atol=2e-5)
        for s in sims[1:]:                               # every member ends with the same prediction
            np.testing.assert_array_equal(s.read_slot(prog.pred_off, prog.pred_stride, prog.horizon, prog.dim),
                                          sims[0].read_slot(prog.pred_off, prog.pred_stride, prog.horizon, prog.dim))


@pytest.mark.parametrize("k", [2, 4])
def test_lane_sim2_grouped_program_reproduces_reference_forward(k, amd_lib):
    """k trajectories over the k workgroups of a group (full-batch mode, VERDICT r3 'next' #2): the config-2 net's stream-bound layers at
    4 positions are GROUPED ops -- a member computes its 1/k of the output channels for all k trajectories (16 or 8 tile columns), the
    slots they touch hold the k trajectories side by side, the members all-gather after every grouped op and after the ordinary op that
    feeds the first one.  The twin steps the k member views in lockstep on k DIFFERENT trajectories of the reference fixture (on
    NaN-poisoned LDS: an unwritten halo row, pad channel or sub-slot shows up) and must land on the reference's first forward for each;
    a second forward on re-poisoned LDS must reproduce the first bit for bit."""
    from oracle.lane_sim2 import run_forward_group
    net = load_synth(amd_lib.JannerUNet1d(23, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5), 53)
    prog = P2.compile_janner2_group(net, 32, k)
    plain = P2.compile_janner2(net, 32, nw=8)
    assert prog.lds_bytes(1) <= 160 * 1024 and prog.meta["group_k"] == k and len(prog.meta["member_ops"]) == k
    n_g = sum(1 for op in prog.ops if int(op[P2.W2_XG]) & P2.XG_GOP)
    n_t = sum(1 for op in prog.ops if int(op[P2.W2_XG]) & P2.XG_TRAJ)
    assert n_g == 10 and n_t == 1, (n_g, n_t)            # the ten 0.6-1.3 MB layers at L = 4; the downsample conv that feeds them
    assert prog.macs_per_forward == plain.macs_per_forward
    g = torch.Generator().manual_seed(17)
    xs = 0.7 * torch.randn(k, 32, 23, generator=g)
    t = torch.tensor([11])
    with torch.no_grad():
        ref = net(xs, t.expand(k), None).numpy()
        row = emb_table(prog, net.map_noise(t).numpy())[0]
    sims = [LaneSim2(prog, member=m) for m in range(k)]
    for m, s in enumerate(sims):
        s.load_x(xs[m].numpy())
    outs = run_forward_group(sims, row)
    for m in range(k):
        np.testing.assert_allclose(outs[m], ref[m], rtol=2e-5, atol=2e-5, err_msg=f"member {m}")
    for s in sims:
        s.poison_arena()
    again = run_forward_group(sims, row)
    assert all(np.array_equal(a, b) for a, b in zip(outs, again))


@pytest.mark.parametrize("k", [4])           # (k = 2: the unguided grouped twin test above; the GPU suite runs both)
def test_lane_sim2_grouped_guided_program_prediction_and_gradient(k, amd_lib):
    """Round 6: the GROUPED GUIDED program (P2.compile_guided2_group) -- the denoiser's ten stream-bound layers grouped as in the
    unguided grouped program, every other op (the rest of the denoiser, the classifier's forward and backward ops with their saved
    tensors) on the member's own trajectory.  The twin steps the k member views in lockstep on k different trajectories (NaN-poisoned
    LDS): every member's prediction is the module's forward of ITS trajectory and its gradient slot torch.autograd's
    d classifier(x, t).sum() / d x of that trajectory; the classifier's descriptors carry no cut / exchange words."""
    from oracle.lane_sim2 import run_forward_group
    H, D = 32, 23
    net = load_synth(amd_lib.JannerUNet1d(D, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5), 53).eval()
    clf = load_synth(amd_lib.HalfJannerUNet1d(H, D, out_dim=1, model_dim=32, emb_dim=32, dim_mult=(1, 2, 2, 2), kernel_size=3), 54).eval()
    prog = P2.compile_guided2_group(net, clf, H, k)
    assert prog.lds_bytes(1) <= 160 * 1024 and prog.meta["group_k"] == k and len(prog.meta["member_ops"]) == k and prog.grad_off > 0
    n_den = prog.meta["n_den"]
    for ops in prog.meta["member_ops"]:
        assert sum(1 for op in ops[:n_den] if int(op[P2.W2_XG]) & P2.XG_GOP) == 10
        # behind the denoiser: ordinary ops only (W2_XG aliases W2_DST2 there: an LDS offset, none of the cut / exchange bits)
        assert all(int(op[P2.W2_XG]) & (P2.XG_XCHG | P2.XG_GOP | P2.XG_TRAJ) == 0 for op in ops[n_den:])
    g = torch.Generator().manual_seed(23)
    xs = 0.7 * torch.randn(k, H, D, generator=g)
    t = torch.tensor([9])
    xr = xs.clone().requires_grad_()
    clf._forward_torch(xr, t.expand(k), None).sum().backward()
    with torch.no_grad():
        ref = net._forward_torch(xs, t.expand(k), None).numpy()
        row = emb_table(prog, None, [net.map_noise(t).numpy(), clf.map_noise(t).numpy()])[0]
    sims = [LaneSim2(prog, member=m) for m in range(k)]
    for m, s in enumerate(sims):
        s.load_x(xs[m].numpy())
    outs = run_forward_group(sims, row)
    for m in range(k):
        np.testing.assert_allclose(outs[m], ref[m], rtol=2e-5, atol=2e-5, err_msg=f"member {m}: prediction")
        want = xr.grad[m].numpy()
        np.testing.assert_allclose(sims[m].grad(), want, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(want).max())), err_msg=f"member {m}: gradient")


@pytest.mark.parametrize("k", [4])           # (k = 2: the unguided split twin test; the GPU suite runs B = 100 on two workgroups per trajectory)
def test_lane_sim2_split_guided_program_prediction_and_gradient(k, amd_lib):
    """Round 6: the SMALL-BATCH guided program (P2.compile_guided2_split) -- one trajectory over k workgroups, the denoiser's ops cut by
    row tiles exactly as in the unguided split program, the classifier's forward / backward ops computed by every member on its own
    copy.  The twin steps the k member views in lockstep on ONE trajectory (NaN-poisoned LDS): every member ends with the module's
    prediction and with torch.autograd's d classifier(x, t).sum() / d x."""
    from oracle.lane_sim2 import run_forward_split
    H, D = 32, 23
    net = load_synth(amd_lib.JannerUNet1d(D, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5), 53).eval()
    clf = load_synth(amd_lib.HalfJannerUNet1d(H, D, out_dim=1, model_dim=32, emb_dim=32, dim_mult=(1, 2, 2, 2), kernel_size=3), 54).eval()
    prog = P2.compile_guided2_split(net, clf, H, k)
    plain = P2.compile_janner2_split(net, H, k)
    assert prog.lds_bytes(1) <= 160 * 1024 and prog.meta["split_k"] == k and len(prog.meta["member_ops"]) == k and prog.grad_off > 0
    n_den = prog.meta["n_den"]
    cut = [i for i, op in enumerate(prog.ops) if int(op[P2.W2_XG]) & P2.XG_XCHG]
    assert cut == [i for i, op in enumerate(plain.ops) if int(op[P2.W2_XG]) & P2.XG_XCHG] and max(cut) < n_den
    g = torch.Generator().manual_seed(29)
    x = 0.7 * torch.randn(1, H, D, generator=g)
    t = torch.tensor([5])
    xr = x.clone().requires_grad_()
    clf._forward_torch(xr, t, None).sum().backward()
    with torch.no_grad():
        ref = net._forward_torch(x, t, None)[0].numpy()
        row = emb_table(prog, None, [net.map_noise(t).numpy(), clf.map_noise(t).numpy()])[0]
    sims = [LaneSim2(prog, member=m) for m in range(k)]
    for s in sims:
        s.load_x(x[0].numpy())
    np.testing.assert_allclose(run_forward_split(sims, row), ref, rtol=2e-5, atol=2e-5)
    want = xr.grad[0].numpy()
    for m, s in enumerate(sims):
        np.testing.assert_allclose(s.read_slot(prog.pred_off, prog.pred_stride, H, D), ref, rtol=2e-5, atol=2e-5, err_msg=f"member {m}")
        np.testing.assert_allclose(s.grad(), want, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(want).max())), err_msg=f"member {m}: gradient")


def test_grouped_program_of_other_shapes_compiles_or_refuses_cleanly(amd_lib):
    """Nets whose deepest level does not offer a grouped op (too few channels for whole lane groups per member, a horizon whose deepest
    level is longer than 16 / k positions) must answer ValueError -- the signal that keeps a request on the ordinary program."""
    small = load_synth(amd_lib.JannerUNet1d(6, model_dim=16, emb_dim=16, dim_mult=[1, 2], kernel_size=5), 3)
    with pytest.raises(ValueError):
        P2.compile_janner2_group(small, 32, 4)           # deepest level: 16 positions x 32 channels
    with pytest.raises(ValueError):
        P2.compile_janner2_group(small, 32, 3)
