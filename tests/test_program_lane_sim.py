"""Program compiler (weight packing, LDS plan, op flags) proven on CPU: the lane-level model of the kernel
(oracle/lane_sim.py) interprets the compiled program and must reproduce the reference's first forward (``pred0`` in
the fixtures, produced by the real reference).  Tolerance 2e-5: same fp32 math, different summation order."""
import numpy as np
import pytest
import torch

from cleandiffuser_amd.engine import program as P
from oracle import cases
from oracle.lane_sim import LaneSim
from conftest import golden_path


def _first_forward_inputs(name, agent):
    c = cases.CASES[name]
    inp = cases.make_inputs(name)
    temp = c["sample"].get("temperature", 1.0)
    xt0 = inp["noise"][0] * np.float32(temp)
    if inp["fix_mask"] is not None:
        xt0 = xt0 * (1 - inp["fix_mask"][None]) + inp["prior"] * inp["fix_mask"][None]
    return inp, xt0.astype(np.float32)


@pytest.mark.parametrize("name,edm", [("janner_cfg2_ddim", False), ("janner_h4_ddpm", True), ("janner_tiny_disc_ddim", True),
                                      ("janner_tiny_disc_ddim", False), ("janner_tiny_cond_w1", False),
                                      ("janner_tiny_cont_ddim", True), ("janner_h64_single", False)])
def test_lane_sim_reproduces_reference_forward(name, edm, amd_lib):
    """`edm`: with / without the two EDM-only dense state buffers in the LDS plan (the runtime compiles the lean plan unless the
    step plan needs them) -- the interpreter works off the plan's offsets, so an overlap would show up here."""
    gold = np.load(golden_path(name))
    agent, net = cases.build(amd_lib, name)
    c = cases.CASES[name]
    prog = P.compile_janner(agent.model_ema["diffusion"], c["horizon"], edm=edm)
    assert prog.lds_floats * 4 <= 160 * 1024
    inp, xt0 = _first_forward_inputs(name, agent)
    # timestep-embedding row exactly as the solver would hand it to the kernel
    from cleandiffuser_amd.engine import plan as _plan
    S = c["sample"]["sample_steps"]
    if c["solver"][0] == "DiscreteDiffusionSDE":
        from cleandiffuser_amd.utils import SUPPORTED_SAMPLING_STEP_SCHEDULE as SS
        sched = SS[c["sample"].get("sample_step_schedule", "uniform")](agent.diffusion_steps, S)
        t = torch.tensor([int(sched[S])], dtype=torch.long)
    else:
        from cleandiffuser_amd.utils import SUPPORTED_SAMPLING_STEP_SCHEDULE as SS
        sched = SS[c["sample"].get("sample_step_schedule", "uniform_continuous")](agent.t_diffusion, S)
        t = torch.tensor([float(sched[S])], dtype=torch.float32)
    temb = agent.model_ema["diffusion"].map_noise(t)[0].numpy()
    nb = 2 if name == "janner_cfg2_ddim" else c["batch"]
    for b in range(nb):
        sim = LaneSim(prog)
        sim.load_x(xt0[b])
        cond = inp["cond"][b] if inp["cond"] is not None else None
        pred = sim.run_forward(temb, cond)
        np.testing.assert_allclose(pred, gold["pred0"][b], rtol=2e-5, atol=2e-5)


def test_program_accounting_matches_survey(amd_lib):
    """19.67 M MAC / sample / forward for the north-star config, 47 conv-type ops."""
    agent, net = cases.build(amd_lib, "janner_cfg2_ddim")
    prog = P.compile_janner(net, 32)
    assert prog.n_conv == 47
    assert abs(prog.macs_per_forward - 19.67e6) / 19.67e6 < 0.01


def test_lane_sim_reproduces_classifier_logp(amd_lib):
    """HalfJannerUNet1d program (flatten op, raw-embedding copy, vector head) vs the reference's log_p fixture."""
    name = "janner_cfg2_diffuser_logp"
    gold = np.load(golden_path(name))
    agent, _ = cases.build(amd_lib, name)
    clf = agent.classifier.model_ema
    prog = P.compile_half_janner(clf, 32)
    temb = clf.map_noise(torch.zeros(1, dtype=torch.long))[0].numpy()
    for b in range(3):
        sim = LaneSim(prog)
        sim.load_x(gold["x_out"][b])
        np.testing.assert_allclose(sim.run_forward(temb), gold["log_p"][b], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("tile", [4, 8, 16])
@pytest.mark.parametrize("kind", ["pearce", "pearce192", "dql", "sfbc", "mlpnn"])
def test_lane_sim_reproduces_mlp_tile_programs(kind, tile, amd_lib):
    """Batch-tiled MLP programs (sample index on the MFMA column axis, per-sample GroupNorm, GELU/Mish/LeakyReLU,
    pre-scaled skips, context slot) against the module forward, which is bit-identical to the reference's."""
    from cleandiffuser_amd.utils import load_synth
    S = tile            # samples per workgroup: 16 = one 16x16x4 column tile; 4 / 8 = 4x4x1 column blocks (runtime.mlp_tile)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(S, 6, generator=g)
    t = torch.full((S,), 13, dtype=torch.long)
    if kind.startswith("pearce"):                     # hidden 192: GroupNorm groups of 24 channels (not a power of two)
        net = load_synth(amd_lib.PearceMlp(6, 1, emb_dim=64, hidden_dim=192 if kind == "pearce192" else 256)).eval()
        prog = P.compile_pearce_mlp(net, S)
        cond = torch.randn(S, 1, 64, generator=g)
        temb = np.concatenate([net.map_noise(t[:1])[0].numpy(), [13.0]]).astype(np.float32)
    elif kind == "sfbc":                              # Linear / SiLU residual blocks, concat skips, context = t_layer(temb) + condition
        net = load_synth(amd_lib.SfBCUNet(6, emb_dim=32, hidden_dims=[128, 64, 32])).eval()
        prog = P.compile_sfbc_unet(net, S)
        cond = torch.randn(S, 32, generator=g)
        with torch.no_grad():
            temb = net.t_layer(net.map_noise(t[:1]))[0].numpy()
    elif kind == "mlpnn":                             # plain MLP over [x | map_noise(t) + condition]
        net = load_synth(amd_lib.MlpNNDiffusion(6, emb_dim=16, hidden_dims=[64, 128], activation=torch.nn.SiLU())).eval()
        prog = P.compile_mlp_nn(net, S)
        cond = torch.randn(S, 16, generator=g)
        with torch.no_grad():
            temb = net.map_noise(t[:1])[0].numpy()
    else:
        net = load_synth(amd_lib.DQLMlp(17, 6)).eval()
        prog = P.compile_dql_mlp(net, S)
        cond = torch.randn(S, 17, generator=g)
        temb = net.map_noise(t[:1])[0].numpy()
    with torch.no_grad():
        ref = net(x, t, cond).numpy()
    sim = LaneSim(prog)
    sim.load_x(x.numpy(), cond.flatten(1).numpy())
    np.testing.assert_allclose(sim.run_forward(temb), ref, rtol=2e-5, atol=2e-5)


def test_lane_sim_reproduces_chiunet_forward(amd_lib):
    """ChiUNet1d program: FiLM (scale, bias) epilogue, just-in-time per-block FiLM vectors, raw condition load."""
    from cleandiffuser_amd.utils import load_synth
    net = load_synth(amd_lib.ChiUNet1d(2, 20, 2, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2])).eval()
    prog = P.compile_chiunet(net, 16)
    g = torch.Generator().manual_seed(1)
    x, c, t = torch.randn(2, 16, 2, generator=g), torch.randn(2, 2, 20, generator=g), torch.tensor([3, 7])
    with torch.no_grad():
        ref, temb = net(x, t, c).numpy(), net.map_noise(t).numpy()
    for b in range(2):
        sim = LaneSim(prog)
        sim.load_x(x[b].numpy())
        np.testing.assert_allclose(sim.run_forward(temb[b], c[b].flatten().numpy()), ref[b], rtol=2e-5, atol=2e-5)


def test_config3_program_fits_one_workgroup(amd_lib):
    """BASELINE config 3 (68.9 M parameters, 298.4 M MAC per forward) compiles into <= 160 KiB of LDS."""
    net = amd_lib.ChiUNet1d(2, 20, 2, model_dim=256, emb_dim=256, dim_mult=[1, 2, 2])
    prog = P.compile_chiunet(net, 16)
    assert prog.lds_floats * 4 <= 160 * 1024
    assert abs(prog.macs_per_forward - 298.4e6) / 298.4e6 < 0.01
