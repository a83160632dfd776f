"""The committed bench line (profiles/round4_bench_n1.json, written by `python bench.py` on the GPU box) carries every field of the
driver's contract -- and, since round 4, the other BASELINE configs as `secondary` legs and the arithmetic as a string -- and bench.py's command line
parses the driver's invocation."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    line = json.load(open(os.path.join(ROOT, "profiles", "round4_bench_n1.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
    assert line["metric"] == "pairs_per_sec" and line["unit"] == "pairs/s" and line["n_gpus"] == 1 and line["higher_is_better"] is True
    assert line["scaling"] == "weak" and line["vs_baseline"] is None and line["data"] == "synthetic" and "workload" in line["config"]
    assert abs(line["value"] - 64 * line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) < 1e-6 * line["value"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("reference", "port") and cpu["value"] > 0 and cpu["cores"] >= 1
    assert line["parity"]["R_err_rad_max"] < 1e-5 and line["parity"]["t_err_max"] < 1e-5 and line["parity"]["pairs_checked"] == 64
    # the arithmetic travels in strings (the driver's parsed record drops nested dicts): dtype names the split and the reduced-term layers
    assert isinstance(line["dtype"], str) and "f16x3" in line["dtype"] and "similarity=1" in line["dtype"]
    assert "arithmetic:" in line["config"]["workload"] and line["config"]["term_budget"] == {"similarity": 1}
    # the other BASELINE configs, observed by the same command: configs[2] / [3] shapes, the headline workload on the sharp weight family and the configs[4]
    # training step, each with a roofline fraction and a parity sample
    legs = line["secondary"]
    assert len(legs) == 4 and all("error" not in leg for leg in legs)
    for leg, tag in zip(legs, ("configs[2]", "configs[3]", "SHARP family", "configs[4]")):
        assert tag in leg["workload"] and leg["value"] > 0 and 0 < leg["roofline"]["frac"] < 1 and leg["parity"]["R_err_rad_max"] < 1e-5


def test_bench_command_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--precision"):
        assert flag in out.stdout
