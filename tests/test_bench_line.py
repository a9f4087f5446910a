"""The ONE line bench.py prints must be small, strict JSON with the contract's keys (the 21 KB line of that round could not
be parsed by the driver).  CPU only: the line assembler runs on the committed detail record of an earlier GPU run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DETAIL = os.path.join(ROOT, "profiles", "r05_bench_n1.json")


def _strict(text):
    def fail(x):
        raise ValueError("non-finite constant in the bench line: " + x)

    return json.loads(text, parse_constant=fail)


def test_dry_line_is_small_strict_json_with_the_contract_keys():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-line", DETAIL], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    assert len(lines[0].encode()) < 8192
    d = _strict(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert "workload" in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches", "avg_launch_us", "algorithmic_bytes_per_launch"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-4
    for k in ("scans_per_sec", "free_running", "host_seam", "patched_reference", "shipped_configuration", "cpu_baseline", "parity_vs_cpu"):
        assert k in d["scans_per_sec"], k
    for leg in ("m1_f64", "m1_gicp", "m1_large_map"):
        assert isinstance(d[leg]["value"], float)


def test_line_survives_nan_numpy_scalars_and_oversize():
    sys.path.insert(0, ROOT)
    import numpy as np

    import bench

    out = json.load(open(DETAIL))
    out["value"] = np.float64(out["value"])
    out["roofline"]["traffic"] = float("nan")
    out["m1_gicp"] = {"error": "x" * 5000}
    out["scans_per_sec"]["free_running"]["scans_per_sec"] = float("inf")
    d = _strict(bench.compact_line(out))
    assert d["roofline"]["traffic"] is None and d["scans_per_sec"]["free_running"] is None
    assert len(d["m1_gicp"]["error"]) <= 120
    # a config string of absurd length is cut, never printed whole
    out["config"]["workload"] = "w" * 20000
    assert len(bench.compact_line(out)) < bench.LINE_LIMIT


def test_nothing_but_the_line_reaches_stdout():
    """Libraries write to stdout on their own (RCCL's version banner comes out of C stdio at process exit, behind the line): bench.py sets
    file descriptor 1 aside, points it at stderr for the run and writes the one line to the real stdout."""
    code = (
        "import sys, json, ctypes; sys.path.insert(0, %r); import bench\n"
        "bench.claim_stdout()\n"
        "print('a library banner through Python')\n"
        "libc = ctypes.CDLL(None); libc.puts(b'a library banner through C stdio (buffered until exit)')\n"
        "bench.emit(json.load(open(%r)))\n"
        "libc.puts(b'and one more at exit')\n" % (ROOT, DETAIL))
    env = dict(os.environ, O3DS_BENCH_DETAIL=os.devnull)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert len(lines) == 1, p.stdout
    assert _strict(lines[0])["metric"] == "icp_iterations_per_sec"
    assert "banner" in p.stderr
