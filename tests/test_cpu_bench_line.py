"""bench.py's final stdout line is what the driver parses: ONE JSON object under 4 KB that carries `roofline` and `cpu_baseline`
(round 3's 21 KB line left BENCH_r03.json with parsed = null).  main() runs here with every GPU leg replaced by the detail of a
real run (profiles/r03c_cdu_b100000_bench.json), so the assembly of the line and its size limit are what is tested."""
import glob
import json
import os
import re
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Obj:
    def __getattr__(self, name):
        return lambda *a, **k: None


def _fake_main(monkeypatch, tmp_path, capsys, argv):
    import bench
    detail = json.load(open(os.path.join(ROOT, "profiles", "r03c_cdu_b100000_bench.json")))
    ctx = types.SimpleNamespace(rank=0, world=1, comm=None, lib=_Obj(), sync=lambda: None, max_over_ranks=lambda v: v)
    monkeypatch.setattr(bench, "Ctx", lambda args: ctx)
    monkeypatch.setattr(bench, "start_workers", lambda n: None)

    def fake_bench_qp(ctx_, workload, B, steps, warmup, sx, **kw):
        src = detail if workload == "cdu" else detail["configs"]["cstrs_10k"]
        res = {k: src[k] for k in ("value", "unit", "ms_per_step", "solver", "roofline", "roofline_step", "time_shares", "dtype")}
        res["roofline_gemm_group"] = detail["roofline"]
        res["survey_model"] = bench.survey_model(4480, 284, res["value"])
        h = dict(qp=_Obj(), buf=_Obj(), pl={}, P=np.eye(4), tq=np.ones((4, 2)), nu=2, N=2, n=4, host=(np.zeros((64, 2)),) * 4)
        return res, h
    monkeypatch.setattr(bench, "bench_qp", fake_bench_qp)
    monkeypatch.setattr(bench, "parity_leg", lambda h, a, b: detail["parity"])
    monkeypatch.setattr(bench, "pdip_leg", lambda *a: detail["pdip_path"])
    monkeypatch.setattr(bench, "host_io_leg", lambda *a: detail["host_io"])
    monkeypatch.setattr(bench, "first_move_leg", lambda *a: detail["first_move_output"])
    monkeypatch.setattr(bench, "sweep_leg", lambda *a: detail["sweep_sx"])
    monkeypatch.setattr(bench, "chains_leg", lambda *a: detail["chains"])
    monkeypatch.setattr(bench, "chains_task_leg", lambda *a: detail["chains_task"])
    monkeypatch.setattr(bench, "unstable_leg", lambda *a: {"value": 1.0e6, "ms_per_step": 16.0, "window": 640, "farfield_rank": 256, "note": "x" * 3000})
    monkeypatch.setattr(bench, "bench_nn", lambda *a, **k: detail["configs"]["nn_1m"])
    monkeypatch.setattr(bench, "cpu_baseline", lambda *a, **k: dict(detail["cpu_baseline"], sample="y" * 2000))
    monkeypatch.setenv("NNMPC_BENCH_DETAIL", str(tmp_path / "detail.json"))
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    bench.main()
    return capsys.readouterr().out.strip().splitlines()


def test_final_line_is_small_and_carries_roofline_and_cpu_baseline(monkeypatch, tmp_path, capsys):
    lines = _fake_main(monkeypatch, tmp_path, capsys, ["--steps", "20", "--warmup", "5"])
    last = lines[-1]
    assert len(last) < 4096, len(last)
    line = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1 and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and len(r["kernel"]) <= 120
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and len(c["sample"]) <= 160
    assert line["parity"]["active_set_hamming"] == 0
    assert line["roofline_step"]["frac"] > 0 and line["survey_model_TFLOPs"] > 0
    assert set(line["configs"]) >= {"cstrs_10k", "nn_1m"}
    # the detail went to the side file, whole
    d = json.load(open(tmp_path / "detail.json"))
    assert "leg_seconds" in d and "sweep_sx" in d and d["value"] == line["value"] or abs(d["value"] - line["value"]) / d["value"] < 1e-5


def test_compact_line_never_exceeds_the_limit():
    import bench
    big = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "w" * 5000}, "roofline": {"kernel": "k" * 5000, "bound": "mfma", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s", "frac": 0.5,
                                                              "traffic": None, "time_share": {"a": 0.123456789}},
           "cpu_baseline": {"value": 1.0, "unit": "solves/s", "cores": 8, "kind": "port", "sample": "s" * 5000},
           "sweep_sx": {f"sx={i}": {"value": float(i)} for i in range(400)}}
    s = bench.compact_line(big, "bench_detail.json")
    assert len(s) < 4096
    d = json.loads(s)
    assert "roofline" in d and "cpu_baseline" in d


def test_gpu_tests_never_fork_after_hip_init():
    """No fork() anywhere a GPU test can reach: oracle workers are fresh interpreters (tests/helpers.py, tests/oracle_worker.py)."""
    pat = re.compile(r"get_context\(\s*[\"']fork[\"']\s*\)|os\.fork\(|\bmp\.Pool\(|multiprocessing\.Pool\(|os\.exec[lv]")
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "*_gpu.py"))) + [os.path.join(ROOT, "tests", "helpers.py"), os.path.join(ROOT, "tests", "oracle_worker.py")]:
        src = "\n".join(l for l in open(f).read().splitlines() if not l.lstrip().startswith("#"))
        src = re.sub(r'"""(?:.|\n)*?"""', "", src)
        assert not pat.search(src), f
