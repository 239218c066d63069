"""The line the driver parses (round 4's final stdout line was 25.9 KB and BENCH_r04.json has `parsed: null`).
bench.compact_line on a canned full record -- round 4's own 25.9-KB record, committed under profiles/ -- must give one
strictly-JSON object below the limit that still carries the contract's keys, `roofline` and `cpu_baseline`.  CPU only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def _canned():
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    assert len(json.dumps(full)) > 20000  # the record that did not parse
    return full


def _check(text):
    assert len(text.encode()) < 8192 and "\n" not in text
    assert "NaN" not in text and "Infinity" not in text
    j = json.loads(text, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    for k in CONTRACT:
        assert k in j, k
    return j


def test_compact_line_of_round_4s_record():
    import bench
    full = _canned()
    full["prefilter_dtype"] = bench.prefilter_dtype_of(full["roofline"])
    j = _check(bench.compact_line(full))
    assert j["value"] == float("%.6g" % full["value"]) and j["n_gpus"] == 1 and j["metric"] == "queries/sec"
    r = j["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "launch_ms", "hbm_frac"):
        assert k in r, k
    assert r["kernel"].startswith("scan_mfma_kernel") and "f16" in j["prefilter_dtype"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "queries/s" and c["sample"]
    assert j["config"]["workload"].startswith("cfg3") and "model" not in j["config"]
    assert set(j["other_configs"]) >= {"cfg2", "cfg4_one_of_8_shards", "cfg5_one_of_8_shards", "reference_default_options", "scale64m_n1"}
    assert j["host_buffers_qps"] > 0


def test_non_finite_values_and_oversized_parts():
    import bench
    full = _canned()
    full["roofline"]["frac"] = float("nan")
    full["roofline"]["traffic"] = float("inf")
    full["stage_ms_per_batch"]["hash"] = float("-inf")
    full["cpu_baseline"]["sample"] = "x" * 5000
    full["config"]["workload"] = "cfg3: " + "y" * 3000
    full["other_configs"] = {("cfg%d" % i): dict(full["other_configs"]["cfg2"]) for i in range(80)}  # an optional part that cannot fit
    j = _check(bench.compact_line(full))
    assert j["roofline"]["frac"] is None and j["roofline"]["traffic"] is None
    assert "other_configs" in j["dropped_for_length"] and "other_configs" not in j
    assert len(j["cpu_baseline"]["sample"]) <= 400


def test_detail_record_is_strict_json(tmp_path, monkeypatch):
    import bench
    full = _canned()
    full["x"] = float("nan")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    doc, where = bench.write_detail(full)
    assert "NaN" not in doc and json.loads(doc)["x"] is None
    assert sorted(where) == ["bench_detail.json", os.path.join("gpurun_out", "bench_detail.json")]
    assert json.load(open(tmp_path / "gpurun_out" / "bench_detail.json"))["value"] > 0
