"""The line bench.py prints is the record the driver parses: one JSON object, < 4 KB whatever was measured
(the round-5 line had grown to 19.4 KB and the driver, which keeps a bounded tail of stdout, lost its head:
BENCH_r05.json.parsed == null).  Convention matched: the reference's benchmarks emit one small record per
measurement (/root/reference/tests/perf/test_benchmark.py, /root/reference/profiler/profiling_utils.py)."""
import io
import json
import os
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _full_record():
    # the largest record this script has produced (round 5, every variant measured)
    txt = open(os.path.join(ROOT, "profiles", "r05_bench.json")).read().strip().splitlines()[-1]
    return json.loads(txt)


def test_compact_line_of_the_largest_record_is_under_4k_and_keeps_what_the_driver_reads():
    import bench
    full = _full_record()
    assert len(json.dumps(full)) > 15000
    line = json.dumps(bench.compact(full), separators=(",", ":"))
    assert len(line) < 4096
    rec = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["value"] == float("%.6g" % full["value"])
    rf = rec["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5
    assert rf["traffic"] and rf["traffic_source"].startswith("profiles/") and " " not in rf["traffic_source"]
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] >= 1
    # the two NUTS blocks, as numbers
    assert rec["secondary"]["value"] > 0 and 0 < rec["secondary"]["roofline"]["frac"] < 1
    runs = rec["secondary_model_nuts"]["runs"]
    assert set(runs) == set(full["secondary_model_nuts"]["runs"])
    for r in runs.values():
        assert r["value"] > 0 and "max_r_hat" in r and "frac" in r["roofline"]
    # no prose: every string value short
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(x) for x in strings({k: v for k, v in rec.items() if k != "config"})) <= 64
    assert len(rec["config"]["workload"]) <= 900


def test_emit_prints_exactly_one_line_and_writes_the_full_record(tmp_path, monkeypatch):
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "gpurun_out").mkdir()
    full = _full_record()
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(full)
    lines = buf.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096
    assert json.loads(lines[0])["full_record"] == "bench_full.json"
    for d in (tmp_path, tmp_path / "gpurun_out"):
        assert json.loads((d / "bench_full.json").read_text()) == full


def test_reference_cpu_fixture_is_the_unmodified_reference_and_is_printed_beside_the_port():
    import bench
    ref = bench.reference_cpu_record()
    assert ref is not None and ref["kind"] == "reference" and ref["cores"] >= 1
    assert 0.1 < ref["value"] < 100 and 0.1 < ref["validated"] < 100
    j = json.load(open(os.path.join(ROOT, ref["source"])))
    assert "unmodified reference pyro 1.9.1" in j["what"] and j["how"] == "tools/time_reference_cpu.py"
