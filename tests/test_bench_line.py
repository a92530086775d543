"""The driver keeps an 8 KB tail of bench.py's stdout: the ONE JSON line must fit it (the 21 KB line
of round 5 came back `parsed: null`).  Input = the full round-5 record committed under profiles/ (the record that did
not parse), pushed through bench.compact_line; plus the worst case of an 8-rank record."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _full_record():
    with open(os.path.join(ROOT, "profiles", "r05fin_bench.json")) as f:
        return json.load(f)


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "detail_file")


def test_round5_record_compacts_under_the_cap():
    bench = _bench()
    res = _full_record()
    assert len(json.dumps(res)) > 20000                      # the record that did not parse
    line = json.dumps(bench.compact_line(res))
    assert len(line) < bench.LINE_TARGET, len(line)
    back = json.loads(line)
    for k in REQUIRED:
        assert k in back, k
    assert back["value"] == float(f"{res['value']:.5g}")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
        assert k in back["roofline"], k
        assert k in back["roofline_view_gather_attention"], k
    assert abs(back["roofline"]["frac"] - res["roofline"]["frac"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert len(back["cpu_baseline"]["sample"]) <= 200
    assert set(back["workloads_ms"]) == set(res["workloads"])
    assert all(isinstance(v, float) for v in back["workloads_ms"].values())
    assert len(back["kernels_ms_per_step"]) == 8
    assert "workload" in back["config"] and "model" not in back["config"]


def test_eight_rank_record_stays_under_the_hard_cap():
    bench = _bench()
    res = _full_record()
    res["n_gpus"] = 8
    res["collective"] = {"per_rank_ms_per_step": [10.123456789] * 8, "standin_allreduce_ms": [1.23456789] * 8,
                         "standin_exposed_ms": [0.0] * 8, "pooling_bucket_allreduce_ms": [0.1] * 8,
                         "pooling_bucket_exposed_ms": [0.0] * 8, "ranks_devices": [[i, i] for i in range(8)],
                         "note": "x" * 500}
    res["allreduce_ms"], res["exposed_ms"] = 1.23456789, 0.0
    res["config"]["gradient_allreduce"] = {"pooling_parameters_bytes": 12345, "standin_bucket_MB": 112.0, "note": "y" * 400}
    # a future field that overgrows: the optional keys are shed, the contract keys stay
    res["workloads"] = {f"workload_{i:03d}_with_a_long_name": {"ms_per_step": 1.0 + i} for i in range(200)}
    line = json.dumps(bench.compact_line(res))
    assert len(line) < bench.LINE_HARD_CAP, len(line)
    back = json.loads(line)
    for k in REQUIRED:
        assert k in back, k
    assert "note" not in back["config"]["gradient_allreduce"]


def test_emit_writes_detail_and_one_line(tmp_path, monkeypatch):
    bench = _bench()
    res = _full_record()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    r, w = os.pipe()
    bench.emit(res, w)
    os.close(w)
    out = os.read(r, 1 << 16).decode()
    os.close(r)
    assert out.endswith("\n") and out.count("\n") == 1
    assert len(out) < bench.LINE_TARGET
    with open(tmp_path / bench.DETAIL_FILE) as f:
        assert json.load(f) == res                            # nothing is lost: the full record is next to the line
