"""bench.py's output contract: the LAST stdout line is a compact headline object of at most 4096 bytes -- round 4's 20 KB
line overflowed the driver's ~8 KB tail and the round's headline went unparsed.  Everything else travels on earlier `[leg]` / `[detail]` lines
and in gpurun_out/bench_detail.json.  CPU only: the full object of a real round-4 run (profiles/r04am/bench.json) is the input."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
             "config", "roofline", "cpu_baseline", "value_cold", "parity_rel_rms_vs_oracle", "secondary")


def _full():
    return json.load(open(os.path.join(ROOT, "profiles", "r04am", "bench.json")))


def test_final_line_is_compact_and_complete():
    full = _full()
    assert len(json.dumps(full)) > 15000                      # the object that broke the driver's parser
    line = bench.compact_line(full)
    assert len(line) <= bench.LINE_LIMIT == 4096 and "\n" not in line
    o = json.loads(line)
    for k in HEAD_KEYS:
        assert k in o, k
    assert o["value"] == float(f"{full['value']:.6g}") and o["n_gpus"] == 1 and o["dtype"] == "f32"
    for k in ("workload", "T", "P", "C", "L", "fs", "entry_point", "distributed"):
        assert k in o["config"], k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "kernel", "compute"):
        assert k in o["roofline"], k
    assert abs(o["roofline"]["frac"] - o["roofline"]["achieved"] / o["roofline"]["peak"]) < 1e-5
    assert "frac" in o["roofline"]["compute"]
    for k in ("value", "cores", "kind", "seconds_measured", "sample"):
        assert k in o["cpu_baseline"], k
    assert set(o["secondary"]) == set(full["secondary"])
    for name, row in o["secondary"].items():                  # flat rows only
        assert all(not isinstance(v, (dict, list)) for v in row.values()), name
        for k in ("workload", "value", "ms_per_step", "roofline_frac", "cpu_baseline_value", "parity"):
            assert k in row, (name, k)
    assert o["secondary"]["cfg5"]["roofline_frac"] == float(f"{full['secondary']['cfg5']['roofline']['frac']:.4g}")


def test_line_never_exceeds_the_limit_whatever_the_legs_carry():
    full = _full()
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    for i in range(12):                                       # more legs than any run prints
        full["secondary"][f"extra_leg_{i}"] = dict(full["secondary"]["cfg5"], workload="x" * 900)
    full["secondary"]["broken"] = {"error": "RuntimeError('" + "e" * 3000 + "')", "traceback": "t" * 1500}
    line = bench.compact_line(full)
    assert len(line) <= 4096
    o = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "config", "roofline", "cpu_baseline"):
        assert k in o, k


def test_multi_gpu_lines_stay_compact():
    """the N > 1 objects (rank 0 of `bench.py --gpus 8` for cfg2 and `--config cfg4`): no secondary legs, but `distributed`, the gather and the
    per-rank-max timing ride along"""
    full = _full()
    full.pop("secondary")
    full["n_gpus"] = 8
    full["config"]["distributed"] = {"backend": "nccl (= RCCL on ROCm)", "rccl_version": "2.22.3", "ranks": 8, "distinct_gpus": 8}
    full["config"]["gather"] = "every 5th render of every rank to rank 0, overlapped with the next renders"
    full["config"]["gathered_bytes_at_root"] = 8 * 4 * 30720000
    line = bench.compact_line(full)
    o = json.loads(line)
    assert len(line) <= 4096 and o["n_gpus"] == 8 and o["config"]["distributed"]["ranks"] == 8 and o["config"]["gathered_bytes_at_root"] > 0
    scene = dict(full["secondary"]["cfg4_per_gpu_share"]) if "secondary" in full else _full()["secondary"]["cfg4_per_gpu_share"]
    scene["n_gpus"] = 8
    scene["config"]["distributed"] = full["config"]["distributed"]
    scene["gather_verification"] = {"scenes_rerendered_by_rank0": [0, 63, 64, 127, 256, 319, 448, 511], "mismatching": [], "same_bits": True, "shard_sizes": [64] * 8}
    line = bench.compact_line(scene)
    o = json.loads(line)
    assert len(line) <= 4096 and o["unit"] == "scene-sec/sec" and o["gather_same_bits"] is True
    assert o["config"]["scenes_total"] and o["roofline"]["stages_ms"] and "scene_frac" in o["roofline"]


def test_emit_prints_details_first_and_the_headline_last(tmp_path):
    full = _full()
    buf = io.StringIO()
    with redirect_stdout(buf):
        line = bench.emit(full, detail_dir=str(tmp_path))
    lines = buf.getvalue().splitlines()
    assert lines[-1] == line and len(lines[-1]) <= 4096
    assert json.loads(lines[-1])["metric"] == full["metric"]
    legs = [ln for ln in lines[:-1] if ln.startswith("[leg] ")]
    assert len(legs) == len(full["secondary"])
    assert sum(ln.startswith("[detail] ") for ln in lines[:-1]) == 1
    assert all(not ln.startswith("{") for ln in lines[:-1])   # exactly ONE line of stdout is a JSON object
    saved = json.load(open(tmp_path / "bench_detail.json"))
    assert saved["roofline"]["traffic_details"] == full["roofline"]["traffic_details"]
    # the driver keeps the tail of stdout: the headline must sit entirely inside the last 8 KB
    tail = buf.getvalue()[-8192:]
    assert json.loads(tail.splitlines()[-1])["value"] == json.loads(line)["value"]
