"""CPU: the parts of bench.py that do not need a GPU (argument contract, CPU baseline legs on tiny samples)."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_defaults_follow_the_driver_contract(monkeypatch):
    b = load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0
    assert (a.boards, a.clusters, a.tree, a.dtype, a.opp) == (9216, 1000, "river", "i32", "full")      # BASELINE configs[1]
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 2)


def test_cpu_baseline_legs_run_on_a_tiny_sample():
    b = load_bench()
    ref = b.cpu_baseline(50, "clamp", 0.2)
    tuned = b.cpu_baseline_tuned(50, "clamp", 0.2)
    for r in (ref, tuned):
        assert r["value"] > 0 and r["unit"] == "board-iterations/s" and r["kind"] == "port" and r["cores"] >= 1
    assert ref["cores"] <= 8                                   # the reference's N_THREADS (cfr.rs:195)
