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
    assert ref["value"] > 0 and ref["unit"] == "board-iterations/s" and ref["kind"] == "port" and 1 <= ref["cores"] <= 8   # the reference's N_THREADS (cfr.rs:195)
    for mode in ("clamp", "wrap"):
        soa = b.cpu_soa(50, mode, 0.2, boards=40, threads=3)      # asserts bit-identity with the per-lane oracle before it times anything
        assert soa["identical_to_per_lane_oracle"] is True
        assert soa["value"] > 0 and soa["unit"] == "board-iterations/s" and soa["cores"] == 3 and soa["algo_GBps"] > 0


def _run_bench(args, env_extra, timeout=240):
    import subprocess
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus N` without torch.distributed.run must start N children with distinct ranks that can rendezvous (here: gloo, no
    GPU -- RS_BENCH_PROBE makes a child stop after the rendezvous), and print exactly rank 0's single JSON line."""
    import json
    r = _run_bench(["--gpus", "3", "--steps", "2", "--warmup", "1"], {"RS_BENCH_PROBE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["ranks"] == [0, 1, 2] and out["local_ranks"] == [0, 1, 2] and out["distinct_pids"] == 3
    assert out["master"] == "127.0.0.1"


def test_a_failing_rank_fails_the_whole_launch():
    r = _run_bench(["--gpus", "3"], {"RS_BENCH_PROBE": "fail"})
    assert r.returncode == 3 and r.stdout.strip() == ""
    assert "rank 1 exited with code 3" in r.stderr


def test_the_launching_parent_never_touches_the_gpu():
    """the parent of a self-launched run may not initialise HIP (a process that has must not start GPU children on this pool): launch_ranks and
    everything main() runs before it import neither torch nor the engine"""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    fn = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    for name in ("launch_ranks", "_free_port", "parse"):
        for node in ast.walk(fn[name]):
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                mods = [a.name for a in node.names] if isinstance(node, ast.Import) else [node.module or ""]
                assert not any(m.split(".")[0] in ("torch", "rustsolver_amd", "oracle") for m in mods), (name, mods)
    main_src = ast.get_source_segment(src, fn["main"])
    assert main_src.index("launch_ranks(") < main_src.index("import torch") and main_src.index("launch_ranks(") < main_src.index("import rustsolver_amd")
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    for node in top:
        mods = [a.name for a in node.names] if isinstance(node, ast.Import) else [node.module or ""]
        assert not any(m.split(".")[0] in ("torch", "rustsolver_amd", "oracle") for m in mods)


def test_rank_local_setup_failures_are_reported_not_hung_on():
    """bench._agree: a setup step that fails raises on the spot with the leg's name in the message (with a process group every rank first learns
    whether it worked everywhere, so nobody enters an RCCL collective alone); a step that works hands its result through"""
    import pytest
    b = load_bench()
    assert b._agree(None, lambda: 41 + 1, "setup") == 42

    def boom():
        raise MemoryError("no room")
    with pytest.raises(RuntimeError) as e:
        b._agree(None, boom, "config4 tables and launch plan")
    assert "config4 tables and launch plan" in str(e.value) and "no room" in str(e.value)
