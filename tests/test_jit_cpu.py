"""CPU (-m "not gpu"): the compile path of the tree-specialised kernels.  hipRTC cross-compiles gfx950 without a GPU, so the CPU suite checks that the generated sources build --
through the helper processes rs_jit_cache.cpp starts (rs_jitc: hipRTC serialises compiles inside a process) and through the in-process fallback a missing or failing helper
leaves (solver creation must not depend on one compile route)."""
import os
import subprocess

import pytest

import rustsolver_amd as rs
from rustsolver_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JITC = os.path.join(ROOT, "rustsolver_amd", "rs_jitc")
needs_rtc = pytest.mark.skipif(not L.load().rs_jit_available(), reason="libhiprtc.so cannot be loaded here")


@needs_rtc
def test_compile_helper_compiles_and_reports(tmp_path):
    """rs_jitc <source> <output> ...: a good source leaves a code object, a bad one `<output>.log` with the compiler's words and a non-zero exit code; the other pairs of the
    same call are still compiled"""
    assert os.access(JITC, os.X_OK), "rustsolver_amd/build.py builds the helper next to the library"
    good, bad = tmp_path / "good.hip", tmp_path / "bad.hip"
    good.write_text('extern "C" __global__ void k(float *x) { x[threadIdx.x] = 1.0f; }\n')
    bad.write_text('extern "C" __global__ void k(float *x) { x[threadIdx.x] = nonsense; }\n')
    r = subprocess.run([JITC, str(bad), str(tmp_path / "bad.hsaco"), str(good), str(tmp_path / "good.hsaco")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1
    assert (tmp_path / "good.hsaco").stat().st_size > 1000 and (tmp_path / "good.hsaco").read_bytes()[:4] == b"\x7fELF"
    assert not (tmp_path / "bad.hsaco").exists() and "nonsense" in (tmp_path / "bad.hsaco.log").read_text()
    assert subprocess.run([JITC], capture_output=True).returncode == 100    # usage


@needs_rtc
def test_generated_kernels_compile_through_helpers_and_in_process(monkeypatch):
    """the river tree's generated lane kernels: the same sources through the helper processes and (RS_JIT_NO_PROCS, = what a missing helper leaves) one by one in this process"""
    _, tree = rs.build_game_tree(rs.default_flop())
    n_helpers = rs.jit_check_tree(tree, rs.I32, rs.UPD_CLAMP_I64)
    monkeypatch.setenv("RS_JIT_NO_PROCS", "1")
    assert rs.jit_check_tree(tree, rs.I32, rs.UPD_CLAMP_I64) == n_helpers == 2


@needs_rtc
def test_a_failing_helper_falls_back_to_this_process(monkeypatch):
    """a helper that cannot do its job (here: hipRTC hidden from it) delivers nothing; the library then compiles the sources itself instead of failing the call"""
    _, tree = rs.build_game_tree(rs.default_flop())
    before = rs.jit_check_tree(tree, rs.I32, rs.UPD_WRAP_I32)
    monkeypatch.setenv("RS_JITC_SELFTEST_FAIL", "1")   # the helper's own test hook: it exits at once with nothing done
    assert rs.jit_check_tree(tree, rs.I32, rs.UPD_WRAP_I32) == before
