"""Builds librustsolver_amd.so in-tree with hipcc for gfx950 (MI355X).

    python -m rustsolver_amd.build [--force]

-ffp-contract=off is part of the numerical contract (the Rust reference never fuses a*b+c);
do not remove it.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "librustsolver_amd.so")
SOURCES = ["rs_kernels.hip", "rs_table.cpp", "rs_tree.cpp", "rs_plan.cpp", "rs_plan_deals.cpp", "rs_solver.cpp", "rs_jit_check.cpp", "rs_knobs.cpp", "rs_comm.cpp", "rs_jit.cpp", "rs_jit_cache.cpp", "rs_abstraction.cpp", "rs_cards.hip", "rs_trainer.cpp", "rs_kmeans.hip", "rs_br.hip"]
HEADERS = [os.path.join(CSRC, "rs_internal.hpp"), os.path.join(CSRC, "rs_plan.hpp"), os.path.join(CSRC, "rs_plan_builder.hpp"), os.path.join(CSRC, "rs_device.hpp"), os.path.join(CSRC, "rs_hand_index.hpp"), os.path.join(CSRC, "rs_eval.hpp"), os.path.join(ROOT, "include", "rustsolver_amd.h"), os.path.join(ROOT, "include", "rustsolver_amd_diag.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + os.path.join(HERE, "_build")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _embed_device_source():
    """rs_device.hpp as a C++ raw string literal (rs_device_src.inc) for the hipRTC kernels of rs_jit.cpp"""
    src = os.path.join(CSRC, "rs_device.hpp")
    inc = os.path.join(HERE, "_build", "rs_device_src.inc")
    text = open(src).read().replace("#pragma once\n", "")
    assert ')RSDEV"' not in text
    body = 'R"RSDEV(' + text + ')RSDEV"\n'
    if not os.path.exists(inc) or open(inc).read() != body:
        with open(inc, "w") as f:
            f.write(body)
    return inc


def build(force=False, verbose=False):
    os.makedirs(os.path.join(HERE, "_build"), exist_ok=True)
    _embed_device_source()
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(HERE, "_build", src + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + HEADERS + [os.path.join(HERE, "_build", "rs_device_src.inc")]):
            cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    jitc_src, jitc = os.path.join(CSRC, "rs_jitc.cpp"), os.path.join(HERE, "rs_jitc")   # the compile helper rs_jit_cache.cpp starts (plain C++: hipRTC is loaded with dlopen)
    if force or _stale(jitc, [jitc_src]):
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", jitc_src, "-o", jitc, "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if force or _stale(SO, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
