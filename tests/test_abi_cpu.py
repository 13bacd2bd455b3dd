"""CPU: the C-ABI library loads and exports every symbol include/rustsolver_amd.h declares (no compute
calls), and the host-side logic that needs no GPU (the public-tree builder) matches the oracle."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import rustsolver_amd as rs
from oracle import np_restate as npr
from oracle import orc
from rustsolver_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(which=("rustsolver_amd.h", "rustsolver_amd_diag.h")):
    names = set()
    for h in which:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_every_declared_symbol_is_exported_and_bound():
    names = header_functions()
    assert len(names) >= 50
    lib = C.CDLL(L.SO_PATH)
    for n in names:
        assert hasattr(lib, n), "librustsolver_amd.so does not export %s" % n
    assert sorted(L.SYMBOLS) == names, set(L.SYMBOLS) ^ set(names)
    assert lib.rs_abi_version() == 6


def test_no_gpu_means_loud_failure_not_fallback():
    if rs.device_count() > 0:
        pytest.skip("a GPU is visible")
    n, tree = rs.build_game_tree(rs.default_flop())
    with pytest.raises(rs.RsError) as e:
        rs.create_infosets(n, tree, [8])
    assert e.value.code == L.ERR_HIP and "no CPU fallback" in str(e.value)


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rustsolver_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "rs_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def _same_tree(tree, otree):
    assert (tree.n_nodes, tree.n_action_nodes) == (otree.n_nodes, otree.n_action_nodes)
    for nd, d in zip(tree.nodes, otree.as_dicts()):
        assert (nd.kind, nd.parent, [nd.children[k] for k in range(nd.n_children)]) == (d["kind"], d["parent"], d["children"])
        if nd.kind == L.NODE_ACTION:
            assert (nd.index, nd.player, nd.round_idx) == (d["index"], d["player"], d["round_idx"])
            assert [[nd.action_kind[k], nd.action_amt[k]] for k in range(nd.n_children)] == d["actions"]
        elif nd.kind == L.NODE_TERMINAL:
            assert (nd.value, nd.ttype, nd.last_to_act, nd.round) == (d["value"], d["ttype"], d["last_to_act"], d["round"])
        elif nd.kind == L.NODE_PUBLIC_CHANCE:
            assert nd.round == d["round"]


def test_default_tree_matches_oracle_and_golden(golden_dir):
    n, tree = rs.build_game_tree(rs.default_flop())        # tree_builder.rs:9 on options.rs:52-81
    assert n == 14 and tree.n_nodes == 40
    _same_tree(tree, orc.OracleTree(orc.options_default_river()))
    fx = json.load(open(os.path.join(golden_dir, "tree_river.json")))
    acts = {L.ACT_BET: "bet", L.ACT_RAISE: "raise", L.ACT_CHECK: "check", L.ACT_CALL: "call", L.ACT_FOLD: "fold"}
    for nd, g in zip(tree.nodes, fx["nodes"]):
        if g["kind"] == "action":
            assert [[acts[nd.action_kind[k]], nd.action_amt[k]] for k in range(nd.n_children)] == g["actions"]
            assert nd.index == g["index"] and nd.player == g["player"]
        if g["kind"] == "terminal":
            assert nd.value == g["value"] and nd.last_to_act == g["last_to_act"]


def test_three_street_tree_matches_oracle_and_golden(golden_dir):
    n, tree = rs.build_game_tree(rs.three_street_options())
    assert (n, tree.n_nodes) == (706, 1864)
    _same_tree(tree, orc.OracleTree(orc.options_three_street()))
    fx = json.load(open(os.path.join(golden_dir, "tree_three_street.json")))
    assert [[nd.index, nd.player, nd.round_idx, nd.n_children] for nd in tree.nodes if nd.kind == L.NODE_ACTION] == fx["action_nodes"]


@pytest.mark.parametrize("seed", range(12))
def test_random_options_trees_match_oracle(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    nb = int(rng.integers(3, 6))
    rounds = 6 - nb
    bets = [sorted(rng.choice([0.25, 0.33, 0.5, 0.75, 1.0, 1.5, 2.0], size=int(rng.integers(1, 4)), replace=False).tolist())
            for _ in range(rounds)]
    raises = [sorted(rng.choice([2.0, 2.5, 3.0, 4.0], size=int(rng.integers(1, 3)), replace=False).tolist()) for _ in range(rounds)]
    stacks = (int(rng.integers(20, 2000)), int(rng.integers(20, 2000)))
    pot = int(rng.integers(2, 300))
    n, tree = rs.build_game_tree(rs.Options(stacks, pot, nb, bets, raises))
    otree = orc.OracleTree(orc.make_options(stacks, pot, nb, bets, raises))
    _same_tree(tree, otree)
    pn, pcount = npr.build_tree(stacks, pot, nb, bets, raises)
    assert pcount == n and len(pn) == tree.n_nodes


def test_tree_errors_instead_of_panics():
    with pytest.raises(rs.RsError) as e:                      # state.rs:64 panic!("invalid board mask")
        rs.build_game_tree(rs.Options(n_board_cards=2))
    assert "invalid board mask" in str(e.value)
    with pytest.raises(rs.RsError) as e:                      # bet_sizes[round_idx] index panic on the turn
        rs.build_game_tree(rs.Options(n_board_cards=3, bet_sizes=((0.5,),), raise_sizes=((3.0,),)))
    assert e.value.code == L.ERR_OOB


def test_tree_from_nodes_roundtrip_and_validation():
    _, tree = rs.build_game_tree(rs.default_flop())
    t2 = rs.tree_from_nodes(tree.nodes)
    assert (t2.n_nodes, t2.n_action_nodes) == (40, 14)
    for a, b in zip(tree.nodes, t2.nodes):
        assert bytes(a) == bytes(b)
    bad = [L.TreeNode.from_buffer_copy(bytes(n)) for n in tree.nodes]
    bad[1].children[0] = 99
    with pytest.raises(rs.RsError):
        rs.tree_from_nodes(bad)
    bad = [L.TreeNode.from_buffer_copy(bytes(n)) for n in tree.nodes]
    bad[1].index = 5                                           # duplicate ActionNode.index
    with pytest.raises(rs.RsError):
        rs.tree_from_nodes(bad)


def test_discount_factor_matches_reference_rule():
    for tc in (0, 99999, 100000, 100001, 250000, 1999999, 20000000, 2**40):
        assert rs.discount_factor(tc).view(np.uint32) == orc.discount_factor(tc).view(np.uint32)
        assert rs.discount_factor(tc).view(np.uint32) == npr.discount_factor(tc).view(np.uint32)


def test_synth_mirror_is_deterministic():
    a = rs.synth.fill_values(7, np.arange(1000), -10**6, 10**6)
    b = rs.synth.fill_values(7, np.arange(1000), -10**6, 10**6)
    assert (a == b).all() and a.min() >= -10**6 and a.max() <= 10**6 and len(set(a.tolist())) > 900
    u = rs.synth.uniform_f32(3, 1000, -1.0, 1.0)
    assert u.dtype == np.float32 and (u >= -1).all() and (u < 1).all()


def test_tree_specialised_kernels_compile_without_a_gpu():
    """hipRTC cross-compiles gfx950 here: every generated kernel of the default trees must build"""
    from rustsolver_amd import _lib as L2
    if not L2.load().rs_jit_available():
        pytest.skip("libhiprtc.so not present")
    _, tree = rs.build_game_tree(rs.default_flop())
    assert rs.jit_check_tree(tree, rs.I32, rs.UPD_CLAMP_I64) == 2          # one kernel per traverser
    assert rs.jit_check_tree(tree, rs.F16, rs.UPD_CLAMP_I64) == 2
    assert rs.jit_check_tree(tree, rs.I32, rs.UPD_CLAMP_I64, rs.OPP_SAMPLE) == 2     # mccfr(): sampled opponent
    _, tree3 = rs.build_game_tree(rs.Options(n_board_cards=4, bet_sizes=((0.5,), (1.0,)), raise_sizes=((3.0,), (3.0,))))
    assert rs.jit_check_tree(tree3, rs.I32, rs.UPD_WRAP_I32) >= 2
    # pruned lane sweeps (RS_UPD_PRUNE): the generated kernels' pruned forms, river tree and round subtrees
    assert rs.jit_check_tree(tree, rs.I32, rs.UPD_CLAMP_I64 | rs.UPD_PRUNE) == 2
    assert rs.jit_check_tree(tree3, rs.I32, rs.UPD_CLAMP_I64 | rs.UPD_PRUNE) >= 2
    # deal batches: round subtrees (reach-down half, walk; dense / live-deal list; LDS tiles / direct atomics), sampled and full width
    # river tree: no reach-down half; 4 walks per traverser, each with 4 and 1 deals per thread, + the work-list form of the list walk with LDS tiles (20), + the ordered forms
    # (rs_kernel_forms.deal_order): the segment-summing walk over the whole batch and over a list, one deal per thread (4), + the f32-table forms (per-deal delta rows, dense
    # walks: 4; round 5: the one-deal-per-thread walk on a binary16 table too: 2), + the delta-rows forms (rs_kernel_forms.delta_rows: the list walk stores its deltas by position, 4 and 1 deals per thread: 4), + the staged
    # form of the list walk (round 4: the wave copies its deals' shadow rows into LDS, no gather per node: 2)
    assert rs.jit_check_tree_deals(tree, rs.UPD_CLAMP_I64, rs.OPP_SAMPLE) == 36
    assert rs.jit_check_tree_deals(tree3, rs.UPD_CLAMP_I64, rs.OPP_SAMPLE) > 8
    assert rs.jit_check_tree_deals(tree3, rs.UPD_WRAP_I32, rs.OPP_FULL) >= 4
    # the pruned forms (cfr.rs:379-386 per deal) of the same kernels
    assert rs.jit_check_tree_deals(tree, rs.UPD_CLAMP_I64 | rs.UPD_PRUNE, rs.OPP_SAMPLE) == 36
    assert rs.jit_check_tree_deals(tree3, rs.UPD_CLAMP_I64 | rs.UPD_PRUNE, rs.OPP_SAMPLE) > 8


def test_generated_deal_kernels_use_no_scratch(monkeypatch):
    """round 5 found 40-64 bytes of private memory per lane in every reach-down kernel (a chain of selects over a small array, compiled into one load at a computed address: a
    store and a dependent load through memory at each handed-over node; 5 % of a 4 M-deal batch, 6 % of a 64 K one).  The generated sources of a two-round tree's sampled,
    pruned deal kernels -- dumped by the compile check, compiled here with hipcc as hipRTC compiles them -- must report `ScratchSize [bytes/lane]: 0`."""
    import glob, os, shutil, subprocess, time
    from rustsolver_amd import _lib as L2
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not L2.load().rs_jit_available() or not os.path.exists(hipcc):
        pytest.skip("libhiprtc.so or hipcc not present")
    monkeypatch.setenv("RS_JIT_DUMP", "1")
    t0 = time.time() - 1.0
    _, tree3 = rs.build_game_tree(rs.Options(n_board_cards=4, bet_sizes=((0.5,), (1.0,)), raise_sizes=((3.0,), (3.0,))))
    assert rs.jit_check_tree_deals(tree3, rs.UPD_CLAMP_I64 | rs.UPD_PRUNE, rs.OPP_SAMPLE) > 8
    fresh = [f for f in glob.glob("/tmp/rs_tree_kernel_*.hip") if os.path.getmtime(f) >= t0]
    import re
    down = sorted(f for f in fresh if "_deals_down_sampled" in open(f).read() and re.search(r"hand_pick<\d+>\(", open(f).read()))
    assert down, "the compile check dumped no reach-down kernel that hands its draws over"
    picked = down[:3] + sorted(f for f in fresh if f not in down)[:2]   # three reach-down kernels that hand their draws over (the ones that had it) and two others
    procs = [subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-include", "hip/hip_runtime.h", "-c", f, "-o", os.devnull,
                               "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for f in picked]
    for f, pr in zip(picked, procs):
        out = pr.communicate()[0]
        assert pr.returncode == 0, out[-2000:]
        sizes = [l.split("ScratchSize [bytes/lane]:")[1].split()[0] for l in out.splitlines() if "ScratchSize [bytes/lane]:" in l]
        assert sizes and all(x == "0" for x in sizes), (f, sizes)


# ---- card-abstraction plumbing (card_abstraction.rs) --------------------------------------------------

def test_cluster_file_roundtrip_and_format(tmp_path):
    from rustsolver_amd import abstraction as ab
    rng = np.random.Generator(np.random.PCG64(3))
    arr = rng.integers(0, 5000, size=12345, dtype=np.uint32)
    path = str(tmp_path / "round_1_emd.dat")
    ab.write_cluster_file(path, arr)
    raw = open(path, "rb").read()
    assert len(raw) == 4 * len(arr) and raw[:8] == arr[:2].astype("<u4").tobytes()      # flat little-endian u32
    assert (ab.read_cluster_file(path) == arr).all()
    with pytest.raises(rs.RsError):                                                      # create_new(true): never overwrite
        ab.write_cluster_file(path, arr)
    with pytest.raises(rs.RsError):                                                      # File::open(..).unwrap()
        ab.read_cluster_file(str(tmp_path / "missing.dat"))
    open(str(tmp_path / "odd.dat"), "wb").write(b"abcde")
    with pytest.raises(rs.RsError):
        ab.read_cluster_file(str(tmp_path / "odd.dat"))


def test_index_to_cluster_and_dense_map():
    from rustsolver_amd import abstraction as ab
    cluster_arr = np.array([7, 7, 3, 9, 3, 0], dtype=np.uint32)
    idx = np.array([5, 0, 1, 2, 4, 3], dtype=np.uint64)
    assert ab.index_to_cluster(idx).tolist() == idx.tolist()                             # ISOMORPHIC: the index itself
    b = ab.index_to_cluster(idx, cluster_arr)
    assert b.tolist() == [0, 7, 7, 3, 3, 9]
    with pytest.raises(IndexError):                                                      # arr[index] out of bounds panics
        ab.index_to_cluster([6], cluster_arr)
    m = ab.DenseMap(b)
    assert len(m) == 4 and m.keys().tolist() == [0, 7, 3, 9]                              # first-appearance order, deterministic
    assert m.lookup([9, 0, 3, 7, 7]).tolist() == [3, 0, 2, 1, 1]
    with pytest.raises(KeyError):                                                        # .get(&cluster).unwrap() on None
        m.lookup([5])
    # two players, different ranges -> different sizes (size[0] != size[1] in general, infoset.rs:28-32)
    m2 = ab.DenseMap(ab.index_to_cluster([0, 1], cluster_arr))
    assert len(m2) == 1


def test_oversize_and_malformed_tables_are_rejected_before_touching_the_gpu():
    from rustsolver_amd import _lib as L2
    with pytest.raises(rs.RsError) as e:                    # a node of 2^31+ cells must be sharded over the board axis
        rs.InfosetTable.create([(8, 5000, 60000, 0, 0)])
    assert e.value.code == L2.ERR_UNSUPPORTED
    for bad in ([(3, 0, 1, 0, 0)], [(3, 10, 0, 0, 0)], [(3, 10, 1, 2, 0)], [(3, 10, 1, 0, 3)], [(9, 10, 1, 0, 0)]):
        with pytest.raises(rs.RsError) as e:
            rs.InfosetTable.create(bad)
        assert e.value.code == L2.ERR_INVALID


def test_header_is_plain_c_and_the_example_driver_links(tmp_path):
    """include/rustsolver_amd.h must be consumable by a C compiler (bindgen / cgo / a C host): examples/solver_main.c -- the reference's
    `solver` binary (src/solver/main.rs:29-36) over the C ABI -- builds as strict C99 and links against the in-tree library.  Without a GPU it
    must fail loudly, never fall back."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "solver_main")
    libdir = os.path.join(root, "rustsolver_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-D_POSIX_C_SOURCE=199309L", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "solver_main.c"), "-L" + libdir, "-lrustsolver_amd", "-Wl,-rpath," + libdir, "-o", exe])
    exe1 = str(tmp_path / "config1_main")   # BASELINE configs[0]: the 169-bucket two-action tree adopted with rs_tree_from_nodes
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "config1_main.c"), "-L" + libdir, "-lrustsolver_amd", "-Wl,-rpath," + libdir, "-o", exe1])
    import rustsolver_amd as rs
    if rs.device_count() > 0:
        pytest.skip("a GPU is visible: the run itself is covered by the GPU tests")
    r = subprocess.run([exe, "1000"], capture_output=True, text=True)
    assert r.returncode == 1 and "no usable HIP device" in r.stderr and "1081 / 1081 river clusters" in r.stdout
    r = subprocess.run([exe1], capture_output=True, text=True)
    assert r.returncode == 1 and "no usable HIP device" in r.stderr


def test_rust_binding_covers_every_export():
    """rust/ffi.rs is generated from the headers (tools/gen_rust_ffi.py): it must be current, declare EVERY exported function with the header's arity,
    and mirror every struct field by field (ctypes declares the same structs: their field names must agree too)."""
    import subprocess
    import sys
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"]) == 0
    rust = open(os.path.join(ROOT, "rust", "ffi.rs")).read()
    rfns = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (rs_\w+)\(([^)]*)\)", rust)}
    assert sorted(rfns) == header_functions()
    csrc = ""
    for h in ("rustsolver_amd.h", "rustsolver_amd_diag.h"):
        csrc += re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", h)).read(), flags=re.S)
    for name, rargs in rfns.items():
        m = re.search(r"\b%s\s*\(([^)]*)\)\s*;" % name, csrc)
        cargs = m.group(1).strip()
        n_c = 0 if cargs in ("", "void") else cargs.count(",") + 1
        n_r = 0 if not rargs.strip() else rargs.count(":")
        assert n_c == n_r, "%s: %d arguments in the header, %d in rust/ffi.rs" % (name, n_c, n_r)
        assert len(L.SYMBOLS[name][1]) == n_c, "%s: ctypes arity" % name
    for cname, cls in (("rs_tree_node", L.TreeNode), ("rs_options", L.OptionsC), ("rs_node_desc", L.NodeDesc), ("rs_leaf_desc", L.LeafDesc),
                       ("rs_deal_batch", L.DealBatch), ("rs_solver_params", L.SolverParams)):
        m = re.search(r"pub struct %s \{(.*?)\n\}" % cname, rust, flags=re.S)
        rfields = re.findall(r"pub (\w+):", m.group(1))
        assert rfields == [f[0] for f in cls._fields_], cname


def test_rust_binding_is_well_formed():
    """There is no rustc in the build image, so the generated binding is held to a strict line grammar instead (round 3 shipped `-> endif int` for one prototype and
    nothing noticed): every item line is one of the forms the generator emits, every type is a known scalar, a declared struct or a pointer / array of those; with a rustc on
    PATH the file is also compiled as a library."""
    import shutil
    import subprocess
    rust = open(os.path.join(ROOT, "rust", "ffi.rs")).read()
    declared = set(re.findall(r"pub struct (\w+)", rust))
    scalars = {"c_char", "c_int", "c_uint", "c_void", "f32", "f64", "usize", "u8", "u16", "u32", "u64", "i8", "i16", "i32", "i64"}

    def type_ok(t):
        t = t.strip()
        while True:
            m = re.match(r"^\*(?:const|mut) (.*)$", t)
            if m:
                t = m.group(1)
                continue
            m = re.match(r"^\[(.*); (\w+)\]$", t)
            if m:
                t = m.group(1)
                continue
            return t in scalars or t in declared

    in_extern = False
    for ln, line in enumerate(rust.split("\n"), 1):
        s = line.strip()
        if not s or s.startswith("//") or s.startswith("#![") or s.startswith("#[") or s.startswith("use "):
            continue
        if s == 'extern "C" {':
            in_extern = True
            continue
        if s == "}":
            in_extern = False
            continue
        if in_extern:
            m = re.fullmatch(r"pub fn (rs_\w+)\((.*)\)(?: -> (.+))?;", s)
            assert m, "rust/ffi.rs:%d: not a prototype: %s" % (ln, s)
            for arg in (a for a in m.group(2).split(", ") if a):
                nm, ty = arg.split(": ", 1)
                assert re.fullmatch(r"[a-z_]\w*", nm) and type_ok(ty), "rust/ffi.rs:%d: argument %r" % (ln, arg)
            assert m.group(3) is None or type_ok(m.group(3)), "rust/ffi.rs:%d: return type %r" % (ln, m.group(3))
            continue
        if re.fullmatch(r"pub const \w+: (usize|c_int) = [-\w]+;", s) or re.fullmatch(r"pub struct \w+ \{( _private: \[u8; 0\] \})?", s):
            continue
        m = re.fullmatch(r"pub (\w+): (.+),", s)
        assert m and type_ok(m.group(2)), "rust/ffi.rs:%d: unexpected line: %s" % (ln, s)
    rustc = shutil.which("rustc")
    if rustc:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            assert subprocess.call([rustc, "--crate-type", "lib", "--emit", "metadata", "-o", os.path.join(d, "ffi.rmeta"), os.path.join(ROOT, "rust", "ffi.rs")]) == 0
