import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# The generated deal kernels exist in three forms -- one, two or four deals per thread (rs_solver.cpp: kernels with LDS tiles take one, the first round's two beyond
# 256 K deals per batch; kernels without tiles one and four) -- and every test batch is small: the deal-sweep tests below
# therefore run once with the library's own choice (one), once with RS_JIT_LANES=2 and once with RS_JIT_LANES=4, the forms big batches get.
DEALS_PER_THREAD_TESTS = {
    "test_deal_batches_vs_oracle", "test_deal_batches_large_cluster_counts", "test_deal_batches_many_trips_per_workgroup",
    "test_sparse_subtree_sweeps_three_streets_many_deals", "test_wide_nodes_in_deal_batches", "test_deal_trainer_reference_as_coded",
    "test_deal_trainer_ragged_batches", "test_data_parallel_ranks_equal_one_gpu_with_the_union_batch",
}
# (round 3: the two heaviest trainer tests -- the prune schedule and the flop-start game with bucket files, 50 s between them -- run the engine's own choice only; the
# four-deal forms of their kernels are the ones test_deal_batches_vs_oracle, test_sparse_subtree_sweeps_three_streets_many_deals and the ragged-batch test force)
# the two-deal forms are one more value of the same template parameter: the cheaper half of the list runs them too
TWO_DEALS_PER_THREAD_TESTS = {
    "test_deal_batches_vs_oracle", "test_deal_batches_many_trips_per_workgroup", "test_sparse_subtree_sweeps_three_streets_many_deals",
    "test_wide_nodes_in_deal_batches", "test_deal_trainer_reference_as_coded", "test_deal_trainer_ragged_batches",
}


# Lane tables of at least 2^20 lanes keep the rows of a node interleaved in 16 384-lane tiles (rs_table.cpp); the lane-model tests below run once
# with the library's own choice (plain rows at their sizes) and once with 64-lane tiles forced on every node wider than that.
TABLE_LAYOUT_TESTS = {
    "test_known_answers_update_and_strategy", "test_known_answers_discount", "test_golden_random_update_cases", "test_golden_random_discount_cases",
    "test_golden_extension_cases", "test_update_node_vs_oracle", "test_rmplus_i32_vs_oracle", "test_null_reach_means_one_and_per_board_copies",
    "test_zero_init_and_fill_mirror", "test_iterate_river_tree_vs_oracle", "test_iterate_three_street_tree_vs_oracle", "test_iterate_three_street_tree_pruned_vs_oracle",
    "test_iterate_extension_dtypes_vs_oracle", "test_iterate_sampled_opponent_vs_oracle", "test_wide_nodes_through_both_plans",
    "test_action_node_without_valid_actions", "test_train_with_discount_schedule_vs_oracle", "test_leaf_util_buffers_per_traverser",
    "test_checkpoint_roundtrip", "test_sharded_enum_sweep_equals_single_gpu", "test_allreduce_replicated_single_rank_is_identity",
    "test_calc_br_equals_oracle",
}


# Lane sweeps with ENUM chance nodes: a fused subtree directly below a chance node takes over the node's expand step (rs_jit.cpp `_xr` kernels, the default when
# n_clusters % 4 == 0) or leaves it to an expand launch (rs_kernel_forms.lane_fan = RS_FAN_NONE: what cluster counts that are no multiple of four get anyway): these tests run
# in both forms (the fixture sets the wrapper's default rs_kernel_forms for the test).
FAN_LOOP_TESTS = {
    "test_iterate_three_street_tree_vs_oracle", "test_iterate_three_street_tree_pruned_vs_oracle", "test_sharded_enum_sweep_equals_single_gpu",
    "test_wide_nodes_through_both_plans", "test_action_node_without_valid_actions",
}
# (round 3, to keep the GPU suite under 450 s: the 5 000-cluster whole-table test runs the engine's own form only; the other fan forms meet 5 000 clusters in the full-size tests)
# (test_randomised_differential draws its own form per seed: 32 cases instead of 192, every form still met eight times)


# Deal batches up to 128 K deals run the kernels of a group of independent round subtrees as ONE merged kernel on one stream (rs_solver.cpp merge_small_groups); larger ones
# keep a launch per subtree shape, spread over the auxiliary streams.  Every test batch is small: these tests run both ways (RS_JIT_NO_MERGE forces the large-batch form).
MERGE_TESTS = {
    "test_sparse_subtree_sweeps_three_streets_many_deals", "test_ordered_deal_sweeps_vs_oracle", "test_delta_rows_deal_sweeps_vs_oracle",
    "test_deal_trainer_three_streets_from_a_flop_with_bucket_files", "test_deal_trainer_deals_and_sorts_ahead",
}


def pytest_generate_tests(metafunc):
    if metafunc.function.__name__ in MERGE_TESTS:
        if "merge_form" not in metafunc.fixturenames:
            metafunc.fixturenames.append("merge_form")
        metafunc.parametrize("merge_form", ["merged", "separate"], indirect=True)
    if metafunc.function.__name__ in FAN_LOOP_TESTS:
        if "fan_loop" not in metafunc.fixturenames:
            metafunc.fixturenames.append("fan_loop")
        metafunc.parametrize("fan_loop", ["xr", "off"], indirect=True)
    if metafunc.function.__name__ in TABLE_LAYOUT_TESTS:
        if "table_layout" not in metafunc.fixturenames:
            metafunc.fixturenames.append("table_layout")
        metafunc.parametrize("table_layout", ["plain", "tiled64"], indirect=True)
    if metafunc.function.__name__ in DEALS_PER_THREAD_TESTS:
        if "deals_per_thread" not in metafunc.fixturenames:
            metafunc.fixturenames.append("deals_per_thread")
        metafunc.parametrize("deals_per_thread", ["auto", "2", "4"] if metafunc.function.__name__ in TWO_DEALS_PER_THREAD_TESTS else ["auto", "4"], indirect=True)


@pytest.fixture
def merge_form(request, monkeypatch):
    if request.param == "separate":
        monkeypatch.setenv("RS_JIT_NO_MERGE", "1")
    return request.param


@pytest.fixture
def deals_per_thread(request, monkeypatch):
    if request.param != "auto":
        monkeypatch.setenv("RS_JIT_LANES", request.param)
    return request.param


@pytest.fixture
def fan_loop(request, monkeypatch):
    from rustsolver_amd import solver
    monkeypatch.setattr(solver, "DEFAULT_FORMS", {"lane_fan": 2 if request.param == "xr" else 1})   # RS_FAN_EXPAND / RS_FAN_NONE
    return request.param


@pytest.fixture
def table_layout(request, monkeypatch):
    if request.param == "tiled64":
        monkeypatch.setenv("RS_TABLE_TILE_LANES", "64")   # every node wider than 64 lanes is tiled in 64-lane tiles
    return request.param
