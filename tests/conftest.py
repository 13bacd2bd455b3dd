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


# The generated deal kernels exist in two forms -- four deals per thread, and one for batches of up to 256 K deals (rs_solver.cpp,
# kSmallDealBatch) -- and every test batch is small: the deal-sweep tests below therefore run once with the library's own choice (one) and
# once with RS_JIT_LANES=4, the form the big batches of bench.py use.
DEALS_PER_THREAD_TESTS = {
    "test_deal_batches_vs_oracle", "test_deal_batches_large_cluster_counts", "test_deal_batches_many_trips_per_workgroup",
    "test_sparse_subtree_sweeps_three_streets_many_deals", "test_wide_nodes_in_deal_batches", "test_deal_trainer_reference_as_coded",
    "test_deal_trainer_prune_schedule", "test_deal_trainer_three_streets_from_a_flop_with_bucket_files", "test_deal_trainer_ragged_batches",
    "test_data_parallel_ranks_equal_one_gpu_with_the_union_batch",
}


def pytest_generate_tests(metafunc):
    if metafunc.function.__name__ in DEALS_PER_THREAD_TESTS:
        if "deals_per_thread" not in metafunc.fixturenames:
            metafunc.fixturenames.append("deals_per_thread")
        metafunc.parametrize("deals_per_thread", ["auto", "4"], indirect=True)


@pytest.fixture
def deals_per_thread(request, monkeypatch):
    if request.param != "auto":
        monkeypatch.setenv("RS_JIT_LANES", request.param)
    return request.param
