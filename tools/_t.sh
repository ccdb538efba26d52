cd /root/repo
python -m pytest tests -m gpu -q --timeout=2400 --deselect tests/test_bench_multirank_gpu.py::test_c4_at_full_size_eight_ranks_equal_one_process 2>&1 | tail -12
