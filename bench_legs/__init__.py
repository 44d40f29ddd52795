"""Legs of the benchmark that are not the timed step: bench.py runs them behind --full and writes them to bench_detail.json."""
