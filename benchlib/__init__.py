"""Parts of bench.py (the repo-root entry point the driver runs): common, cpu_baseline, traffic, legs."""
