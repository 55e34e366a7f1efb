"""Test suite of vsdeoldify_amd: `-m "not gpu"` (oracle pinning, host logic, C ABI surface) and `-m gpu` (HIP vs oracle)."""
