"""Launcher: `python3 scratch/fuzz_api.py` (FUZZ_SEED, FUZZ_CASES, FUZZ_SIZE, FUZZ_LANCZOS) runs tests/fuzz_api.py."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fuzz_api

if __name__ == "__main__":
    sys.exit(1 if fuzz_api.run(int(os.environ.get("FUZZ_SEED", "0")), int(os.environ.get("FUZZ_CASES", "100")),
                         os.environ.get("FUZZ_SIZE"), os.environ.get("FUZZ_LANCZOS", "1") == "1") else 0)
